"""Scenario container (reference: main/lib/scenario.py:7-13) and the stock 4-way intersection worlds
(reference: main/envs/intersection.py:10-216) rebuilt from data/intersection_scenarios.npz. The reference's own
`envs.*` modules produce compatible objects and can be passed to MotionPrimitiveSearch unchanged."""
import os
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from .obstacles import BoxObstacle, CircleObstacle, Obstacle

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'data', 'intersection_scenarios.npz')


@dataclass
class Scenario:
    start: Tuple[float, float, float]
    goal_point: Tuple[float, float, float]
    goal_area: Obstacle
    allowed_goal_theta_difference: float
    obstacles: List[Obstacle]


def intersection(turn_indicator: int, start_pos: int) -> Scenario:
    z = np.load(_DATA)
    key = 'int_%d_%d' % (start_pos, turn_indicator)
    if key + '/start' not in z.files:
        raise KeyError('no stock intersection for start_pos=%r turn_indicator=%r' % (start_pos, turn_indicator))
    ga = z[key + '/goal_area']
    goal_area = BoxObstacle(xy_width=(ga[2] - ga[0], ga[3] - ga[1]), height=0.5,
                            xy_center=((ga[0] + ga[2]) / 2, (ga[1] + ga[3]) / 2))
    goal_area.xy1, goal_area.xy2 = (float(ga[0]), float(ga[1])), (float(ga[2]), float(ga[3]))
    obstacles: List[Obstacle] = []
    for kind, prm in zip(z[key + '/obst_kind'], z[key + '/obst_param']):
        if kind == 0:
            o = BoxObstacle(xy_width=(prm[2] - prm[0], prm[3] - prm[1]), height=0.5,
                            xy_center=((prm[0] + prm[2]) / 2, (prm[1] + prm[3]) / 2), hidden=bool(prm[4]))
            o.xy1, o.xy2 = (float(prm[0]), float(prm[1])), (float(prm[2]), float(prm[3]))   # exact corner values
        else:
            o = CircleObstacle(radius=float(prm[2]), height=0.5, xy_center=(float(prm[0]), float(prm[1])), hidden=bool(prm[4]))
        obstacles.append(o)
    return Scenario(start=tuple(float(v) for v in z[key + '/start']), goal_point=tuple(float(v) for v in z[key + '/goal_point']),
                    goal_area=goal_area, allowed_goal_theta_difference=float(z[key + '/allowed_dtheta']), obstacles=obstacles)
