"""Scenario container (reference: main/lib/scenario.py:7-13) and every world of the reference's main/envs/*.py.

The geometry lives in data/worlds.npz as plain obstacle-parameter tables (box corners / circle centre + radius, start
pose, goal pose, goal box), one group per (environment, arguments) combination the reference can build; the
functions below look a world up under the reference's names and argument meaning and rebuild Box/CircleObstacle
objects with the exact corner values (so half-planes match bit for bit).  `free_area` and `ArterialMultiLanes` are
closed-form (no obstacle layout to tabulate).  Scenario objects produced by the reference's own `envs.*` modules
are interchangeable: MotionPrimitiveSearch only needs `.start/.goal_point/.goal_area/.obstacles[*].to_convex`.
"""
import os
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from .obstacles import BoxObstacle, CircleObstacle, Obstacle

_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'data')
_WORLDS = None


@dataclass
class Scenario:
    start: Tuple[float, float, float]
    goal_point: Tuple[float, float, float]
    goal_area: Obstacle
    allowed_goal_theta_difference: float
    obstacles: List[Obstacle]


def _tables():
    global _WORLDS
    if _WORLDS is None:
        _WORLDS = np.load(os.path.join(_DIR, 'worlds.npz'))
    return _WORLDS


def _box(x1, y1, x2, y2, hidden=False) -> BoxObstacle:
    o = BoxObstacle(xy_width=(x2 - x1, y2 - y1), height=0.5, xy_center=((x1 + x2) / 2, (y1 + y2) / 2), hidden=bool(hidden))
    o.xy1, o.xy2 = (float(x1), float(y1)), (float(x2), float(y2))     # exact corner values of the reference's object
    return o


def available_worlds() -> List[str]:
    return sorted({k.rsplit('/', 1)[0] for k in _tables().files})


def world(key: str) -> Scenario:
    """world('roundabout_big/1_1'), world('intersection_multi_lanes/1_1_2_1_2'), ... (see available_worlds())"""
    z = _tables()
    if key + '/start' not in z.files:
        raise KeyError('no tabulated world %r (the reference does not build it either, or builds it with other arguments)' % key)
    obstacles: List[Obstacle] = []
    for kind, prm in zip(z[key + '/obst_kind'], z[key + '/obst_param']):
        if kind == 0:
            obstacles.append(_box(prm[0], prm[1], prm[2], prm[3], prm[4]))
        else:
            obstacles.append(CircleObstacle(radius=float(prm[2]), height=0.5, xy_center=(float(prm[0]), float(prm[1])), hidden=bool(prm[4])))
    return Scenario(start=tuple(float(v) for v in z[key + '/start']), goal_point=tuple(float(v) for v in z[key + '/goal_point']),
                    goal_area=_box(*z[key + '/goal_area']), allowed_goal_theta_difference=float(z[key + '/allowed_dtheta']),
                    obstacles=obstacles)


def intersection(turn_indicator: int, start_pos: int) -> Scenario:
    """envs/intersection.py:10"""
    return world('intersection/%d_%d' % (start_pos, turn_indicator))


def t_intersection(turn_indicator: int, start_pos: int) -> Scenario:
    """envs/t_intersection.py:10"""
    return world('t_intersection/%d_%d' % (start_pos, turn_indicator))


def roundabout(turn_indicator: int, start_pos: int) -> Scenario:
    """envs/roundabout.py:10"""
    return world('roundabout/%d_%d' % (start_pos, turn_indicator))


def roundabout_big(turn_indicator: int, start_pos: int) -> Scenario:
    """envs/roundabout_big.py:10 (the module's function is also called `roundabout`)"""
    return world('roundabout_big/%d_%d' % (start_pos, turn_indicator))


def intersection_multi_lanes(turn_indicator: int = 1, start_pos: int = 1, start_lane: int = 1, goal_lane: int = 1,
                             number_of_lanes: int = 1) -> Scenario:
    """envs/intersection_multi_lanes.py:9 (1..3 lanes tabulated)"""
    return world('intersection_multi_lanes/%d_%d_%d_%d_%d' % (start_pos, turn_indicator, start_lane, goal_lane, number_of_lanes))


class ArterialMultiLanes:
    """envs/arterial_multi_lanes.py:10-57: straight multi-lane road along +y between two pavements"""

    def __init__(self, num_lanes=2, goal_lane=1):
        self.num_lanes, self.goal_lane = num_lanes, goal_lane
        self.width_road, self.width_pavement, self.length = 4, 5, 100
        self.allowed_goal_theta_difference = np.pi / 16

    def create_scenario(self):
        n, w, wp, ln = self.num_lanes, self.width_road, self.width_pavement, self.length
        if n < 1 or self.goal_lane > n:
            return None
        left = - (n * w / 2) - (wp / 2)
        right = (n * w / 2) + (wp / 2)
        lane = (n // 2 - 0.5) * w - (self.goal_lane - 1) * w
        if n % 2 != 0:
            lane += w / 2
        start = (w * (n / 2 - 0.5), -ln / 2, np.pi / 2)
        goal = (lane, ln / 2, np.pi / 2)
        return Scenario(start=start, goal_point=goal, goal_area=BoxObstacle(xy_width=(w, w), height=1, xy_center=(goal[0], goal[1])),
                        allowed_goal_theta_difference=self.allowed_goal_theta_difference,
                        obstacles=[BoxObstacle(xy_width=(wp, ln), height=1, xy_center=(left, 0)),
                                   BoxObstacle(xy_width=(wp, ln), height=0.1, xy_center=(right, 0))])


def free_area(test_no=1, angle: float = 0.0, start_pos: float = 0.0, goal_distance=20, acceptable_error=np.pi / 16) -> Scenario:
    """envs/free_area.py:10-37: no obstacles, goal at `goal_distance` along `angle` (goal heading = angle for test 1, 0 for test 2)"""
    gx = start_pos + goal_distance * np.cos(angle)
    gy = start_pos + goal_distance * np.sin(angle)
    if test_no not in (1, 2):
        raise UnboundLocalError("free_area: test_no must be 1 or 2")
    goal = (gx, gy, angle if test_no == 1 else 0)
    return Scenario(start=(start_pos, start_pos, 0.), goal_point=goal,
                    goal_area=BoxObstacle(xy_width=(4 * 1.8, 4), height=0.5, xy_center=(goal[0], goal[1])),
                    allowed_goal_theta_difference=acceptable_error, obstacles=[])
