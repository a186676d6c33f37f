"""Ego-vs-moving-cars conflict search (reference: main/lib/collision_avoidance.py:66-119), GPU-backed.

Same signatures as the reference: the caller passes the resampled ego prediction, the detailed remaining path and
the predicted obstacle trajectories; the search for the first conflicting row and the earliest conflicting pose
runs in mpcx_moving_collision_batch (one problem)."""
from typing import List, Optional, Tuple

import numpy as np

from ..runtime import InteractionParams
from ._session import context


def _cs(yaw):
    return np.column_stack([np.cos(yaw), np.sin(yaw)])


def check_collision_moving_cars(car_dimensions, traj_agent: np.ndarray, path_agent_detailed: np.ndarray,
                                traj_obstacles: List[np.ndarray], frame_window: int = 0) -> Optional[Tuple[float, float, float]]:
    if len(traj_obstacles) == 0:
        return None
    ctx = context()
    steps = len(traj_obstacles[0])
    if any(len(t) != steps for t in traj_obstacles):
        raise ValueError('all obstacle trajectories must have the same number of frames')
    centers = np.asarray(car_dimensions.circle_centers, dtype=np.float64)
    if centers.shape != (2, 2):
        raise ValueError('the device kernel expects two collision discs per car')
    ip = InteractionParams(pred_steps=steps, frame_window=int(frame_window), cutoff_margin=0,
                           L=car_dimensions.distance_back_to_front_wheel, radius=car_dimensions.radius,
                           circle_centers=centers.ravel())
    ego = np.ascontiguousarray(traj_agent[:, :3], dtype=np.float64)
    path = np.ascontiguousarray(path_agent_detailed[:, :3], dtype=np.float64)
    obs = np.ascontiguousarray(np.stack([np.asarray(t)[:, :3] for t in traj_obstacles]), dtype=np.float64)
    i0 = ctx.i32([0])
    out = ctx.moving_collision(ip, ctx.f64(ego), ctx.f64(_cs(ego[:, 2])), i0, ctx.i32([len(ego)]),
                               ctx.f64(path), ctx.f64(_cs(path[:, 2])), i0, ctx.i32([len(path)]),
                               ctx.f64(obs.reshape(-1, 3)), ctx.f64(_cs(obs.reshape(-1, 3)[:, 2])), i0, ctx.i32([len(traj_obstacles)]))
    ctx.synchronize()
    idx = int(out['hit_idx'].cpu()[0])
    if idx == -2:
        raise ValueError('trajectory sizes exceed the device limits (see include/mpcx.h)')
    if idx < 0:
        return None
    x, y = path_agent_detailed[idx, :2]
    return x, y, idx


def get_cutoff_curve_by_position_idx(points: np.ndarray, x: float, y: float, radius: float = 0.001):
    ctx = context()
    pts = np.ascontiguousarray(points[:, :2], dtype=np.float64)
    pts3 = np.column_stack([pts, np.zeros(len(pts))])
    idx = int(ctx.cutoff_index(ctx.f64(pts3), ctx.i32([0]), ctx.i32([len(pts)]), ctx.f64([[x, y]]), radius).cpu()[0])
    if idx < 0:
        return points        # the reference returns its input when nothing matches (collision_avoidance.py:115-117)
    return idx
