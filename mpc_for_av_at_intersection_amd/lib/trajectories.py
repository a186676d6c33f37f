"""Path utilities used at set-up time and by scenario scripts (reference: main/lib/trajectories.py:11-126).
Host numpy; the per-step device versions live in csrc (ref_window_kernel, interaction_kernel)."""
from typing import List, Union

import numpy as np


def shift_car_trajectory_by_objspace_offset(trajectory: np.ndarray, x_offset: float, y_offset: float) -> np.ndarray:
    th = trajectory[:, 2]
    c, s = np.cos(th), np.sin(th)
    xy = np.vstack([c * x_offset - s * y_offset, s * x_offset + c * y_offset]).T
    xy += trajectory[:, :2]
    return np.append(xy, np.atleast_2d(th).T, axis=1)


def car_trajectory_to_collision_point_trajectories(trajectory: np.ndarray, car_dimensions) -> List[np.ndarray]:
    return [shift_car_trajectory_by_objspace_offset(trajectory, cc[0], cc[1]) for cc in car_dimensions.circle_centers]


def resample_curve(points: np.ndarray, dl: Union[float, np.ndarray], keep_last_point: bool = True) -> np.ndarray:
    """keep the points where floor(arc_length / dl) increments (plus the first, and optionally the last)"""
    assert 2 <= points.shape[1]
    seg = np.linalg.norm(points[1:, :2] - points[:-1, :2], axis=1)
    bucket = np.floor(np.append(0., seg).cumsum() / dl).astype(int)
    keep = np.append(True, (bucket[1:] - bucket[:-1]) >= 1.)
    if keep_last_point:
        keep[-1] = True
    return points[keep].copy()


def calc_nearest_index(state, cx: np.ndarray, cy: np.ndarray, start_index: int = 0) -> int:
    d = (cx[start_index:] - state.x) ** 2 + (cy[start_index:] - state.y) ** 2
    return int(np.argmin(d)) + start_index if len(d) > 0 else start_index


def calc_nearest_index_in_direction(state, cx: np.ndarray, cy: np.ndarray, start_index: int = 0, forward: bool = True) -> int:
    dist = np.linalg.norm([cx[start_index:] - state.x, cy[start_index:] - state.y], axis=0)
    n = len(dist)
    if n <= 1:
        return start_index
    if n == 2:
        return 1 + start_index if forward else start_index
    order = np.lexsort((np.arange(n), dist))[:3]      # three nearest, ties by lower index
    if abs(order[1] - order[2]) == 2:
        return int(order[0]) + start_index
    if abs(order[0] - order[1]) == 1:
        return int(max(order[0], order[1]) if forward else min(order[0], order[1])) + start_index
    raise Exception("something wrong")
