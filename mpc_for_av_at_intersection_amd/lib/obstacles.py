"""Static obstacles as half-plane sets (reference: main/lib/obstacles.py:47-176). `to_convex` / `distance_to_point`
are set-up-time scalar code; `check_collision` runs on the device through the expansion kernel (the same
half-plane test `neighbor_function` uses)."""
from typing import Tuple

import numpy as np


class Obstacle:
    hidden = False

    def to_convex(self, margin: float = 0.) -> np.ndarray:
        raise NotImplementedError

    def distance_to_point(self, point) -> float:
        raise NotImplementedError


class BoxObstacle(Obstacle):
    def __init__(self, xy_width, height: float, xy_center, hidden: bool = False):
        self.xy_width, self.height, self.xy_center, self.hidden = xy_width, height, xy_center, hidden
        cx, cy = xy_center
        wx, wy = xy_width
        self.xy1 = cx - wx / 2, cy - wy / 2
        self.xy2 = cx + wx / 2, cy + wy / 2

    def to_convex(self, margin: float = 0.) -> np.ndarray:
        (x1, y1), (x2, y2) = self.xy1, self.xy2
        # rows (a, b, c): inside <=> a*x + b*y + c <= 0 for all rows; order right, left, top, bottom (obstacles.py:87-90)
        return np.array([[1, 0, -(x2 + margin)], [-1, 0, x1 - margin], [0, 1, -(y2 + margin)], [0, -1, y1 - margin]])

    def distance_to_point(self, point) -> float:
        (x1, y1), (x2, y2) = self.xy1, self.xy2
        x, y = point
        dx = max(x1 - x, 0, x - x2)
        dy = max(y1 - y, 0, y - y2)
        return np.sqrt(dx * dx + dy * dy)


class CircleObstacle(Obstacle):
    def __init__(self, radius: float, height: float, xy_center, hidden: bool = False):
        self.radius, self.height, self.xy_center, self.hidden = radius, height, xy_center, hidden

    def to_convex(self, margin: float = 0.) -> np.ndarray:
        cx, cy = self.xy_center
        r = self.radius
        q, m2 = r * np.sqrt(2), 2 * margin   # the reference's (un-normalised) diagonal offset, obstacles.py:145-148;
        # subtracted one after the other, left to right, so the rows round exactly like the reference's
        return np.array([[1, 0, -(cx + r + margin)], [-1, 0, cx - r - margin], [0, 1, -(cy + r + margin)], [0, -1, cy - r - margin],
                         [-1, 1, cx - cy - q - m2], [1, -1, -cx + cy - q - m2], [-1, -1, cx + cy - q - m2], [1, 1, -cx - cy - q - m2]])

    def distance_to_point(self, point) -> float:
        px, py = point
        cx, cy = self.xy_center
        return max(0, np.sqrt((cx - px) ** 2 + (cy - py) ** 2) - self.radius)


def check_collision(obstacle_halfplanes: np.ndarray, points: np.ndarray) -> bool:
    """True if any point (columns of `points`, shape (2, N)) lies inside all half-planes (obstacles.py:157-176)."""
    import torch
    from ._session import context
    n_hp, n_hp_c = obstacle_halfplanes.shape
    assert n_hp_c == 3
    n_c, n_pts = points.shape
    assert n_c == 2
    if n_pts == 0:
        return False
    ctx = context()
    # identity pose (0, 0, 0) leaves the template untouched (linalg.py:13-17 rotation-only branch with theta = 0)
    model = ctx.search_model([np.ascontiguousarray(points.T, dtype=np.float64)], np.zeros((1, 3)), np.zeros(1),
                             np.asarray(obstacle_halfplanes, dtype=np.float64), np.array([0, n_hp], dtype=np.int32))
    out = ctx.expand(model, torch.zeros((1, 3), dtype=torch.float64, device=ctx.device))
    hit = bool(out['collide'].cpu()[0, 0])
    model.close()
    return hit
