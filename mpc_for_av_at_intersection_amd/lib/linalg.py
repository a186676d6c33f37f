"""2-D rigid transform of (x, y[, theta]) rows (reference: main/lib/linalg.py:4-54); the matrix build is scalar host
code, the point transform runs through mpcx_transform_batch."""
import numpy as np

from ._session import context


def create_2d_transform_mtx(x: float, y: float, theta: float) -> np.ndarray:
    c, s = np.cos(theta), np.sin(theta)
    if x == 0 and y == 0:
        return np.array([[c, -s], [s, c]])
    return np.array([[c, -s, x], [s, c, y], [0, 0, 1]])


def transform_2d_pts(theta: float, transform_mtx: np.ndarray, points: np.ndarray) -> np.ndarray:
    if transform_mtx.shape == (3, 3):
        x, y = float(transform_mtx[0, 2]), float(transform_mtx[1, 2])
    elif transform_mtx.shape == (2, 2):
        x = y = 0.0
    else:
        raise RuntimeError()
    if points.shape[1] not in (2, 3):
        raise RuntimeError()
    ctx = context()
    pts3 = np.zeros((len(points), 3))
    pts3[:, :points.shape[1]] = points
    out = ctx.transform(ctx.f64([[x, y, theta]]), ctx.i32([0]), ctx.i32([len(points)]), ctx.f64(pts3), len(points)).cpu().numpy()[0]
    return out[:, :points.shape[1]]
