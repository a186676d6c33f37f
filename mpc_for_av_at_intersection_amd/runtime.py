"""Thin device runtime over the C ABI (include/mpcx.h): torch-ROCm tensors own the device buffers, every
computation happens in libmpcx.so's HIP kernels.  No CPU fallback exists: constructing a Context without a
GPU or without the built library raises."""
import ctypes as C
import functools
from dataclasses import dataclass, field
from typing import Tuple, Optional, Sequence

import numpy as np
import torch

from . import _lib


@dataclass
class MpcParams:
    """main/config/mpc_config.json + lib/mpc.py:17-36 + lib/simulation.py:23-25 of the reference."""
    T: int = 13
    dt: float = 0.2
    L: float = 2.86
    w_perp: float = 20.0
    w_para: float = 1.0
    R: Sequence[float] = (0.01, 0.01)
    Rd: Sequence[float] = (0.01, 1.0)
    Q_v_yaw: Sequence[float] = (0.0, 0.5)
    Qf_base: Sequence[float] = (1.0, 1.0, 0.0, 0.5)   # multiplied by T (mpc.py:25)
    R_end: Sequence[float] = (10.0, 10.0)
    max_speed: float = 30.0 / 3.6
    min_speed: float = -5.0
    max_accel: float = 2.0
    max_decel: float = -10.0
    max_steer: float = float(np.deg2rad(45.0))
    max_dsteer: float = float(np.deg2rad(30.0))
    max_iter: int = 60
    tol: float = 1e-10
    model: int = _lib.MODEL_BICYCLE4     # _lib.MODEL_JERK5: the five-state problem of lib/mpc_jerk.py
    jerk_weight: float = 1.0             # jerk_penalty_weight, mpc_jerk.py:30

    @classmethod
    def jerk(cls, **kw) -> 'MpcParams':
        """the constants of main/lib/mpc_jerk.py:16-39: T 13, cross-track weight 10 (line 167), Rd (0.3, 1), MAX_DECEL -5"""
        base = dict(T=13, w_perp=10.0, w_para=1.0, R=(0.01, 0.01), Rd=(0.3, 1.0), Q_v_yaw=(0.0, 0.5),
                    Qf_base=(1.0, 1.0, 0.0, 0.5), max_accel=2.0, max_decel=-5.0, model=_lib.MODEL_JERK5, jerk_weight=1.0)
        base.update(kw)
        return cls(**base)

    def to_c(self) -> _lib.MpcParamsC:
        p = _lib.MpcParamsC()
        p.T, p.max_iter, p.dt, p.L = int(self.T), int(self.max_iter), float(self.dt), float(self.L)
        p.w_perp, p.w_para = float(self.w_perp), float(self.w_para)
        p.R[:] = list(map(float, self.R)); p.Rd[:] = list(map(float, self.Rd))
        p.Q_v_yaw[:] = list(map(float, self.Q_v_yaw))
        p.Qf[:] = [float(q) * self.T for q in self.Qf_base]
        p.R_end[:] = list(map(float, self.R_end))
        p.max_speed, p.min_speed = float(self.max_speed), float(self.min_speed)
        p.max_accel, p.max_decel = float(self.max_accel), float(self.max_decel)
        p.max_steer, p.max_dsteer, p.tol = float(self.max_steer), float(self.max_dsteer), float(self.tol)
        p.model, p.reserved, p.jerk_weight = int(self.model), 0, float(self.jerk_weight)
        return p

    def tuning_row(self) -> np.ndarray:
        """the 16 doubles of mpcx_qp_tuning (include/mpcx.h) for this parameter set"""
        return np.array([self.w_perp, self.w_para, *self.R, *self.Rd, *self.Q_v_yaw, *[float(q) * self.T for q in self.Qf_base],
                         self.max_accel, self.max_decel, self.max_dsteer, 0.0], dtype=np.float64)


@dataclass
class InteractionParams:
    """scenario constants of main/scenarios/mpc_intersection.py:31,81-84 + car_dimensions.py"""
    pred_steps: int = 35
    frame_window: int = 20
    cutoff_margin: int = 72
    dt: float = 0.2
    L: float = 2.86
    radius: float = 2.0 / 2 ** 0.5
    circle_centers: Sequence[float] = (2.18, 0.0, 0.68, 0.0)
    max_accel: float = 2.0
    max_speed: float = 30.0 / 3.6
    max_path_len: int = 0        # longest path of the batch in points (0 = the kernel's default capacity of 1024)
    path_cum: Optional[torch.Tensor] = None    # per point of the path table: arc length from the start of its path (device, float64) ...
    path_cum_err: float = 0.0                  # ... and the bound on the error of its differences (see mpcx_interaction_params; `path_tables()`)
    path_first_within: Optional[torch.Tensor] = None   # per point: first point of its path within 1 mm of it (device, int32; `path_first_within()`)
    plan: Optional[dict] = None                # ego prediction per path point (`path_plan()` uploaded: cnt, disc, box, path_disc tensors + cap, steps, dl, radius)

    def to_c(self) -> _lib.InteractionParamsC:
        p = _lib.InteractionParamsC()
        p.pred_steps, p.frame_window, p.cutoff_margin, p.max_path_len = int(self.pred_steps), int(self.frame_window), int(self.cutoff_margin), int(self.max_path_len)
        p.dt, p.L, p.radius = float(self.dt), float(self.L), float(self.radius)
        cc = list(map(float, np.asarray(self.circle_centers, dtype=np.float64).ravel()))
        if len(cc) == 2:
            # car_dimensions.py:51-75 with skip_back_circle_collision_checking=True: ONE disc.  The kernels test two; the same disc twice
            # gives the reference's one-disc answers (every test is a min / first-hit over the discs, and the frame index is taken
            # modulo the path length, collision_avoidance.py:92-98)
            cc = cc + cc
        if len(cc) != 4:
            raise MpcxError('car_dimensions.circle_centers must hold one or two discs (x, y offsets), got %d numbers' % len(cc))
        p.circle_centers[:] = cc
        p.max_accel, p.max_speed = float(self.max_accel), float(self.max_speed)
        p.path_cum = None if self.path_cum is None else C.c_void_p(self.path_cum.data_ptr())
        p.path_cum_err = float(self.path_cum_err) if self.path_cum is not None else 0.0
        p.path_first_within = None if self.path_first_within is None else C.c_void_p(self.path_first_within.data_ptr())
        if self.plan is not None and self.path_cum is not None:
            pl = self.plan
            p.plan_cnt, p.plan_disc, p.plan_box, p.path_disc = (C.c_void_p(pl[k].data_ptr()) for k in ('cnt', 'disc', 'box', 'path_disc'))
            p.plan_cap, p.plan_steps, p.plan_dl, p.plan_radius = int(pl['cap']), int(pl['steps']), float(pl['dl']), float(pl['radius'])
        return p


class MpcxError(RuntimeError):
    pass


def path_tables(table: np.ndarray, offs) -> Tuple[np.ndarray, float]:
    """Arc-length table for mpcx_interaction_params.path_cum: for the concatenated path table `table` ((n, >= 2): x, y, ...) whose
    paths start at offs[0], offs[1], ... (offs[-1] = n), per point the running sum of the step lengths of ITS path (np.cumsum, restarted
    per path) and the bound on the distance between a difference of two entries and the reference's own running sum over the same steps
    (trajectories.py:72-79): both are sequential float64 sums of at most n_max non-negative terms adding up to at most L, each within
    (n_max - 1) * 2^-53 * L of the exact sum (first-order bound, doubled here), plus one rounding of the subtraction."""
    table = np.asarray(table, dtype=np.float64)
    cum = np.zeros(len(table))
    worst = 0.0
    for a, b in zip(offs[:-1], offs[1:]):
        if b - a < 1:
            continue
        d = table[a + 1:b, :2] - table[a:b - 1, :2]
        steps = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
        cum[a + 1:b] = np.cumsum(steps)
        n, length = b - a, float(cum[b - 1]) if b - a > 1 else 0.0
        worst = max(worst, 2.0 * (3 * n + 2) * 2.0 ** -53 * length)
    return cum, worst


def path_first_within(table: np.ndarray, offs, radius: float = 0.001) -> np.ndarray:
    """mpcx_interaction_params.path_first_within: for every point k of the concatenated path table (paths start at offs[0], offs[1], ...,
    offs[-1] = n) the index, relative to its path's first point, of the first point j <= k of the same path with
    np.linalg.norm(p_j - p_k) <= radius -- what get_cutoff_curve_by_position_idx(path, *p_k[:2]) returns (collision_avoidance.py:107-119),
    with the reference's own numpy expression."""
    table = np.asarray(table, dtype=np.float64)
    out = np.zeros(len(table), dtype=np.int32)
    for a, b in zip(offs[:-1], offs[1:]):
        pts = table[a:b, :2]
        for k0 in range(0, b - a, 256):                     # (k, j) blocks of 256 x n: bounded memory for long paths
            k1 = min(k0 + 256, b - a)
            diff = pts[None, :, :] - pts[k0:k1, None, :]      # points_diff[:, 0] -= x; points_diff[:, 1] -= y
            near = np.linalg.norm(diff, axis=2) <= radius
            out[a + k0:a + k1] = np.argmax(near, axis=1)     # first True; the point itself is always one
    return out


def path_plan(table: np.ndarray, cs: np.ndarray, offs, dt: float, max_speed: float, circle_centers, radius: float, pred_steps: int,
              cap: int = 64, n_runs: int = 8) -> dict:
    """mpcx_interaction_params.plan_*: for every point t of the concatenated path table the ego prediction the scenario loop builds from
    trajectory_full[t:] once the predicted speed has saturated (mpc_intersection.py:107-116: resample_curve with dl = DT * MAX_SPEED),
    evaluated with the reference's own numpy expressions: which poses are kept (trajectories.py:58-86, np.cumsum restarted at t), their
    disc centres (trajectories.py:11-37 with the host's cos / sin of the yaw column) and, for the conflict search's cull, the bounding
    boxes of `n_runs` runs of frames of the padded prediction (max(kept, pred_steps) frames), inflated by a little more than 2 * radius.
    Returns numpy arrays cnt (npts,), disc (npts, cap, 4), box (npts, n_runs, 4), path_disc (npts, 4) + cap, steps, dl, radius."""
    table = np.asarray(table, dtype=np.float64)
    cs = np.asarray(cs, dtype=np.float64)
    npts = len(table)
    cc = np.asarray(circle_centers, dtype=np.float64).reshape(-1, 2)
    if len(cc) == 1:
        cc = np.repeat(cc, 2, axis=0)
    c, s = cs[:, 0], cs[:, 1]
    path_disc = np.empty((npts, 4))
    for d in range(2):
        path_disc[:, 2 * d] = (c * cc[d, 0] - s * cc[d, 1]) + table[:, 0]
        path_disc[:, 2 * d + 1] = (s * cc[d, 0] + c * cc[d, 1]) + table[:, 1]
    dl = dt * max_speed
    md = 2.0 * radius
    slack = md * (1.0 + 1e-9) + 1e-9
    cnt = np.zeros(npts, dtype=np.int32)
    disc = np.zeros((npts, cap, 4))
    box = np.empty((npts, n_runs, 4))
    box[:, :, 0::2] = np.inf; box[:, :, 1::2] = -np.inf
    for a, b in zip(offs[:-1], offs[1:]):
        if b - a < 1:
            continue
        seg = np.append(0.0, np.linalg.norm(table[a + 1:b, :2] - table[a:b - 1, :2], axis=1))      # seg[i]: step into point i of the path
        for t in range(a, b):
            steps = seg[t - a:].copy()
            steps[0] = 0.0                                  # np.append(0., norm(diff)) of the sub-path starting at t
            bucket = np.floor(steps.cumsum() / dl).astype(int)
            keep = np.append(True, (bucket[1:] - bucket[:-1]) >= 1.)
            keep[-1] = True
            idx = np.nonzero(keep)[0]
            k = len(idx)
            if k > cap:
                continue                                    # not tabulated: the kernel resamples
            cnt[t] = k
            ego = path_disc[t + idx]
            disc[t, :k] = ego
            F = max(k, pred_steps)
            SL = (F + n_runs - 1) // n_runs
            padded = ego[np.minimum(np.arange(F), k - 1)]   # frames beyond the prediction repeat its last pose
            for sg in range(n_runs):
                run = padded[sg * SL:(sg + 1) * SL]
                if len(run):
                    xs, ys = run[:, 0::2], run[:, 1::2]
                    box[t, sg] = (xs.min() - slack, xs.max() + slack, ys.min() - slack, ys.max() + slack)
    return dict(cnt=cnt, disc=disc, box=box, path_disc=path_disc, cap=cap, steps=int(pred_steps), dl=dl, radius=float(radius))


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _ordered(fn):
    """Every launch goes to the CONTEXT's stream.  When that is not torch's current stream (a Context created on a stream of its
    own while the caller's tensors are produced and consumed on another), the call is ordered on both sides: the context's stream
    first waits for what the current stream has queued (the inputs), and the current stream afterwards waits for the launch (the
    outputs; that also keeps the caching allocator from handing a buffer back while a kernel still uses it).  Two event
    operations, and nothing at all in the default case of one shared stream."""
    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        cur = torch.cuda.current_stream(self.device)
        if cur == self.stream:
            return fn(self, *a, **k)
        self.stream.wait_stream(cur)
        try:
            return fn(self, *a, **k)
        finally:
            cur.wait_stream(self.stream)
    return wrapped


class Context:
    """One mpcx_ctx bound to a torch device and (by default) torch's current stream on it.  A Context on a stream of its own may be
    used from any current stream: see _ordered."""

    def __init__(self, device: int = 0, stream: Optional[torch.cuda.Stream] = None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise MpcxError('no GPU visible: the mpcx hot path is HIP-only (no CPU fallback)')
        self.device = torch.device('cuda', device)
        torch.cuda.set_device(self.device)
        self.stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        self._ctx = self.lib.mpcx_create(device, C.c_void_p(self.stream.cuda_stream))
        if not self._ctx:
            raise MpcxError('mpcx_create failed for device %d' % device)
        self.params: Optional[MpcParams] = None

    def close(self):
        if getattr(self, '_ctx', None):
            self.lib.mpcx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise MpcxError('mpcx error %d: %s' % (rc, self.lib.mpcx_last_error(self._ctx).decode()))

    # ------------------------------------------------------------------ helpers
    def f64(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

    def i32(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)

    def u8(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.uint8)).to(self.device)

    def _want(self, t, dtype, shape=None, name='tensor'):
        if t.device != self.device or t.dtype != dtype or not t.is_contiguous():
            raise MpcxError('%s must be a contiguous %s tensor on %s' % (name, dtype, self.device))
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise MpcxError('%s has shape %s, expected %s' % (name, tuple(t.shape), tuple(shape)))
        return t

    # ------------------------------------------------------------------ entry points
    def set_mpc_params(self, p: MpcParams):
        cp = p.to_c()
        self._chk(self.lib.mpcx_set_mpc_params(self._ctx, C.byref(cp)))
        self.params = p

    @_ordered
    def qp_solve(self, x0, xref, xbar, reaches_end, u_warm=None, out=None):
        """mpcx_qp_solve_batch. Returns dict(x (B,4,T+1), u (B,2,T), status, iters, kkt)."""
        T = self.params.T
        B = x0.shape[0]
        f = torch.float64
        self._want(x0, f, (B, 4), 'x0'); self._want(xref, f, (B, 4, T + 1), 'xref')
        self._want(xbar, f, (B, 4, T + 1), 'xbar'); self._want(reaches_end, torch.uint8, (B, T + 1), 'reaches_end')
        if u_warm is not None:
            self._want(u_warm, f, (B, 2, T), 'u_warm')
        if out is None:
            out = dict(x=torch.empty((B, 4, T + 1), dtype=f, device=self.device),
                       u=torch.empty((B, 2, T), dtype=f, device=self.device),
                       status=torch.empty(B, dtype=torch.int32, device=self.device),
                       iters=torch.empty(B, dtype=torch.int32, device=self.device),
                       kkt=torch.empty((B, 4), dtype=f, device=self.device))
        self._chk(self.lib.mpcx_qp_solve_batch(self._ctx, B, _ptr(x0), _ptr(xref), _ptr(xbar), _ptr(reaches_end),
                                               _ptr(u_warm), _ptr(out['x']), _ptr(out['u']), _ptr(out['status']),
                                               _ptr(out['iters']), _ptr(out['kkt'])))
        return out

    @_ordered
    def prepare(self, state, u_warm, path, path_off, path_len, dl, target_ind, out=None, path_v=None, x_prev=None):
        """mpcx_mpc_prepare_batch. target_ind is updated in place. Returns dict(xref, reaches_end, xbar).
        x_prev (B, 4, T+1): the previous linearisation pass's states -- its speeds space the reference window (lib/mpc.py:226-237,
        MAX_ITER > 1; mpcx_mpc_prepare_batch_ov)."""
        T = self.params.T
        B = state.shape[0]
        f = torch.float64
        self._want(state, f, (B, 4), 'state'); self._want(path, f, None, 'path')
        self._want(path_off, torch.int32, (B,), 'path_off'); self._want(path_len, torch.int32, (B,), 'path_len')
        self._want(target_ind, torch.int32, (B,), 'target_ind')
        if u_warm is not None:
            self._want(u_warm, f, (B, 2, T), 'u_warm')
        if out is None:
            out = dict(xref=torch.empty((B, 4, T + 1), dtype=f, device=self.device),
                       reaches_end=torch.empty((B, T + 1), dtype=torch.uint8, device=self.device),
                       xbar=torch.empty((B, 4, T + 1), dtype=f, device=self.device))
        ov, stride = None, 0
        if x_prev is not None:
            self._want(x_prev, f, (B, 4, T + 1), 'x_prev')
            ov, stride = x_prev.data_ptr() + 2 * (T + 1) * 8, 4 * (T + 1)
        self._chk(self.lib.mpcx_mpc_prepare_batch_ov(self._ctx, B, _ptr(state), _ptr(u_warm), _ptr(path), _ptr(path_v), _ptr(path_off),
                                                     _ptr(path_len), C.c_double(float(dl)), _ptr(target_ind), ov, stride,
                                                     _ptr(out['xref']), _ptr(out['reaches_end']), _ptr(out['xbar'])))
        return out

    @_ordered
    def plant_step(self, state, u, status, applied):
        B = state.shape[0]
        self._want(state, torch.float64, (B, 4), 'state'); self._want(applied, torch.float64, (B, 2), 'applied')
        self._chk(self.lib.mpcx_plant_step_batch(self._ctx, B, _ptr(state), _ptr(u), _ptr(status), _ptr(applied)))

    def search_model(self, templates, last_pose, edge_cost, hp, hp_off):
        return SearchModel(self, templates, last_pose, edge_cost, hp, hp_off)

    @_ordered
    def expand(self, model: 'SearchModel', nodes, out=None, nodes_cs=None):
        """mpcx_expand_batch: nodes (n,3) -> dict(nbr (n,P,3), cost (n,P), collide (n,P) uint8).
        nodes_cs (n,2): optional host-computed cos/sin of the headings (bit-exact replay of the reference's search)."""
        n = nodes.shape[0]
        self._want(nodes, torch.float64, (n, 3), 'nodes')
        if nodes_cs is not None:
            self._want(nodes_cs, torch.float64, (n, 2), 'nodes_cs')
        Pn = model.n_prim
        if out is None:
            out = dict(nbr=torch.empty((n, Pn, 3), dtype=torch.float64, device=self.device),
                       cost=torch.empty((n, Pn), dtype=torch.float64, device=self.device),
                       collide=torch.empty((n, Pn), dtype=torch.uint8, device=self.device))
        self._chk(self.lib.mpcx_expand_batch(self._ctx, model._h, n, _ptr(nodes), _ptr(nodes_cs), _ptr(out['nbr']), _ptr(out['cost']),
                                             _ptr(out['collide'])))
        return out

    @_ordered
    def astar_batch(self, models, specs, cs_theta, cs_val, ov_key=None, ov_val=None, max_expansions=4096, path_cap=64, heap_cap=None, push_cap=None):
        """mpcx_astar_batch: one device-resident best-first search per entry of `specs` (dicts: start, goal_box, goal_point,
        allowed_dtheta, variant [, wh, wc, hp_norm (device tensor), ov_off, ov_cnt]) against models[i]; cs_theta (ascending) / cs_val: the
        host's cos / sin table; ov_key (n, 4) / ov_val: the override table, every search's slice sorted as tuples.  Capacities: the log
        holds max_expansions expansions, the heap `heap_cap` entries and the successor log `push_cap` (default for both: what n_prim
        successors per expansion can need).  Returns a dict of device tensors: status, n_exp, n_push, cost, miss, path_len, path
        (n, path_cap, 3; goal first), path_prim, log (n, max_expansions, 8), push_log (n, push_cap, 8)."""
        n = len(specs)
        f, dev = torch.float64, self.device
        P = max(m.n_prim for m in models)
        full = max(16, P * max_expansions + 1)
        heap_cap = full if heap_cap is None else max(16, min(int(heap_cap), full))
        push_cap = full if push_cap is None else max(16, min(int(push_cap), full))
        table_cap = 1 << int(np.ceil(np.log2(2 * (max_expansions + 1))))
        # only the closed set needs a fill (NaN = empty slot); everything else is written by the kernel before anyone reads it
        out = dict(heap=torch.empty((n, heap_cap, 10), dtype=f, device=dev), table=torch.full((n, table_cap, 8), float('nan'), dtype=f, device=dev),
                   log=torch.empty((n, max_expansions, 8), dtype=f, device=dev), push_log=torch.empty((n, push_cap, 8), dtype=f, device=dev),
                   path=torch.empty((n, path_cap, 3), dtype=f, device=dev), path_prim=torch.empty((n, path_cap), dtype=torch.int32, device=dev),
                   cost=torch.empty(n, dtype=f, device=dev), miss=torch.zeros(n, dtype=f, device=dev),
                   status=torch.empty(n, dtype=torch.int32, device=dev), n_exp=torch.empty(n, dtype=torch.int32, device=dev),
                   n_push=torch.empty(n, dtype=torch.int32, device=dev), path_len=torch.empty(n, dtype=torch.int32, device=dev))
        b = _lib.AstarBuffersC()
        b.heap_cap, b.table_cap, b.log_cap, b.push_cap, b.path_cap = heap_cap, table_cap, max_expansions, push_cap, path_cap
        for k in ('heap', 'table', 'log', 'push_log', 'path', 'cost', 'miss', 'status', 'n_exp', 'n_push', 'path_len', 'path_prim'):
            setattr(b, k, out[k].data_ptr())
        if isinstance(specs, np.ndarray):       # rows of _lib.ASTAR_SEARCH_DTYPE, filled by the caller (hp_norm: device addresses it keeps alive)
            if specs.dtype != _lib.ASTAR_SEARCH_DTYPE:
                raise MpcxError('astar_batch: spec rows must have dtype _lib.ASTAR_SEARCH_DTYPE')
            rows = np.ascontiguousarray(specs)
        else:
            rows = np.zeros(n, dtype=_lib.ASTAR_SEARCH_DTYPE)
            for i, s in enumerate(specs):
                r = rows[i]
                r['start'] = s['start']; r['goal_box'] = s['goal_box']; r['goal_point'] = s['goal_point']
                r['allowed_dtheta'] = s['allowed_dtheta']
                r['wh'] = s.get('wh', (1.0, 2.7, 15.0, 0.0, 0.0)); r['wc'] = s.get('wc', (1.0, 5.0, 0.1, 0.0))
                hn = s.get('hp_norm')
                if hn is not None:
                    self._want(hn, f, (models[i].n_rows,), 'hp_norm')
                    r['hp_norm'] = hn.data_ptr()
                r['variant'] = s['variant']; r['ov_off'] = s.get('ov_off', 0); r['ov_cnt'] = s.get('ov_cnt', 0)
        rows['max_expansions'] = int(max_expansions)
        sp = C.c_void_p(rows.ctypes.data)
        hs = (C.c_void_p * n)(*[m._h for m in models])
        self._want(cs_theta, f, None, 'cs_theta'); self._want(cs_val, f, (cs_theta.shape[0], 2), 'cs_val')
        n_ov = 0 if ov_key is None else int(ov_key.shape[0])
        if n_ov:
            self._want(ov_key, f, (n_ov, 4), 'ov_key'); self._want(ov_val, f, (n_ov,), 'ov_val')
        self._chk(self.lib.mpcx_astar_batch(self._ctx, n, hs, sp, int(cs_theta.shape[0]), _ptr(cs_theta), _ptr(cs_val),
                                            n_ov, _ptr(ov_key) if n_ov else None, _ptr(ov_val) if n_ov else None, C.byref(b)))
        del out['heap'], out['table']
        return out

    @_ordered
    def expand_multi(self, models, seg_off, nodes, nodes_cs=None):
        """mpcx_expand_multi_batch: nodes (n,3) grouped by search, seg_off = n_seg+1 offsets (host ints), models = one
        SearchModel per segment.  Returns dict(nbr (n,P,3), cost (n,P), collide (n,P)) laid out like separate expand() calls."""
        n = nodes.shape[0]
        self._want(nodes, torch.float64, (n, 3), 'nodes')
        if nodes_cs is not None:
            self._want(nodes_cs, torch.float64, (n, 2), 'nodes_cs')
        Pn = models[0].n_prim
        out = dict(nbr=torch.empty((n, Pn, 3), dtype=torch.float64, device=self.device),
                   cost=torch.empty((n, Pn), dtype=torch.float64, device=self.device),
                   collide=torch.empty((n, Pn), dtype=torch.uint8, device=self.device))
        handles = (C.c_void_p * len(models))(*[m._h for m in models])
        offs = (C.c_int32 * (len(models) + 1))(*[int(v) for v in seg_off])
        self._chk(self.lib.mpcx_expand_multi_batch(self._ctx, len(models), handles, offs, _ptr(nodes), _ptr(nodes_cs), _ptr(out['nbr']),
                                                   _ptr(out['cost']), _ptr(out['collide'])))
        return out

    @_ordered
    def interaction(self, ip: InteractionParams, state, path, path_cs, path_off, path_len, prev_cut_len,
                    obs6, obs_off, obs_cnt, obs_skip, traj_idx, out=None):
        """mpcx_interaction_batch. traj_idx updated in place. Returns dict(hit_idx, hit_xy, cut_len)."""
        Pn = state.shape[0]
        i32 = torch.int32
        self._want(state, torch.float64, (Pn, 4), 'state')
        self._want(path_cs, torch.float64, (path.shape[0], 2), 'path_cs')
        for nm, t in (('path_off', path_off), ('path_len', path_len), ('obs_off', obs_off), ('obs_cnt', obs_cnt), ('traj_idx', traj_idx)):
            self._want(t, i32, (Pn,), nm)
        if ip.path_cum is not None:
            self._want(ip.path_cum, torch.float64, (path.shape[0],), 'path_cum')
        if ip.path_first_within is not None:
            self._want(ip.path_first_within, torch.int32, (path.shape[0],), 'path_first_within')
        nobs = 0 if obs6 is None else obs6.shape[0]
        if out is None:
            out = dict(hit_idx=torch.empty(Pn, dtype=i32, device=self.device),
                       hit_xy=torch.empty((Pn, 2), dtype=torch.float64, device=self.device),
                       cut_len=torch.empty(Pn, dtype=i32, device=self.device))
        cip = ip.to_c()
        self._chk(self.lib.mpcx_interaction_batch(self._ctx, C.byref(cip), Pn, _ptr(state), _ptr(path), _ptr(path_cs),
                                                  _ptr(path_off), _ptr(path_len), _ptr(prev_cut_len), nobs, _ptr(obs6),
                                                  _ptr(obs_off), _ptr(obs_cnt), _ptr(obs_skip), _ptr(traj_idx),
                                                  _ptr(out['hit_idx']), _ptr(out['hit_xy']), _ptr(out['cut_len'])))
        return out

    @_ordered
    def moving_collision(self, ip: InteractionParams, ego, ego_cs, ego_off, ego_len, path, path_cs, path_off, path_len,
                         obs, obs_cs, obs_off, obs_cnt, out=None):
        """mpcx_moving_collision_batch (explicit trajectories). Returns dict(hit_idx, hit_xy)."""
        Pn = ego_off.shape[0]
        if out is None:
            out = dict(hit_idx=torch.empty(Pn, dtype=torch.int32, device=self.device),
                       hit_xy=torch.empty((Pn, 2), dtype=torch.float64, device=self.device))
        nobs = 0 if obs is None else obs.shape[0]
        cip = ip.to_c()
        self._chk(self.lib.mpcx_moving_collision_batch(self._ctx, C.byref(cip), Pn, _ptr(ego), _ptr(ego_cs), _ptr(ego_off),
                                                       _ptr(ego_len), _ptr(path), _ptr(path_cs), _ptr(path_off), _ptr(path_len),
                                                       nobs, _ptr(obs), _ptr(obs_cs), _ptr(obs_off), _ptr(obs_cnt),
                                                       _ptr(out['hit_idx']), _ptr(out['hit_xy'])))
        return out

    @_ordered
    def transform(self, nodes, pts_off, pts_cnt, pts, max_pts):
        """mpcx_transform_batch -> (n, max_pts, 3)"""
        n = nodes.shape[0]
        out = torch.empty((n, max_pts, 3), dtype=torch.float64, device=self.device)
        self._chk(self.lib.mpcx_transform_batch(self._ctx, n, int(max_pts), _ptr(nodes), _ptr(pts_off), _ptr(pts_cnt), _ptr(pts), _ptr(out)))
        return out

    @_ordered
    def cutoff_index(self, pts, off, ln, xy, radius=0.001):
        Pn = off.shape[0]
        out = torch.empty(Pn, dtype=torch.int32, device=self.device)
        self._chk(self.lib.mpcx_cutoff_index_batch(self._ctx, Pn, _ptr(pts), _ptr(off), _ptr(ln), _ptr(xy), C.c_double(radius), _ptr(out)))
        return out

    @_ordered
    def predict_obstacles(self, obs6, steps, dt, L):
        n = obs6.shape[0]
        out = torch.empty((n, steps, 3), dtype=torch.float64, device=self.device)
        self._chk(self.lib.mpcx_predict_obstacles_batch(self._ctx, n, int(steps), C.c_double(dt), C.c_double(L), _ptr(obs6), _ptr(out)))
        return out

    @_ordered
    def closed_loop_run(self, ip: InteractionParams, desc: '_lib.ClosedLoopC', n_steps: int, graph: bool = False):
        """mpcx_closed_loop_run: n_steps of the scenario loop body on the buffers `desc` names, no host work between."""
        cip = ip.to_c()
        self._chk(self.lib.mpcx_closed_loop_run(self._ctx, C.byref(cip), C.byref(desc), int(n_steps), 1 if graph else 0))

    @_ordered
    def closed_loop_stats(self, reset: bool = True):
        """mpcx_closed_loop_stats: dict(agent_steps, iterations, failures, max_iterations) accumulated on the device by
        closed_loop_run since the last reset (synchronises)"""
        out = (C.c_int64 * 4)()
        self._chk(self.lib.mpcx_closed_loop_stats(self._ctx, out, 1 if reset else 0))
        return dict(agent_steps=int(out[0]), iterations=int(out[1]), failures=int(out[2]), max_iterations=int(out[3]))

    @_ordered
    def set_instance_tuning(self, rows: Optional[torch.Tensor]):
        """mpcx_set_instance_tuning: rows (B,16) float64 device tensor (MpcParams.tuning_row per problem) or None to clear.
        The tensor is kept alive by the context while set."""
        if rows is None:
            self._chk(self.lib.mpcx_set_instance_tuning(self._ctx, None, 0))
        else:
            self._want(rows, torch.float64, (rows.shape[0], 16), 'tuning rows')
            self._chk(self.lib.mpcx_set_instance_tuning(self._ctx, _ptr(rows), int(rows.shape[0])))
        self._tuning = rows

    @_ordered
    def set_qp_order_hint(self, prev_iters: Optional[torch.Tensor], ref_now: Optional[torch.Tensor] = None,
                          ref_prev: Optional[torch.Tensor] = None):
        """mpcx_qp_set_order_hint: int32 device tensors -- the previous solve's iteration counts (may be the `iters` output) and,
        optionally, a pair whose inequality marks problems whose reference jumped (e.g. cut lengths now / before); None clears"""
        for t in (prev_iters, ref_now, ref_prev):
            if t is not None:
                self._want(t, torch.int32, None, 'order hint')
        self._chk(self.lib.mpcx_qp_set_order_hint(self._ctx, _ptr(prev_iters), _ptr(ref_now), _ptr(ref_prev)))
        self._order_hint = (prev_iters, ref_now, ref_prev)

    def set_qp_solver(self, which: str):
        """'auto', 'condensed' (one wavefront per QP) or 'stage' (stage-structured solver, eight lanes per QP)"""
        self._chk(self.lib.mpcx_set_qp_solver(self._ctx, {'auto': 0, 'condensed': 1, 'stage': 2}[which]))

    def set_linearisation_passes(self, passes: int):
        """lib/mpc.py MAX_ITER inside mpcx_closed_loop_run: (window, rollout, QP) passes per step (stock configuration: 1)"""
        self._chk(self.lib.mpcx_set_linearisation_passes(self._ctx, int(passes)))
        self.lin_passes = int(passes)

    def profile_qp(self, enable: bool):
        """bracket every qp_kernel launch with HIP events on the context's stream (mpcx_profile_qp)"""
        self._chk(self.lib.mpcx_profile_qp(self._ctx, 1 if enable else 0))

    @_ordered
    def profile_qp_read(self):
        """(summed milliseconds, launches) of the bracketed qp_kernel launches since the last read"""
        ms, n = C.c_double(0.0), C.c_int32(0)
        self._chk(self.lib.mpcx_profile_qp_read(self._ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ------------------------------------------------------------------ multi-GPU exchange (RCCL over xGMI)
    def comm_unique_id(self) -> bytes:
        """mpcx_comm_unique_id: rank 0 creates the id, the caller distributes it (sharding.init_comm broadcasts it)"""
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        rc = self.lib.mpcx_comm_unique_id(buf)
        if rc != 0:
            raise MpcxError('mpcx_comm_unique_id failed (%d)' % rc)
        return buf.raw

    def comm_init(self, world: int, rank: int, unique_id: bytes):
        buf = C.create_string_buffer(bytes(unique_id), _lib.COMM_ID_BYTES)
        self._chk(self.lib.mpcx_comm_init(self._ctx, int(world), int(rank), buf))
        self.comm_world, self.comm_rank = int(world), int(rank)

    def comm_destroy(self):
        self._chk(self.lib.mpcx_comm_destroy(self._ctx))
        self.comm_world, self.comm_rank = 1, 0

    @_ordered
    def allgather_states(self, layout: int, local: torch.Tensor, out: torch.Tensor):
        """mpcx_allgather_states: local (n_inst, agents_local, 6) -> out, laid out as include/mpcx.h describes for `layout`"""
        n_inst, a_loc = int(local.shape[0]), int(local.shape[1])
        self._want(local, torch.float64, (n_inst, a_loc, 6), 'local'); self._want(out, torch.float64, None, 'out')
        world = getattr(self, 'comm_world', 1)
        if out.numel() != world * local.numel():
            raise MpcxError('allgather_states: out holds %d doubles, expected %d' % (out.numel(), world * local.numel()))
        self._chk(self.lib.mpcx_allgather_states(self._ctx, int(layout), n_inst, a_loc, _ptr(local), _ptr(out)))
        return out

    def synchronize(self):
        self.stream.synchronize()


class SearchModel:
    """Device copy of the tables `neighbor_function` needs (primitive id = position in the lists given)."""

    def __init__(self, ctx: Context, templates, last_pose, edge_cost, hp, hp_off):
        self.ctx = ctx
        tmpl_off = np.cumsum([0] + [len(t) for t in templates]).astype(np.int32)
        tmpl_xy = np.ascontiguousarray(np.concatenate([np.asarray(t)[:, :2] for t in templates]), np.float64)
        last_pose = np.ascontiguousarray(last_pose, np.float64)
        edge_cost = np.ascontiguousarray(edge_cost, np.float64)
        hp = np.ascontiguousarray(hp, np.float64).reshape(-1, 3)
        hp_off = np.ascontiguousarray(hp_off, np.int32)
        self.n_prim = len(templates)
        self.n_obst = len(hp_off) - 1
        self.n_pts_of = [len(t) for t in templates]      # collision-template points per primitive
        self.n_rows = int(len(hp))                        # half-plane rows of all obstacles
        self.edge_cost = edge_cost
        vp = C.c_void_p
        self._h = ctx.lib.mpcx_search_model_create(ctx._ctx, self.n_prim, vp(tmpl_off.ctypes.data), vp(tmpl_xy.ctypes.data),
                                                   vp(last_pose.ctypes.data), vp(edge_cost.ctypes.data), self.n_obst,
                                                   vp(hp_off.ctypes.data), vp(hp.ctypes.data))
        if not self._h:
            raise MpcxError('mpcx_search_model_create: %s' % ctx.lib.mpcx_last_error(ctx._ctx).decode())

    def close(self):
        if getattr(self, '_h', None):
            self.ctx.lib.mpcx_search_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
