"""ctypes binding of oracle/liboracle.so plus the pure-Python parts of the oracle (A* queue, MPC step glue).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package never imports this module.
"""
import ctypes as C
import heapq
import math
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_bp = C.POINTER(C.c_uint8)


def build(force=False):
    so = os.path.join(_HERE, 'liboracle.so')
    src = [os.path.join(_HERE, f) for f in ('oracle.c', 'oracle_batch.c', 'oracle_jerk.c', 'oracle.h')]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'liboracle.so'], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_nearest_index_in_direction.argtypes = [C.c_double, C.c_double, c_dp, c_dp, C.c_int32, C.c_int32, C.c_int32]
        _LIB.orc_nearest_index_in_direction.restype = C.c_int32
        _LIB.orc_calc_ref_trajectory.restype = C.c_int32
        _LIB.orc_calc_ref_trajectory_ov.restype = C.c_int32
        _LIB.orc_qp_build.restype = C.c_int32
        _LIB.orc_qp_solve.restype = C.c_int32
        _LIB.orc_qp_build_jerk.restype = C.c_int32
        _LIB.orc_qp_solve_jerk.restype = C.c_int32
        _LIB.orc_resample_curve.restype = C.c_int32
        _LIB.orc_check_collision_moving_cars.restype = C.c_int32
        _LIB.orc_cutoff_idx.restype = C.c_int32
        _LIB.orc_linear_model.argtypes = [C.c_double] * 5 + [c_dp] * 3
        _LIB.orc_xy_cost_mtx.argtypes = [C.c_double, c_dp]
        _LIB.orc_agent_steps_mt.restype = C.c_int32
    return _LIB


def _d(a):
    return a.ctypes.data_as(c_dp)


def _i(a):
    return a.ctypes.data_as(c_ip)


def _b(a):
    return a.ctypes.data_as(c_bp)


class _CParams(C.Structure):
    _fields_ = [('T', C.c_int32), ('max_iter', C.c_int32), ('dt', C.c_double), ('L', C.c_double),
                ('w_perp', C.c_double), ('w_para', C.c_double), ('R', C.c_double * 2), ('Rd', C.c_double * 2),
                ('Q_v_yaw', C.c_double * 2), ('Qf', C.c_double * 4), ('R_end', C.c_double * 2),
                ('max_speed', C.c_double), ('min_speed', C.c_double), ('max_accel', C.c_double),
                ('max_decel', C.c_double), ('max_steer', C.c_double), ('max_dsteer', C.c_double), ('tol', C.c_double),
                ('model', C.c_int32), ('reserved', C.c_int32), ('jerk_weight', C.c_double)]


@dataclass
class MpcParams:
    """main/config/mpc_config.json + lib/mpc.py:17-36 + lib/simulation.py:23-25"""
    T: int = 13
    dt: float = 0.2
    L: float = 2.86
    w_perp: float = 20.0
    w_para: float = 1.0
    R: tuple = (0.01, 0.01)
    Rd: tuple = (0.01, 1.0)
    Q_v_yaw: tuple = (0.0, 0.5)
    Qf_base: tuple = (1.0, 1.0, 0.0, 0.5)
    R_end: tuple = (10.0, 10.0)
    max_speed: float = 30.0 / 3.6
    min_speed: float = -5.0
    max_accel: float = 2.0
    max_decel: float = -10.0
    max_steer: float = float(np.deg2rad(45.0))
    max_dsteer: float = float(np.deg2rad(30.0))
    max_iter: int = 60
    tol: float = 1e-10
    model: int = 0              # 1: the five-state problem of lib/mpc_jerk.py
    jerk_weight: float = 1.0    # mpc_jerk.py:30

    @classmethod
    def jerk(cls, **kw):
        """the constants of main/lib/mpc_jerk.py:16-39 (T 13, cross-track weight 10, Rd (.3, 1), MAX_DECEL -5)"""
        base = dict(T=13, w_perp=10.0, w_para=1.0, R=(0.01, 0.01), Rd=(0.3, 1.0), Q_v_yaw=(0.0, 0.5),
                    Qf_base=(1.0, 1.0, 0.0, 0.5), max_accel=2.0, max_decel=-5.0, model=1, jerk_weight=1.0)
        base.update(kw)
        return cls(**base)

    def c(self):
        p = _CParams()
        p.T, p.max_iter, p.dt, p.L = self.T, self.max_iter, self.dt, self.L
        p.w_perp, p.w_para = self.w_perp, self.w_para
        p.R[:] = self.R; p.Rd[:] = self.Rd; p.Q_v_yaw[:] = self.Q_v_yaw
        p.Qf[:] = [q * self.T for q in self.Qf_base]
        p.R_end[:] = self.R_end
        p.max_speed, p.min_speed, p.max_accel, p.max_decel = self.max_speed, self.min_speed, self.max_accel, self.max_decel
        p.max_steer, p.max_dsteer, p.tol = self.max_steer, self.max_dsteer, self.tol
        p.model, p.jerk_weight = self.model, self.jerk_weight
        return p


def linear_model(v, phi, delta, dt, L):
    A = np.zeros((4, 4)); B = np.zeros((4, 2)); Cc = np.zeros(4)
    lib().orc_linear_model(v, phi, delta, dt, L, _d(A), _d(B), _d(Cc))
    return A, B, Cc


def xy_cost_mtx(angle):
    M = np.zeros((2, 2))
    lib().orc_xy_cost_mtx(angle, _d(M))
    return M


def smooth_yaw(yaw):
    yaw = np.ascontiguousarray(yaw, dtype=np.float64)
    lib().orc_smooth_yaw(_d(yaw), C.c_int32(len(yaw)))
    return yaw


def nearest_index_in_direction(x, y, cx, cy, start, forward=True):
    cx = np.ascontiguousarray(cx, np.float64); cy = np.ascontiguousarray(cy, np.float64)
    return lib().orc_nearest_index_in_direction(float(x), float(y), _d(cx), _d(cy), len(cx), int(start), int(forward))


def calc_ref_trajectory(p: MpcParams, state4, cx, cy, cyaw, dl, start_idx, cv=None, ov=None):
    """mpc.py:86-109; ov = the previous linearisation pass's speeds (mpc.py:226-237, MAX_ITER > 1) or None"""
    cx = np.ascontiguousarray(cx, np.float64); cy = np.ascontiguousarray(cy, np.float64)
    cyaw = np.ascontiguousarray(cyaw, np.float64)
    st = np.ascontiguousarray(state4, np.float64)
    xref = np.zeros((4, p.T + 1)); re = np.zeros(p.T + 1, np.uint8)
    cp = p.c()
    cvv = None if cv is None else np.ascontiguousarray(cv, np.float64)
    ovv = None if ov is None else np.ascontiguousarray(ov, np.float64)
    s = lib().orc_calc_ref_trajectory_ov(C.byref(cp), _d(st), _d(cx), _d(cy), _d(cyaw), None if cvv is None else _d(cvv),
                                         None if ovv is None else _d(ovv), C.c_int32(len(cx)), C.c_double(dl),
                                         C.c_int32(start_idx), _d(xref), _b(re))
    return xref, s, re


def predict_motion(p: MpcParams, x0, oa, od):
    x0 = np.ascontiguousarray(x0, np.float64); oa = np.ascontiguousarray(oa, np.float64); od = np.ascontiguousarray(od, np.float64)
    xbar = np.zeros((4, p.T + 1))
    cp = p.c()
    lib().orc_predict_motion(C.byref(cp), _d(x0), _d(oa), _d(od), _d(xbar))
    return xbar


def plant_step(p: MpcParams, state4, a, delta):
    s = np.array(state4, np.float64)
    cp = p.c()
    lib().orc_plant_step(C.byref(cp), _d(s), C.c_double(a), C.c_double(delta))
    return s


def qp_build(p: MpcParams, x0, xref, xbar, reaches_end):
    T = p.T; n = 2 * T + (p.model == 1); nx = 5 if p.model == 1 else 4
    x0 = np.ascontiguousarray(x0, np.float64); xref = np.ascontiguousarray(np.asarray(xref)[:4], np.float64)
    xbar = np.ascontiguousarray(np.asarray(xbar)[:4], np.float64); re = np.ascontiguousarray(reaches_end, np.uint8)
    H = np.zeros((n, n)); g = np.zeros(n); G = np.zeros((8 * T, n)); h = np.zeros(8 * T)
    S = np.zeros((T + 1, nx, n)); c = np.zeros((T + 1, nx))
    cp = p.c()
    fn = lib().orc_qp_build_jerk if p.model == 1 else lib().orc_qp_build
    m = fn(C.byref(cp), _d(x0), _d(xref), _d(xbar), _b(re), _d(H), _d(g), _d(G), _d(h), _d(S), _d(c))
    return H, g, G[:m], h[:m], S, c


@dataclass
class QpSolution:
    status: int
    x: np.ndarray
    u: np.ndarray
    lam: np.ndarray
    iters: int
    kkt: np.ndarray


def qp_solve(p: MpcParams, x0, xref, xbar, reaches_end, u_warm=None) -> QpSolution:
    """model 0: x is 4 x (T+1).  model 1 (lib/mpc_jerk.py): x is 5 x (T+1), row 4 the acceleration state (the reference
    returns rows 0..3 only, mpc_jerk.py:201-206)"""
    T = p.T
    x0 = np.ascontiguousarray(x0, np.float64); xref = np.ascontiguousarray(np.asarray(xref)[:4], np.float64)
    xbar = np.ascontiguousarray(np.asarray(xbar)[:4], np.float64); re = np.ascontiguousarray(reaches_end, np.uint8)
    x = np.zeros((5 if p.model == 1 else 4, T + 1)); u = np.zeros((2, T)); lam = np.zeros(8 * T); it = C.c_int32(0); kkt = np.zeros(4)
    uw = None if u_warm is None else np.ascontiguousarray(u_warm, np.float64)
    cp = p.c()
    fn = lib().orc_qp_solve_jerk if p.model == 1 else lib().orc_qp_solve
    st = fn(C.byref(cp), _d(x0), _d(xref), _d(xbar), _b(re), None if uw is None else _d(uw),
                            _d(x), _d(u), _d(lam), C.byref(it), _d(kkt))
    return QpSolution(st, x, u, lam[:8 * T - 2], it.value, kkt)


# ---------------------------------------------------------------- search model
class _CSearch(C.Structure):
    _fields_ = [('n_prim', C.c_int32), ('n_obst', C.c_int32), ('tmpl_off', c_ip), ('tmpl_xy', c_dp),
                ('last_pose', c_dp), ('edge_cost', c_dp), ('hp_off', c_ip), ('hp', c_dp)]


class SearchModel:
    """Flattened tables the expansion needs (primitive ids = sorted names)."""

    def __init__(self, templates, last_pose, edge_cost, hp, hp_off):
        self.tmpl_off = np.cumsum([0] + [len(t) for t in templates]).astype(np.int32)
        self.tmpl_xy = np.ascontiguousarray(np.concatenate([np.asarray(t)[:, :2] for t in templates]), np.float64)
        self.last_pose = np.ascontiguousarray(last_pose, np.float64)
        self.edge_cost = np.ascontiguousarray(edge_cost, np.float64)
        self.hp = np.ascontiguousarray(hp, np.float64)
        self.hp_off = np.ascontiguousarray(hp_off, np.int32)
        self.n_prim = len(templates)
        self.n_obst = len(hp_off) - 1

    def c(self):
        return _CSearch(self.n_prim, self.n_obst, _i(self.tmpl_off), _d(self.tmpl_xy), _d(self.last_pose),
                        _d(self.edge_cost), _i(self.hp_off), _d(self.hp))


def expand(model: SearchModel, nodes, host_trig=True):
    """-> nbr (n,P,3), collide (n,P). host_trig: cos/sin from numpy, as the reference computes them."""
    nodes = np.ascontiguousarray(nodes, np.float64).reshape(-1, 3)
    n = len(nodes)
    nbr = np.zeros((n, model.n_prim, 3)); col = np.zeros((n, model.n_prim), np.uint8)
    cs = np.ascontiguousarray(np.column_stack([np.cos(nodes[:, 2]), np.sin(nodes[:, 2])])) if host_trig else None
    cm = model.c()
    lib().orc_expand(C.byref(cm), C.c_int32(n), _d(nodes), None if cs is None else _d(cs), _d(nbr), _b(col))
    return nbr, col


def resample_curve(pts, dl, keep_last=True):
    pts = np.ascontiguousarray(pts, np.float64)
    n, stride = pts.shape
    keep = np.zeros(n, np.int32)
    if np.ndim(dl) == 0:
        k = lib().orc_resample_curve(_d(pts), n, stride, None, C.c_double(float(dl)), int(keep_last), _i(keep))
    else:
        dlv = np.ascontiguousarray(dl, np.float64)
        k = lib().orc_resample_curve(_d(pts), n, stride, _d(dlv), C.c_double(0.0), int(keep_last), _i(keep))
    return pts[keep[:k]].copy()


def predict_obstacle(six, dt, L, steps=35):
    six = np.ascontiguousarray(six, np.float64)
    out = np.zeros((steps, 4))
    lib().orc_predict_obstacle(_d(six), C.c_double(dt), C.c_double(L), C.c_int32(steps), _d(out))
    return out


def check_collision_moving_cars(centers, radius, traj_agent, path, traj_obs, frame_window):
    centers = np.ascontiguousarray(centers, np.float64)
    ta = np.ascontiguousarray(traj_agent[:, :3], np.float64); pa = np.ascontiguousarray(path[:, :3], np.float64)
    if len(traj_obs) == 0:
        return None
    to = np.ascontiguousarray(np.stack(traj_obs), np.float64)
    hit = np.zeros(2)
    idx = lib().orc_check_collision_moving_cars(_d(centers), len(centers), C.c_double(radius), _d(ta), len(ta), _d(pa), len(pa),
                                                _d(to), to.shape[0], to.shape[1], int(frame_window), _d(hit))
    return None if idx < 0 else (hit[0], hit[1], idx)


def cutoff_idx(pts, x, y, radius=0.001):
    pts = np.ascontiguousarray(pts[:, :3], np.float64)
    return lib().orc_cutoff_idx(_d(pts), len(pts), C.c_double(x), C.c_double(y), C.c_double(radius))


# ---------------------------------------------------------------- A* (lib/a_star.py:31-78)
def a_star(start, is_goal, heuristic, neighbors, max_expansions=10 ** 7):
    """Best-first search with the reference's queue semantics: entries (g+h, g, node, pred) ordered as tuples,
    lazy deletion (skip when seen with g >= best), push only when unseen or strictly better.
    -> (cost, path, log[(node, g, h, pred)])"""
    heap = [(0, 0, start, start)]
    best = {}
    log = []
    while heap:
        f, g, node, pred = heapq.heappop(heap)
        seen = best.get(node)
        if seen is not None and g >= seen[0]:
            continue
        log.append((node, g, f - g, pred))
        best[node] = (g, pred)
        if is_goal(node):
            path = [node]
            while node != start:
                path.append(pred)
                node, pred = pred, best[pred][1]
            return g, path[::-1], log
        if len(log) > max_expansions:
            break
        for cost, nb in neighbors(node):
            ng = g + cost
            seen = best.get(nb)
            if seen is None or ng < seen[0]:
                heapq.heappush(heap, (ng + heuristic(nb), ng, nb, node))
    raise Exception("No solution found.")


# ---------------------------------------------------------------- one closed-loop step of one ego
def agent_step(p: MpcParams, full, dl, state4, obs6, traj_idx, prev_cut, target_ind, u_warm, centers, radius,
               cutoff_margin, pred_steps=35, frame_window=20, max_accel=2.0):
    """The body of main/scenarios/mpc_intersection.py:95-159 for one ego, on the oracle's C functions.
    full: (n,3) path with unwrapped yaw; obs6: (K,6) other vehicles; prev_cut: length of the previous tmp_trajectory
    (0/None = none yet); u_warm: (2,T) previous solution or None. Returns a dict with every intermediate."""
    full = np.ascontiguousarray(full, np.float64)
    x, y, v, yaw = state4
    # :103-105 advance traj_agent_idx unless tmp_trajectory collapsed onto it
    advance = True
    if prev_cut:
        advance = bool(np.any(full[traj_idx] != full[prev_cut - 1]))
    if advance:
        traj_idx = nearest_index_in_direction(x, y, full[:, 0], full[:, 1], traj_idx)
        if traj_idx < 0:
            raise Exception("something wrong")
    traj = full[traj_idx:]
    # :110-116 ego prediction
    if v < p.max_speed:
        rdl = np.cumsum(np.zeros(len(traj)) + max_accel) + v
        rdl = p.dt * np.minimum(rdl, p.max_speed)
        tres = resample_curve(traj, rdl)
    else:
        tres = resample_curve(traj, p.dt * p.max_speed)
    # :119-122 predictions of the others, :125-136 conflict + cut
    trajs = [predict_obstacle(s6, p.dt, p.L, pred_steps) for s6 in obs6]
    hit = check_collision_moving_cars(centers, radius, tres, traj, trajs, frame_window)
    if hit is not None:
        cut = cutoff_idx(full, hit[0], hit[1]) - cutoff_margin
        cut = max(traj_idx + 1, cut)
    else:
        cut = len(full)
    tmp = full[:cut]
    # mpc.py:211-239
    xref, target_ind, re = calc_ref_trajectory(p, state4, tmp[:, 0], tmp[:, 1], tmp[:, 2], dl, target_ind)
    if target_ind < 0:
        raise Exception("something wrong")
    uw = np.zeros((2, p.T)) if u_warm is None else np.asarray(u_warm, float)
    xbar = predict_motion(p, [x, y, v, yaw], uw[0], uw[1])
    sol = qp_solve(p, [x, y, v, yaw], xref, xbar, re, uw)
    return dict(traj_idx=traj_idx, hit=hit, cut=cut, target_ind=target_ind, xref=xref, xbar=xbar, re=re, sol=sol,
                n_res=len(tres))


def agent_steps_batch(p: MpcParams, n_threads, A, path, path_off, path_len, dl, state, applied, u_warm, traj_idx, prev_cut,
                      target_ind, centers, radius, cutoff_margin, pred_steps=35, frame_window=20, max_accel=2.0):
    """orc_agent_steps_mt: agent_step for every (instance, agent) pair of a captured batch state, on n_threads host threads, all
    in C.  Agent q belongs to instance q // A and sees its A-1 instance mates as moving obstacles (x, y, v, yaw, a, steer).
    Returns dict(out6 (P,6: traj_idx, cut, target_ind, hit, status, iters), x (P,4,T+1), u (P,2,T), threads)."""
    P = len(state)
    T = p.T
    path = np.ascontiguousarray(path, np.float64)
    po = np.ascontiguousarray(path_off, np.int32); pl = np.ascontiguousarray(path_len, np.int32)
    st = np.ascontiguousarray(state, np.float64); ap = np.ascontiguousarray(applied, np.float64)
    uw = np.ascontiguousarray(u_warm, np.float64).reshape(P, 2, T)
    ti = np.ascontiguousarray(traj_idx, np.int32); pc = np.ascontiguousarray(prev_cut, np.int32); tg = np.ascontiguousarray(target_ind, np.int32)
    ce = np.ascontiguousarray(centers, np.float64).reshape(2, 2)
    out6 = np.zeros((P, 6), np.int32); x = np.zeros((P, 4, T + 1)); u = np.zeros((P, 2, T))
    cp = p.c()
    n = lib().orc_agent_steps_mt(C.c_int32(int(n_threads)), C.byref(cp), C.c_int32(P), C.c_int32(int(A)), _d(path), _i(po), _i(pl),
                                 C.c_double(float(dl)), _d(st), _d(ap), _d(uw), _i(ti), _i(pc), _i(tg), _d(ce), C.c_double(float(radius)),
                                 C.c_int32(int(cutoff_margin)), C.c_int32(int(pred_steps)), C.c_int32(int(frame_window)),
                                 C.c_double(float(max_accel)), _i(out6), _d(x), _d(u))
    if n < 1:
        raise RuntimeError('orc_agent_steps_mt failed')
    return dict(out6=out6, x=x, u=u, threads=n)
