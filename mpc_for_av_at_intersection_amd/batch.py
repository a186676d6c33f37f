"""Batched closed loop: B independent intersection instances x A agents advanced in lock-step on one GPU.

One `step()` is the body of the reference's scenario loop (main/scenarios/mpc_intersection.py:95-159) for every
(instance, agent) pair: nearest index on the full path, ego prediction, prediction of the other agents, conflict
search and path cut (mpcx_interaction_batch), reference window + warm-start rollout (mpcx_mpc_prepare_batch), the
QP (mpcx_qp_solve_batch) and the plant update (mpcx_plant_step_batch). Every other agent of the same instance plays
the role of the reference's `moving_obstacles`: it is described by (x, y, v, yaw, a, steer) exactly like
`MovingObstacle*.get()` (a, steer = the controls it applied last step). All state lives on the device.
`run(n)` hands the whole loop to mpcx_closed_loop_run (n steps enqueued back to back, optionally as a replayed
hipGraph); `step_staged()` drives the same kernels stage by stage through the per-stage entry points.
"""
import dataclasses
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .runtime import path_first_within, path_plan, path_tables, Context, InteractionParams, MpcParams, MpcxError


class IntersectionBatch:
    def __init__(self, ctx: Context, params: MpcParams, ip: InteractionParams, routes: Sequence[np.ndarray], dl: float,
                 route_of_agent: np.ndarray, start_index: np.ndarray, v0: Optional[np.ndarray] = None,
                 tuning: Optional[np.ndarray] = None, agent_shard: Optional[tuple] = None, exchange=None,
                 pose_offset: Optional[np.ndarray] = None):
        """routes: list of (n_r, 3) paths whose yaw column is already unwrapped (MPC.__init__, mpc.py:257);
        route_of_agent, start_index: integer arrays of shape (B, A); tuning: optional (B, 16) or (B*A, 16) array of
        MpcParams.tuning_row()s -- one cost/limit set per instance (or agent), the batched form of the reference's
        sensitivity sweeps (scenarios/mpc_sensitivity_analysis.py).

        agent_shard = (rank, world): the AGENT-SHARDED multi-GPU layout (sharding.py): this rank drives agents
        rank*A/world .. of EVERY instance and sees the other ranks' agents only as moving obstacles, through one
        all-gather of 6-double agent states per step.  `exchange` says who moves the rows: 'rccl' (mpcx_allgather_states
        on the context's communicator, inside mpcx_closed_loop_run) or a callable local(B, A_loc, 6) -> pool(B, A, 6)
        (sharding.torch_exchange: torch.distributed, used for gloo rehearsals).
        pose_offset: optional (B, A, 2) array (lateral offset [m] to the left of the path, heading error [rad]) added to the start
        poses, which otherwise sit exactly on the path (config2_batch)."""
        # (a copy: the arc-length table below belongs to THIS batch's paths)
        ip = dataclasses.replace(ip, max_path_len=max(int(ip.max_path_len), max(len(r) for r in routes)))     # sizes the interaction kernel's LDS
        self.ctx, self.params, self.ip, self.dl = ctx, params, ip, float(dl)
        ctx.set_mpc_params(params)
        route_of_agent = np.asarray(route_of_agent, dtype=np.int64)
        start_index = np.asarray(start_index, dtype=np.int64)
        self.A_total = route_of_agent.shape[1]
        self.shard_rank, self.shard_world = (0, 1) if agent_shard is None else (int(agent_shard[0]), int(agent_shard[1]))
        self.exchange = exchange
        a_lo = 0
        if agent_shard is not None:
            if self.A_total % self.shard_world:
                raise ValueError('agent-sharded layout: %d agents do not divide over %d ranks' % (self.A_total, self.shard_world))
            if exchange is None and self.shard_world > 1:
                raise ValueError('agent-sharded layout needs an exchange (\'rccl\' or a callable)')
            a_loc = self.A_total // self.shard_world
            a_lo = self.shard_rank * a_loc
            route_of_agent = route_of_agent[:, a_lo:a_lo + a_loc]
            start_index = start_index[:, a_lo:a_lo + a_loc]
            if v0 is not None:
                v0 = np.asarray(v0, dtype=np.float64).reshape(-1, self.A_total)[:, a_lo:a_lo + a_loc]
            if pose_offset is not None:
                pose_offset = np.asarray(pose_offset, dtype=np.float64).reshape(-1, self.A_total, 2)[:, a_lo:a_lo + a_loc]
        self.B, self.A = route_of_agent.shape
        P = self.P = self.B * self.A
        T = params.T
        offs = np.cumsum([0] + [len(r) for r in routes])
        table = np.concatenate(routes, axis=0).astype(np.float64)
        self.path = ctx.f64(table)
        self.path_cs = ctx.f64(np.column_stack([np.cos(table[:, 2]), np.sin(table[:, 2])]))
        # paths are constants of the run: their arc lengths are summed once here instead of by every agent in every step (the conflict
        # search takes its resampling buckets from this table wherever that is safe, mpcx_interaction_params.path_cum)
        cum, cum_err = path_tables(table, offs)
        self.path_cum = ctx.f64(cum)
        ip.path_cum, ip.path_cum_err = self.path_cum, cum_err
        # ... and so is the answer of get_cutoff_curve_by_position_idx for every path point (the conflict search only ever asks it for one)
        self.path_first_within = ctx.i32(path_first_within(table, offs))
        ip.path_first_within = self.path_first_within
        # ... and so is the ego prediction from every path point once the predicted speed has saturated (kept poses, their discs, run boxes)
        pkey = (hash(table.tobytes()), tuple(int(o) for o in offs), float(ip.dt), float(ip.max_speed), tuple(float(v) for v in np.asarray(ip.circle_centers).ravel()),
                float(ip.radius), int(ip.pred_steps))
        pl = _PLAN_CACHE.get(pkey)
        if pl is None and len(table) * 64 * 32 > (256 << 20):
            pl = False                   # 2 KB of table per path point: beyond 256 MB the kernels resample instead (identical outputs)
        if pl is None:                   # (a few tenths of a second per set of routes: numpy over every start point of every route)
            if len(_PLAN_CACHE) > 8:
                _PLAN_CACHE.clear()
            pl = _PLAN_CACHE[pkey] = path_plan(table, np.column_stack([np.cos(table[:, 2]), np.sin(table[:, 2])]), offs, ip.dt, ip.max_speed,
                                               ip.circle_centers, ip.radius, ip.pred_steps)
        self.plan = dict(pl, cnt=ctx.i32(pl['cnt']), disc=ctx.f64(pl['disc']), box=ctx.f64(pl['box']), path_disc=ctx.f64(pl['path_disc'])) if pl else None
        ip.plan = self.plan
        r = route_of_agent.reshape(-1)
        s = start_index.reshape(-1)
        self.path_off = ctx.i32(offs[r])
        self.path_len = ctx.i32(np.array([len(routes[k]) for k in r]))
        pts = table[offs[r] + s]
        st = np.zeros((P, 4))
        st[:, 0], st[:, 1], st[:, 3] = pts[:, 0], pts[:, 1], pts[:, 2]
        if v0 is not None:
            st[:, 2] = np.asarray(v0, dtype=np.float64).reshape(-1)
        if pose_offset is not None:
            po = np.asarray(pose_offset, dtype=np.float64).reshape(P, 2)
            st[:, 0] -= np.sin(pts[:, 2]) * po[:, 0]
            st[:, 1] += np.cos(pts[:, 2]) * po[:, 0]
            st[:, 3] += po[:, 1]
        dev = ctx.device
        self.state = ctx.f64(st)
        self.applied = torch.zeros((P, 2), dtype=torch.float64, device=dev)      # (steer, accel) of the last step
        self.traj_idx = ctx.i32(s)
        self.target_ind = ctx.i32(s)
        # the obstacle pool holds ALL agents of every instance, in (instance, global agent) order; an agent skips its own row
        self.obs_off = ctx.i32(np.repeat(np.arange(self.B) * self.A_total, self.A))
        self.obs_cnt = torch.full((P,), self.A_total, dtype=torch.int32, device=dev)
        self.obs_skip = ctx.i32((np.arange(self.B)[:, None] * self.A_total + a_lo + np.arange(self.A)[None, :]).reshape(-1))
        f = torch.float64
        self.obs6 = torch.zeros((self.B * self.A_total, 6), dtype=f, device=dev)
        self.obs_local = torch.zeros((P, 6), dtype=f, device=dev) if agent_shard is not None else None
        # cut_len doubles as "length of the previous tmp_trajectory" (0 = none yet) for the next step
        self.inter = dict(hit_idx=torch.empty(P, dtype=torch.int32, device=dev), hit_xy=torch.empty((P, 2), dtype=f, device=dev),
                          cut_len=torch.zeros(P, dtype=torch.int32, device=dev))
        self.pre = dict(xref=torch.empty((P, 4, T + 1), dtype=f, device=dev),
                        reaches_end=torch.empty((P, T + 1), dtype=torch.uint8, device=dev),
                        xbar=torch.empty((P, 4, T + 1), dtype=f, device=dev))
        self.sol = dict(x=torch.empty((P, 4, T + 1), dtype=f, device=dev), u=torch.zeros((P, 2, T), dtype=f, device=dev),
                        status=torch.zeros(P, dtype=torch.int32, device=dev), iters=torch.zeros(P, dtype=torch.int32, device=dev),
                        kkt=torch.zeros((P, 4), dtype=f, device=dev))
        self.steps_done = 0
        self.lin_passes = 1          # lib/mpc.py MAX_ITER: (window, rollout, QP) passes per step; the stock mpc_config.json has 1
        self.path_v = None
        self.tuning = None
        if tuning is not None:
            tuning = np.asarray(tuning, dtype=np.float64)
            if tuning.shape == (self.B, 16):
                tuning = np.repeat(tuning, self.A, axis=0)
            elif agent_shard is not None and tuning.shape == (self.B * self.A_total, 16):
                tuning = tuning.reshape(self.B, self.A_total, 16)[:, a_lo:a_lo + self.A].reshape(-1, 16)
            if tuning.shape != (P, 16):
                raise ValueError('tuning must have shape (B, 16) or (B*A, 16)')
            self.tuning = ctx.f64(tuning)
        self._desc = None

    def _descriptor(self) -> '_lib.ClosedLoopC':
        d = _lib.ClosedLoopC()
        d.P, d.exchange, d.dl = self.P, (_lib.SHARD_AGENTS if self.exchange == 'rccl' else 0), self.dl
        d.n_inst, d.agents_local = self.B, self.A
        d.obs_local = None if self.obs_local is None else self.obs_local.data_ptr()
        bufs = dict(state=self.state, applied=self.applied, obs6=self.obs6, path_xyyaw=self.path, path_cs=self.path_cs,
                    path_v=self.path_v, path_off=self.path_off, path_len=self.path_len, obs_off=self.obs_off,
                    obs_cnt=self.obs_cnt, obs_skip=self.obs_skip, traj_idx=self.traj_idx, target_ind=self.target_ind,
                    hit_idx=self.inter['hit_idx'], cut_len=self.inter['cut_len'], hit_xy=self.inter['hit_xy'],
                    xref=self.pre['xref'], xbar=self.pre['xbar'], reaches_end=self.pre['reaches_end'],
                    x_sol=self.sol['x'], u_sol=self.sol['u'], status=self.sol['status'], iters=self.sol['iters'],
                    kkt=self.sol['kkt'])
        for k, t in bufs.items():
            setattr(d, k, None if t is None else t.data_ptr())
        return d

    def _claim_context(self):
        """the kernels size every access from the context's horizon: another MPC / batch on the same Context may have
        changed it since this batch was built"""
        if self.ctx.params != self.params:
            self.ctx.set_mpc_params(self.params)
        if getattr(self.ctx, 'lin_passes', 1) != self.lin_passes:
            self.ctx.set_linearisation_passes(self.lin_passes)
        self.ctx.set_instance_tuning(self.tuning)

    def run(self, n_steps: int, graph: bool = False):
        """n_steps of the closed loop with no host work in between (mpcx_closed_loop_run)."""
        if callable(self.exchange):          # rehearsal exchange (torch.distributed): the host moves the rows between stages
            for _ in range(n_steps):
                self.step_staged()
            return
        if self._desc is None:
            self._desc = self._descriptor()
        self._claim_context()
        self.ctx.closed_loop_run(self.ip, self._desc, n_steps, graph)
        self.steps_done += n_steps

    def step(self):
        self.run(1)

    def check(self):
        """Raise where the reference raises: Exception('something wrong') of calc_nearest_index_in_direction
        (trajectories.py:120; hit_idx -3 / target_ind -1) and the capacity limits of the interaction kernel (hit_idx -2:
        more than MPCX_MAX_REMAINING path points ahead, MPCX_EGO_FRAMES_MAX resampled poses or MPCX_MAX_OBS obstacles)
        -- in all of which the kernel leaves the agent's path uncut.  One small reduction + sync; call it every N steps."""
        bad = torch.stack([(self.inter['hit_idx'] == -2).sum(), (self.inter['hit_idx'] == -3).sum(), (self.target_ind < 0).sum()]).cpu().numpy()
        if bad.any():
            raise MpcxError('closed loop: %d agents beyond the interaction kernel\'s capacity (hit_idx -2), %d + %d nearest-index failures '
                            '("something wrong", trajectories.py:120) in the conflict search / reference window' % tuple(int(b) for b in bad))

    def step_staged(self):
        """the same step through the per-stage entry points (one host call per stage)"""
        c = self.ctx
        self._claim_context()
        # what MovingObstacle*.get() would return for every agent: (x, y, v, yaw, a, steer)
        rows = self.obs6 if self.obs_local is None else self.obs_local
        rows[:, 0:2] = self.state[:, 0:2]
        rows[:, 2] = self.state[:, 2]
        rows[:, 3] = self.state[:, 3]
        rows[:, 4] = self.applied[:, 1]
        rows[:, 5] = self.applied[:, 0]
        if self.obs_local is not None:       # agent-sharded: every rank assembles the whole pool
            loc = self.obs_local.view(self.B, self.A, 6)
            if callable(self.exchange):
                self.obs6.copy_(self.exchange(loc).reshape(-1, 6))
            else:
                c.allgather_states(_lib.SHARD_AGENTS, loc, self.obs6)
        c.interaction(self.ip, self.state, self.path, self.path_cs, self.path_off, self.path_len,
                      self.inter['cut_len'], self.obs6, self.obs_off, self.obs_cnt, self.obs_skip,
                      self.traj_idx, out=self.inter)
        # the previous solution (zeros where the last solve failed or on the first step) is the warm start
        for it in range(self.lin_passes):       # lib/mpc.py:226-237: from the second pass on the previous pass's speeds space the window
            c.prepare(self.state, self.sol['u'], self.path, self.path_off, self.inter['cut_len'], self.dl, self.target_ind, out=self.pre,
                      x_prev=self.sol['x'] if it else None)
            c.qp_solve(self.state, self.pre['xref'], self.pre['xbar'], self.pre['reaches_end'], self.sol['u'], out=self.sol)
        c.plant_step(self.state, self.sol['u'], self.sol['status'], self.applied)
        self.steps_done += 1

    def snapshot(self):
        """host copies of the per-agent state (synchronises)"""
        self.ctx.synchronize()
        keys = ('state', 'applied', 'traj_idx', 'target_ind')
        out = {k: getattr(self, k).cpu().numpy().copy() for k in keys}
        out['prev_cut'] = self.inter['cut_len'].cpu().numpy().copy()
        out.update({k: v.cpu().numpy().copy() for k, v in self.sol.items()})
        out.update({k: v.cpu().numpy().copy() for k, v in self.inter.items()})
        out.update({k: v.cpu().numpy().copy() for k, v in self.pre.items()})
        return out


def stock_routes(ctx: Context, pairs=((1, 1), (1, 2), (2, 1), (2, 2), (3, 1), (3, 2), (4, 1), (4, 2))):
    """Reference paths of the stock 4-way intersection, planned with the GPU-backed MotionPrimitiveSearch
    (the `_modified` variant every stock MPC scenario uses); yaw columns unwrapped as MPC.__init__ does."""
    from .lib import _session
    from .lib.car_dimensions import BicycleModelDimensions
    from .lib.motion_primitive import load_motion_primitives
    from .lib.motion_primitive_search import plan_many
    from .lib.motion_primitive_search_modified import MotionPrimitiveSearch
    from .lib.mpc import smooth_yaw
    from .lib.scenario import intersection
    _session.set_context(ctx)
    cd = BicycleModelDimensions()
    mps = load_motion_primitives('bicycle_model')
    routes = []
    searches = [MotionPrimitiveSearch(intersection(start_pos=sp, turn_indicator=ti), cd, mps, margin=cd.radius, ctx=ctx) for sp, ti in pairs]
    for _, _, traj in plan_many(searches):           # all routes planned concurrently (one expansion launch per level)
        traj = np.ascontiguousarray(traj)
        smooth_yaw(traj[:, 2])
        routes.append(traj)
    dl = float(np.linalg.norm(routes[0][0, :2] - routes[0][1, :2]))
    return routes, dl, cd


def synthetic_batch(ctx: Context, B: int, A: int = 8, T: int = 20, seed: int = 0, routes=None, dl=None, cd=None,
                    max_start_frac: float = 0.35, instance_slice: Optional[tuple] = None, agent_shard: Optional[tuple] = None,
                    exchange=None, mpc: Optional[MpcParams] = None):
    """B instances x A agents on the stock intersection, one agent per (arm, manoeuvre)
    route, start positions staggered along the approach (seeded), v0 = 0 as in the reference's scripts.
    The workload is a function of (B, A, seed) only; a rank takes its part of it with instance_slice = (lo, hi)
    (instance-sharded) or agent_shard = (rank, world) (agent-sharded, see IntersectionBatch).
    `mpc` replaces the stock controller constants (its T wins over the argument; the wheelbase is always the car's), e.g.
    MpcParams.jerk() for the controller of lib/mpc_jerk.py."""
    if routes is None:
        routes, dl, cd = stock_routes(ctx)
    rng = np.random.default_rng(seed)
    R = len(routes)
    route_of_agent = np.tile(np.arange(A) % R, (B, 1))
    lens = np.array([len(r) for r in routes])[route_of_agent]
    start = (rng.random((B, A)) * max_start_frac * lens).astype(np.int64)
    if instance_slice is not None:
        lo, hi = instance_slice
        route_of_agent, start = route_of_agent[lo:hi], start[lo:hi]
    if mpc is None:
        params = MpcParams(T=T, L=cd.distance_back_to_front_wheel)
    else:
        params = dataclasses.replace(mpc, L=cd.distance_back_to_front_wheel)
    ip = InteractionParams(cutoff_margin=4 * int(np.ceil(cd.radius / dl)), L=cd.distance_back_to_front_wheel, radius=cd.radius,
                           circle_centers=np.asarray(cd.circle_centers).ravel())
    return IntersectionBatch(ctx, params, ip, routes, dl, route_of_agent, start, agent_shard=agent_shard, exchange=exchange)


_PLAN_CACHE = {}
ALL_STOCK_PAIRS = tuple((sp, ti) for sp in (1, 2, 3, 4) for ti in (1, 2, 3))


def config2_batch(ctx: Context, B: int = 256, T: int = 20, seed: int = 0, routes=None, dl=None, cd=None, burn_in: int = 3,
                  instance_slice: Optional[tuple] = None, mpc: Optional[MpcParams] = None):
    """SURVEY section 8(d) config 2 (BASELINE configs[1]): B INDEPENDENT single-ego instances (no other agent, hence no coupling), drawn
    with numpy.random.default_rng(seed): route uniform over the 12 stock A* paths (start_pos 1..4 x turn_indicator 1..3), arc position
    s ~ U[0, len - T vmax dt / dl] path points, lateral offset ~ N(0, 0.3 m), heading error ~ N(0, 0.05 rad), v ~ U[0, 30/3.6 m/s];
    `burn_in` (3) closed-loop steps are taken here so that every later step starts from the previous solution -- the generator
    whose QPs have "realistic active sets" (acceleration bound when slow, steering-rate bound in the turns; lib/mpc.py:184-191).
    `routes` must be the 12 paths of stock_routes(ctx, ALL_STOCK_PAIRS) when given.  The workload is a function of (B, T, seed) only;
    instance_slice = (lo, hi) takes a rank's part of it."""
    if routes is None:
        routes, dl, cd = stock_routes(ctx, ALL_STOCK_PAIRS)
    if len(routes) != len(ALL_STOCK_PAIRS):
        raise ValueError('config2_batch draws from the %d stock routes, got %d' % (len(ALL_STOCK_PAIRS), len(routes)))
    params = MpcParams(T=T, L=cd.distance_back_to_front_wheel) if mpc is None else dataclasses.replace(mpc, L=cd.distance_back_to_front_wheel)
    rng = np.random.default_rng(seed)
    route = rng.integers(0, len(routes), size=B)
    lens = np.array([len(r) for r in routes])[route]
    span = np.maximum(lens - params.T * params.max_speed * params.dt / dl, 1.0)
    start = np.floor(rng.random(B) * span).astype(np.int64)
    lateral = rng.normal(0.0, 0.3, B)
    heading = rng.normal(0.0, 0.05, B)
    v0 = rng.uniform(0.0, params.max_speed, B)
    if instance_slice is not None:
        lo, hi = instance_slice
        route, start, lateral, heading, v0 = route[lo:hi], start[lo:hi], lateral[lo:hi], heading[lo:hi], v0[lo:hi]
    ip = InteractionParams(cutoff_margin=4 * int(np.ceil(cd.radius / dl)), L=cd.distance_back_to_front_wheel, radius=cd.radius,
                           circle_centers=np.asarray(cd.circle_centers).ravel())
    sim = IntersectionBatch(ctx, params, ip, routes, dl, route[:, None], start[:, None], v0=v0[:, None],
                            pose_offset=np.stack([lateral, heading], axis=1)[:, None, :])
    if burn_in > 0:
        sim.run(burn_in)
    return sim


def prius_frontier(ctx: Context, n: int = 1 << 20, seed: int = 0, extent: float = 40.0, free_space: bool = True, embed=None):
    """the search model of the Prius primitives + PriusDimensions on the stock intersection and a
    frontier of n nodes (seeded) as a device tensor.
    free_space = True (the frontier section 8(d) defines): nodes "sampled from free space with theta ~ U[-pi, pi)" -- x, y uniform over
    the junction area, REJECTING every pose at which the car itself collides (one of its two discs inside an obstacle inflated by
    the disc radius: exactly the test check_collision makes on a pose, obstacles.py:157-176), so that no record leaves at a first hit
    that the pose alone decides -- "plus all nodes of the golden expansion logs" (`embed`: (m, 3) array written over the first m
    nodes; bench.py passes the Prius logs of tests/golden/astar_runs.npz).
    free_space = False: round 2's frontier, x, y uniform over the junction area whatever stands there (60 % of its records collide)."""
    from .lib.car_dimensions import PriusDimensions
    from .lib.motion_primitive import load_motion_primitives
    from .lib.motion_primitive_search_modified import MotionPrimitiveSearch
    from .lib.scenario import intersection
    cd = PriusDimensions()
    search = MotionPrimitiveSearch(intersection(start_pos=2, turn_indicator=1), cd, load_motion_primitives('prius'), margin=cd.radius, ctx=ctx)
    rng = np.random.default_rng(seed)

    def draw(m):
        return np.column_stack([rng.uniform(-extent, extent, m), rng.uniform(-extent, extent, m), rng.uniform(-np.pi, np.pi, m)])
    if not free_space:
        nodes = draw(n)
    else:
        hps = search._obstacles_hp                      # per obstacle: rows (a, b, c), inside <=> a x + b y + c <= 0 for every row
        centers = np.asarray(cd.circle_centers, dtype=np.float64)
        parts, have = [], 0
        while have < n:
            cand = draw(max(1 << 16, int(1.6 * (n - have))))
            c, s = np.cos(cand[:, 2]), np.sin(cand[:, 2])
            free = np.ones(len(cand), bool)
            for ox, oy in centers:                      # disc centres of the car at the pose
                px, py = cand[:, 0] + c * ox - s * oy, cand[:, 1] + s * ox + c * oy
                for hp in hps:
                    free &= ~((hp[:, 0][None, :] * px[:, None] + hp[:, 1][None, :] * py[:, None] + hp[:, 2][None, :]) <= 0.0).all(axis=1)
            parts.append(cand[free]); have += int(free.sum())
        nodes = np.concatenate(parts)[:n]
    if embed is not None and len(embed):
        e = np.asarray(embed, dtype=np.float64).reshape(-1, 3)[:n]
        nodes[:len(e)] = e
    return search._model, ctx.f64(nodes)
