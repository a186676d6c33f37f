"""GPU parity tests of the five-state controller of main/lib/mpc_jerk.py (mpcx_mpc_params.model = MPCX_MODEL_JERK5): the
stage-structured solver's seven-state sweep through the C ABI against the oracle (oracle/oracle_jerk.c), against the exact
minimiser of the literal cvxpy problem (tests/qp_literal.py), the drop-in class lib.mpc_jerk.MPC and the batched closed loop.
Parity status as for lib/mpc.py: no ECOS here, no reference output of this module -- see tests/test_oracle_jerk.py."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_oracle_jerk import jerk_cases

pytestmark = pytest.mark.gpu

QP_TOL = 2e-7


@pytest.fixture(scope='module')
def ctx():
    from mpc_for_av_at_intersection_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


def _stack(cases, T):
    x0 = np.stack([c[0] for c in cases]); xref = np.stack([c[1] for c in cases]); xbar = np.stack([c[2] for c in cases])
    re = np.stack([c[3] for c in cases]).astype(np.uint8)
    warm = np.stack([np.zeros((2, T)) if c[4] is None else c[4] for c in cases])
    return x0, xref, xbar, re, warm


@pytest.mark.parametrize('T', [10, 13, 20])
def test_jerk_qp_vs_oracle_and_exact_solution(ctx, T):
    """every closed-loop QP of the golden run as a five-state problem, tiled to 8 copies (several wavefronts, queue refills):
    statuses identical, solutions within QP_TOL of the oracle and 1e-4 / median 1e-7 of the exact minimiser, copies bit-identical"""
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    from oracle import oracle_py as orc
    from tests import qp_literal as QL
    cases = jerk_cases(T)
    x0, xref, xbar, re, warm = _stack(cases, T)
    n, rep = len(cases), 8
    ctx.set_mpc_params(MpcParams.jerk(T=T))
    tile = lambda a: np.concatenate([a] * rep)
    out = ctx.qp_solve(ctx.f64(tile(x0)), ctx.f64(tile(xref)), ctx.f64(tile(xbar)), ctx.u8(tile(re)), ctx.f64(tile(warm)))
    ctx.synchronize()
    u, x = out['u'].cpu().numpy(), out['x'].cpu().numpy()
    status, iters = out['status'].cpu().numpy(), out['iters'].cpu().numpy()
    assert (status == 0).all(), status
    for r in range(1, rep):
        assert np.array_equal(u[r * n:(r + 1) * n], u[:n]) and np.array_equal(x[r * n:(r + 1) * n], x[:n])
        assert np.array_equal(iters[r * n:(r + 1) * n], iters[:n])
    po = orc.MpcParams.jerk(T=T)
    worst, dist, it_diff = 0.0, [], 0
    for k in range(n):
        sol = orc.qp_solve(po, x0[k], xref[k], xbar[k], re[k], warm[k])
        assert sol.status == 0
        worst = max(worst, np.abs(sol.u - u[k]).max(), np.abs(sol.x[:4] - x[k]).max())
        it_diff += int(sol.iters != iters[k])
        # the reference returns rows 0..3; the fifth follows from the dynamics: x4_0 = (v_1 - v_0) / dt - a_0, x4' = x4 + dt a
        x4 = (x[k, 2, 1] - x[k, 2, 0]) / po.dt - u[k, 0, 0] + po.dt * np.concatenate([[0.0], np.cumsum(u[k, 0])])
        z = QL.pack(po, np.vstack([x[k], x4]), u[k])
        ex = QL.exact_solution(po, x0[k], xref[k], xbar[k], re[k], z)
        dist.append(np.abs(z - ex['z']).max())
    dist = np.array(dist)
    print('T=%d jerk: |gpu - oracle| %.2e, |gpu - exact| max %.2e median %.2e, %d of %d iteration counts differ' %
          (T, worst, dist.max(), np.median(dist), it_diff, n))
    assert worst < QP_TOL and it_diff <= max(1, n // 20)
    assert dist.max() < 5e-6 and np.median(dist) < 1e-7      # (1e-4 is the stated tolerance against the reference's optimum)


def test_jerk_long_horizon_kernel(ctx):
    """T = 28: the four-stages-per-lane instantiation; inputs are the T = 20 cases with the window extended by its last column"""
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    from oracle import oracle_py as orc
    T = 28
    cases = jerk_cases(20, stride=4)
    ext = lambda a: np.concatenate([a, np.repeat(a[..., -1:], T - 20, axis=-1)], axis=-1)
    po = orc.MpcParams.jerk(T=T)
    rows = []
    for x0, xref, xbar, re, warm in cases:
        w = np.zeros((2, T)) if warm is None else ext(warm)
        rows.append((x0, ext(xref), orc.predict_motion(po, x0, w[0], w[1]), ext(re), w))
    x0, xref, xbar, re, warm = _stack(rows, T)
    ctx.set_mpc_params(MpcParams.jerk(T=T))
    out = ctx.qp_solve(ctx.f64(x0), ctx.f64(xref), ctx.f64(xbar), ctx.u8(re), ctx.f64(warm))
    ctx.synchronize()
    u, st = out['u'].cpu().numpy(), out['status'].cpu().numpy()
    worst = 0.0
    for k in range(len(rows)):
        sol = orc.qp_solve(po, x0[k], xref[k], xbar[k], re[k], warm[k])
        assert sol.status == st[k]
        if sol.status == 0:
            worst = max(worst, np.abs(sol.u - u[k]).max())
    assert worst < QP_TOL, worst


def test_jerk_needs_the_stage_solver(ctx):
    from mpc_for_av_at_intersection_amd.runtime import MpcParams, MpcxError
    x0, xref, xbar, re, warm = _stack(jerk_cases(13)[:4], 13)
    ctx.set_mpc_params(MpcParams.jerk())
    ctx.set_qp_solver('condensed')
    try:
        with pytest.raises(MpcxError, match='five-state'):
            ctx.qp_solve(ctx.f64(x0), ctx.f64(xref), ctx.f64(xbar), ctx.u8(re), ctx.f64(warm))
    finally:
        ctx.set_qp_solver('auto')
    with pytest.raises(MpcxError, match='model'):
        ctx.set_mpc_params(MpcParams.jerk(model=7))


def test_lib_mpc_jerk_drop_in_closed_loop(ctx):
    """`from lib.mpc_jerk import MPC, MAX_ACCEL` (scenarios/mpc_intersection.py:20): the class in a closed loop on the stock
    route of the golden run, each step checked against the oracle's glue from the same state and warm start"""
    from mpc_for_av_at_intersection_amd.lib import mpc_jerk
    from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions
    from mpc_for_av_at_intersection_amd.lib.simulation import Simulation, State
    from oracle import oracle_py as orc
    assert (mpc_jerk.NX, mpc_jerk.T, mpc_jerk.MAX_ACCEL, mpc_jerk.MAX_DECEL, mpc_jerk.jerk_penalty_weight) == (5, 13, 2.0, -5, 1)
    full = H.gold('closedloop.npz')['T13/full'].copy()
    cd = BicycleModelDimensions()
    dl = float(np.linalg.norm(full[0, :2] - full[1, :2]))
    mpc = mpc_jerk.MPC(cx=full[:, 0], cy=full[:, 1], cyaw=full[:, 2], dl=dl, dt=0.2, car_dimensions=cd, ctx=ctx)
    state = State(x=full[0, 0], y=full[0, 1], yaw=full[0, 2], v=0.0)
    sim = Simulation(initial_state=state, car_dimensions=cd, sample_time=0.2)
    po = orc.MpcParams.jerk(L=cd.distance_back_to_front_wheel)
    worst, tind = 0.0, 0
    for step in range(60):
        warm = None if mpc.oa is None else np.stack([mpc.oa, mpc.odelta])
        s4 = [state.x, state.y, state.v, state.yaw]
        di, ai = mpc.step(state)
        xref, tind, re_ = orc.calc_ref_trajectory(po, s4, mpc.cx, mpc.cy, mpc.cyaw, dl, tind)
        w = np.zeros((2, po.T)) if warm is None else warm
        sol = orc.qp_solve(po, s4, xref, orc.predict_motion(po, s4, w[0], w[1]), re_, w)
        assert mpc.target_ind == tind and mpc.status == sol.status == 0
        worst = max(worst, np.abs(mpc.oa - sol.u[0]).max(), np.abs(mpc.odelta - sol.u[1]).max(), np.abs(mpc.ov - sol.x[2]).max())
        assert mpc.ox.shape == (po.T + 1,)
        state = sim.step(ai, di)
        if mpc.is_goal(state):
            break
    print('lib.mpc_jerk closed loop: %d steps, worst |gpu - oracle| %.2e, final v %.3f' % (step + 1, worst, state.v))
    assert worst < QP_TOL and step >= 30


def test_jerk_batched_closed_loop_vs_oracle(ctx):
    """256 instances x 8 agents with the jerk controller in the coupled closed loop; every agent of four steps replayed on the oracle"""
    from mpc_for_av_at_intersection_amd.batch import stock_routes, synthetic_batch
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    from tests.test_gpu_fullsize import _replay_all_on_oracle
    routes, dl, cd = stock_routes(ctx)
    sim = synthetic_batch(ctx, B=256, A=8, seed=5, routes=routes, dl=dl, cd=cd, mpc=MpcParams.jerk())
    assert sim.params.T == 13 and sim.params.model == 1
    sim.run(10)
    before = sim.snapshot()
    worst = 0.0
    for _ in range(4):
        sim.step()
        after = sim.snapshot()
        w, it_diff, failed = _replay_all_on_oracle(sim, before, after)
        worst = max(worst, w)
        assert it_diff <= 8
        before = after
    sim.check()
    print('jerk 256 x 8: worst |GPU - oracle| = %.2e' % worst)
