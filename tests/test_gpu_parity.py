"""GPU parity tests: every kernel of libmpcx.so, called through the C ABI, against the CPU oracle and the
golden fixtures captured from the reference. Run on the MI355X box: pytest -m gpu."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

QP_TOL = 2e-7      # |u_gpu - u_oracle|, |x_gpu - x_oracle| (north-star bound: 1e-4 vs the reference optimum)


@pytest.fixture(scope='module')
def ctx():
    from mpc_for_av_at_intersection_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


def _orc():
    from oracle import oracle_py as orc
    return orc


@pytest.mark.parametrize('T', [10, 13, 20])
def test_qp_vs_oracle_golden_inputs(ctx, T):
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    orc = _orc()
    g = H.gold('mpc_pre.npz')
    ctx.set_mpc_params(MpcParams(T=T))
    po = orc.MpcParams(T=T)
    st, xref, xbar, re = g['T%d/state' % T], g['T%d/xref' % T], g['T%d/xbar' % T], g['T%d/reaches_end' % T]
    out = ctx.qp_solve(ctx.f64(st), ctx.f64(xref), ctx.f64(xbar), ctx.u8(re))
    ctx.synchronize()
    u, x = out['u'].cpu().numpy(), out['x'].cpu().numpy()
    status, iters, kkt = out['status'].cpu().numpy(), out['iters'].cpu().numpy(), out['kkt'].cpu().numpy()
    assert (status == 0).all(), status
    du = dx = 0.0
    for k in range(len(st)):
        sol = orc.qp_solve(po, st[k], xref[k], xbar[k], re[k])
        assert sol.status == 0
        du = max(du, np.abs(sol.u - u[k]).max()); dx = max(dx, np.abs(sol.x - x[k]).max())
    print('T=%d max|du|=%.3e max|dx|=%.3e iters %d..%d kkt %s' % (T, du, dx, iters.min(), iters.max(), kkt.max(0)))
    assert du < QP_TOL and dx < QP_TOL


@pytest.mark.parametrize('solver', ['condensed', 'stage'])
@pytest.mark.parametrize('T', [13, 20])
def test_qp_trial_step_same_decisions_as_the_oracle(ctx, solver, T):
    """the trial pass (unconstrained minimiser, accepted when it violates no row: 0 iterations) takes the same decision in the
    oracle and in both HIP solvers on every golden closed-loop problem, and the accepted points agree to rounding"""
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    from tests.test_oracle_jerk import jerk_cases
    orc = _orc()
    cases = jerk_cases(T)
    x0 = np.stack([c[0] for c in cases]); xref = np.stack([c[1] for c in cases]); xbar = np.stack([c[2] for c in cases])
    re = np.stack([c[3] for c in cases]).astype(np.uint8)
    warm = np.stack([np.zeros((2, T)) if c[4] is None else c[4] for c in cases])
    ctx.set_mpc_params(MpcParams(T=T))
    ctx.set_qp_solver(solver)
    try:
        out = ctx.qp_solve(ctx.f64(x0), ctx.f64(xref), ctx.f64(xbar), ctx.u8(re), ctx.f64(warm))
        ctx.synchronize()
    finally:
        ctx.set_qp_solver('auto')
    it, st, u, kkt = out['iters'].cpu().numpy(), out['status'].cpu().numpy(), out['u'].cpu().numpy(), out['kkt'].cpu().numpy()
    po = orc.MpcParams(T=T)
    sols = [orc.qp_solve(po, x0[k], xref[k], xbar[k], re[k], warm[k]) for k in range(len(cases))]
    oit = np.array([s.iters for s in sols])
    assert (st == 0).all() and np.array_equal(it == 0, oit == 0) and (oit == 0).sum() >= 30
    assert (np.abs(it - oit) <= 1).all()
    z = oit == 0
    assert max(np.abs(u[k] - sols[k].u).max() for k in np.nonzero(z)[0]) < 1e-10
    assert kkt[z, 1].max() < 1e-12 and kkt[z, 2].max() == 0.0          # primal residual / complementarity of an accepted trial point


@pytest.mark.parametrize('solver', ['condensed', 'stage'])
def test_qp_vs_exact_active_set_solution(ctx, solver):
    """both HIP solvers against the EXACT minimiser of the literal problem of mpc.py:138-208 (tests/qp_literal.exact_solution:
    active-set KKT solve in numpy, no code shared with oracle.c or the kernels), all 60 golden problems, T = 20"""
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    from tests import qp_literal as QL
    orc = _orc()
    T = 20
    g = H.gold('mpc_pre.npz')
    ctx.set_mpc_params(MpcParams(T=T))
    ctx.set_qp_solver(solver)
    try:
        st, xref, xbar, re = g['T20/state'], g['T20/xref'], g['T20/xbar'], g['T20/reaches_end']
        out = ctx.qp_solve(ctx.f64(st), ctx.f64(xref), ctx.f64(xbar), ctx.u8(re))
        ctx.synchronize()
    finally:
        ctx.set_qp_solver('auto')
    u, x = out['u'].cpu().numpy(), out['x'].cpu().numpy()
    assert (out['status'].cpu().numpy() == 0).all()
    po = orc.MpcParams(T=T)
    dist = []
    for k in range(len(st)):
        z = QL.pack(po, x[k], u[k])
        ex = QL.exact_solution(po, st[k], xref[k], xbar[k], re[k], z)
        dist.append(np.abs(z - ex['z']).max())
    dist = np.array(dist)
    print('%s: |z_gpu - z_exact| max %.2e median %.2e' % (solver, dist.max(), np.median(dist)))
    assert dist.max() < 5e-6 and np.median(dist) < 1e-7      # (1e-4 is the stated tolerance against the reference's optimum; at scale: tests/test_gpu_accuracy.py)


@pytest.mark.parametrize('solver', ['condensed', 'stage', 'auto'])
def test_qp_hard_instance_from_the_soak(ctx, solver):
    """tests/golden/qp_hard.npz: the one QP of round 1's 120-step soak on which the condensed solver's factorisation broke down
    short of the tolerance (status 3 -> MAX_DECEL) while the oracle and the stage solver converge.  Problems the condensed
    solver gives up on are now re-solved by the stage solver inside the same call."""
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    orc = _orc()
    g = H.gold('qp_hard.npz')
    T = 20
    ctx.set_mpc_params(MpcParams(T=T))
    po = orc.MpcParams(T=T)
    # embedded in a batch of ordinary problems so that the second-chance list is a strict subset
    gp = H.gold('mpc_pre.npz')
    st = np.concatenate([gp['T20/state'][:15], g['x0'], gp['T20/state'][15:30]])
    xref = np.concatenate([gp['T20/xref'][:15], g['xref'], gp['T20/xref'][15:30]])
    xbar = np.concatenate([gp['T20/xbar'][:15], g['xbar'], gp['T20/xbar'][15:30]])
    re = np.concatenate([gp['T20/reaches_end'][:15], g['reaches_end'], gp['T20/reaches_end'][15:30]])
    uw = np.zeros((31, 2, T)); uw[15] = g['u_warm'][0]
    ctx.set_qp_solver(solver)
    try:
        u_io = ctx.f64(uw)                                  # warm start and output alias, as in the closed loop
        out = dict(x=torch.empty((31, 4, T + 1), dtype=torch.float64, device=ctx.device), u=u_io,
                   status=torch.empty(31, dtype=torch.int32, device=ctx.device), iters=torch.empty(31, dtype=torch.int32, device=ctx.device),
                   kkt=torch.empty((31, 4), dtype=torch.float64, device=ctx.device))
        ctx.qp_solve(ctx.f64(st), ctx.f64(xref), ctx.f64(xbar), ctx.u8(re), u_io, out=out)
        ctx.synchronize()
    finally:
        ctx.set_qp_solver('auto')
    assert (out['status'].cpu().numpy() == 0).all(), out['status']
    u = out['u'].cpu().numpy()
    for k in (0, 14, 15, 16, 30):
        sol = orc.qp_solve(po, st[k], xref[k], xbar[k], re[k], uw[k])
        # the soak's instance stagnates (stationarity stalls near 4e-7 * |g|, exit by the reduced-accuracy rule): two implementations
        # of the same iteration part ways in the last digits there
        assert sol.status == 0 and np.abs(sol.u - u[k]).max() < QP_TOL, k


@pytest.mark.parametrize('solver', ['condensed', 'stage'])
def test_qp_creeping_instance(ctx, solver):
    """tests/golden/qp_hard2.npz: a reversing ego (v = -1.2 m/s) whose reference is clipped to the path end.  The iteration creeps
    along the boundary (step lengths 0.05-0.2) for ~22 iterations and converges in ~30; with a step fraction of 0.999 and an
    unbounded centring target the stage kernel let mu collapse to 1e-23 while the stationarity residual sat at its rounding floor,
    lost its fourth consecutive reduced-accuracy iterate and ran into garbage (status 1 after 60 iterations)."""
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    orc = _orc()
    g = H.gold('qp_hard2.npz')
    T = 20
    ctx.set_mpc_params(MpcParams(T=T))
    ctx.set_qp_solver(solver)
    try:
        out = ctx.qp_solve(ctx.f64(g['x0']), ctx.f64(g['xref']), ctx.f64(g['xbar']), ctx.u8(g['re']), ctx.f64(g['uw']))
        ctx.synchronize()
    finally:
        ctx.set_qp_solver('auto')
    sol = orc.qp_solve(orc.MpcParams(T=T), g['x0'][0], g['xref'][0], g['xbar'][0], g['re'][0], g['uw'][0])
    assert sol.status == 0 and out['status'].item() == 0
    assert abs(out['iters'].item() - sol.iters) <= 2
    assert np.abs(out['u'].cpu().numpy()[0] - sol.u).max() < QP_TOL


def test_qp_warm_start_and_infeasible(ctx):
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    orc = _orc()
    T = 20
    g = H.gold('mpc_pre.npz')
    ctx.set_mpc_params(MpcParams(T=T))
    po = orc.MpcParams(T=T)
    st, xref, xbar, re = (g['T20/state'].copy(), g['T20/xref'], g['T20/xbar'], g['T20/reaches_end'])
    uw = np.stack([g['T20/oa'], g['T20/od']], axis=1)
    uw[:, 1] = np.clip(uw[:, 1], -0.7, 0.7)
    st[3, 2] = 9.5        # above MAX_SPEED: x[2,0] <= MAX_SPEED makes the problem infeasible (mpc.py:187)
    out = ctx.qp_solve(ctx.f64(st), ctx.f64(xref), ctx.f64(xbar), ctx.u8(re), ctx.f64(uw))
    ctx.synchronize()
    status = out['status'].cpu().numpy()
    u = out['u'].cpu().numpy()
    assert status[3] == 2 and (np.delete(status, 3) == 0).all()
    for k in (0, 1, 2, 4, 17, 40):
        sol = orc.qp_solve(po, st[k], xref[k], xbar[k], re[k], uw[k])
        assert np.abs(sol.u - u[k]).max() < QP_TOL


def test_prepare_vs_golden(ctx):
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    g = H.gold('mpc_pre.npz')
    for T in (10, 13, 20):
        ctx.set_mpc_params(MpcParams(T=T))
        st, pc, start = g['T%d/state' % T], g['T%d/path' % T], g['T%d/start' % T]
        paths, off, ln = [], [], []
        cur = 0
        for (sp, ti, cut) in pc:
            full = H.smoothed_path(sp, ti)
            paths.append(full); off.append(cur); ln.append(cut); cur += len(full)
        path = np.concatenate(paths)
        dl = float(np.linalg.norm(paths[0][0, :2] - paths[0][1, :2]))
        uw = np.stack([g['T%d/oa' % T], g['T%d/od' % T]], axis=1)
        tind = ctx.i32(start)
        out = ctx.prepare(ctx.f64(st), ctx.f64(uw), ctx.f64(path), ctx.i32(off), ctx.i32(ln), dl, tind)
        ctx.synchronize()
        assert np.array_equal(tind.cpu().numpy(), g['T%d/target_ind' % T])          # bit-exact index work
        assert np.array_equal(out['reaches_end'].cpu().numpy(), g['T%d/reaches_end' % T])
        assert np.array_equal(out['xref'].cpu().numpy(), g['T%d/xref' % T])          # pure gathers: exact
        d = np.abs(out['xbar'].cpu().numpy() - g['T%d/xbar' % T]).max()
        assert d < 1e-12, d                                                           # device libm vs numpy sin/cos/tan


@pytest.mark.parametrize('tag,version', [('bic', 'bicycle_model'), ('bic1', 'bicycle_model'), ('pri', 'prius')])
def test_expand_vs_golden(ctx, tag, version):
    ex = H.gold('expand.npz')
    sp, ti = ex[tag + '/scenario']
    model = ctx.search_model(*H.search_tables(version, 'int_%d_%d' % (sp, ti)))
    nodes = ex[tag + '/nodes']
    out = ctx.expand(model, ctx.f64(nodes))
    ctx.synchronize()
    col = out['collide'].cpu().numpy(); nbr = out['nbr'].cpu().numpy(); cost = out['cost'].cpu().numpy()
    assert np.array_equal(col, ex[tag + '/collide'])                                  # collide flags bit-exact
    assert np.abs(nbr - ex[tag + "/nbr"]).max() < 1e-12
    cs = np.column_stack([np.cos(nodes[:, 2]), np.sin(nodes[:, 2])])
    out2 = ctx.expand(model, ctx.f64(nodes), nodes_cs=ctx.f64(cs))
    ctx.synchronize()
    assert np.array_equal(out2["nbr"].cpu().numpy(), ex[tag + "/nbr"])            # host trig: bit-exact successors
    assert np.array_equal(out2["collide"].cpu().numpy(), ex[tag + "/collide"])
    assert np.array_equal(cost, np.broadcast_to(np.array(H.prim_meta(version)['total_length']), cost.shape))


@pytest.mark.parametrize('discs', [2, 1])
def test_interaction_vs_golden(ctx, discs):
    """discs = 1: a car with one collision disc (car_dimensions.py:51-75, skip_back_circle_collision_checking=True; golden
    moving_onedisc.npz) -- the kernels are given the same disc twice"""
    from mpc_for_av_at_intersection_amd.runtime import InteractionParams
    mv = H.gold('moving.npz')
    full = H.gold('mpc_pre.npz')['path_4_1']
    car = H.car()
    centers = np.array(car['circle_centers']) if discs == 2 else H.gold('moving_onedisc.npz')['circle_centers']
    ip = InteractionParams(cutoff_margin=int(mv['moving/margin']), L=car['L'], radius=car['radius'], circle_centers=centers.ravel())
    if discs == 1:
        one = H.gold('moving_onedisc.npz')
        mv = dict(mv); mv['moving/hit'] = one['hit']; mv['moving/cut'] = one['cut']
    n = len(mv['moving/in'])
    idx = mv['moving/in'][:, 0].astype(np.int32); v0 = mv['moving/in'][:, 1]; nobs = mv['moving/in'][:, 2].astype(np.int32)
    state = np.zeros((n, 4)); state[:, 0] = full[idx, 0]; state[:, 1] = full[idx, 1]; state[:, 2] = v0
    pool = np.concatenate([mv['moving/obs'][k][:nobs[k]] for k in range(n)])
    off = np.concatenate([[0], np.cumsum(nobs)[:-1]]).astype(np.int32)
    path_cs = np.column_stack([np.cos(full[:, 2]), np.sin(full[:, 2])])
    # prev_cut_len chosen so that traj_idx is NOT advanced (tmp[traj_idx] == tmp[-1]): the golden cases fix idx
    prev = (idx + 1).astype(np.int32)
    tidx = ctx.i32(idx)
    out = ctx.interaction(ip, ctx.f64(state), ctx.f64(full), ctx.f64(path_cs), ctx.i32(np.zeros(n)), ctx.i32(np.full(n, len(full))),
                          ctx.i32(prev), ctx.f64(pool), ctx.i32(off), ctx.i32(nobs), None, tidx)
    ctx.synchronize()
    hit = out['hit_idx'].cpu().numpy(); xy = out['hit_xy'].cpu().numpy(); cut = out['cut_len'].cpu().numpy()
    gh = mv['moving/hit']
    assert np.array_equal(hit, gh[:, 2].astype(np.int32))                             # first conflicting pose: exact
    assert np.array_equal(cut, mv['moving/cut'])
    m = hit >= 0
    assert np.array_equal(xy[m], gh[m, :2])
    assert np.array_equal(tidx.cpu().numpy(), idx)


def test_wave_helpers_selftest(ctx):
    """DPP scans / shifts / reductions used inside qp_kernel"""
    import ctypes as C
    rng = np.random.default_rng(5)
    v = rng.uniform(0.5, 3.0, 64) * rng.choice([-1.0, 1.0], 64)
    inp = ctx.f64(v); out = torch.zeros(322, dtype=torch.float64, device=ctx.device)
    rc = ctx.lib.mpcx_selftest_wave_ops(ctx._ctx, C.c_void_p(inp.data_ptr()), C.c_void_p(out.data_ptr()))
    assert rc == 0
    ctx.synchronize()
    o = out.cpu().numpy()
    assert np.abs(o[:32] - np.cumsum(v[:32])).max() < 1e-13
    assert np.abs(o[64:96] - np.cumsum(v[:32][::-1])[::-1]).max() < 1e-13
    assert np.array_equal(o[128:191], v[1:]) and o[191] == -1.0
    assert np.array_equal(o[193:256], v[:-1]) and o[192] == -1.0
    assert abs(o[256] - v.sum()) < 1e-12 and o[257] == v.max()
    assert np.abs(o[258:] * v - 1.0).max() < 1e-14


def test_mfma_f64_lane_maps(ctx):
    """v_mfma_f64_16x16x4 operand / result lane maps (asymmetric data, so a transposed map cannot pass)"""
    import ctypes as C
    rng = np.random.default_rng(9)
    A = rng.integers(-5, 6, (16, 4)).astype(np.float64); B = rng.integers(-5, 6, (4, 16)).astype(np.float64)
    dA, dB = ctx.f64(A), ctx.f64(B)
    D = torch.zeros((16, 16), dtype=torch.float64, device=ctx.device)
    assert ctx.lib.mpcx_selftest_mfma(ctx._ctx, C.c_void_p(dA.data_ptr()), C.c_void_p(dB.data_ptr()), C.c_void_p(D.data_ptr())) == 0
    ctx.synchronize()
    assert np.array_equal(D.cpu().numpy(), A @ B)


@pytest.mark.parametrize('T', [10, 13, 20, 32])
def test_qp_stage_solver_equals_condensed_solver(ctx, T):
    """The two QP kernels (stage-structured, eight lanes per problem / condensed, one wavefront per problem) on the same
    problems: same status, same iteration count (the iteration is the same, only the linear algebra differs), solutions <= 1e-9;
    both against the oracle on a sample.  T = 32 exercises the widest instantiation of both."""
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    from oracle import oracle_py as orc
    g = H.gold('mpc_pre.npz')
    Tg = T if T in (10, 13, 20) else 20
    n = 60
    st = g['T%d/state' % Tg]
    if T == Tg:
        xref, xbar, re = g['T%d/xref' % T], g['T%d/xbar' % T], g['T%d/reaches_end' % T]
    else:                                   # extend the T = 20 windows to 32 stages by holding the last column
        ext = lambda a: np.concatenate([a, np.repeat(a[..., -1:], T - Tg, axis=-1)], axis=-1)
        xref, xbar, re = ext(g['T20/xref']), ext(g['T20/xbar']), ext(g['T20/reaches_end'])
    reps = 40
    tile = lambda a: np.concatenate([a] * reps)
    ctx.set_mpc_params(MpcParams(T=T))
    dev = [ctx.f64(tile(st)), ctx.f64(tile(xref)), ctx.f64(tile(xbar)), ctx.u8(tile(re))]
    res = {}
    for which in ('stage', 'condensed'):
        ctx.set_qp_solver(which)
        out = ctx.qp_solve(*dev)
        ctx.synchronize()
        res[which] = {k: v.cpu().numpy() for k, v in out.items()}
    ctx.set_qp_solver('auto')
    a, b = res['stage'], res['condensed']
    assert np.array_equal(a['status'], b['status']) and (a['status'] == 0).all()
    assert np.abs(a['iters'].astype(int) - b['iters'].astype(int)).max() <= 1
    assert (a['iters'] == b['iters']).mean() > 0.98
    # at T = 32 the condensed 64x64 system is the less accurate of the two: one stagnating golden problem exits an iteration later
    # 2.5e-6 away, while the stage solver stays within 1e-11 of the oracle
    tol_ab = 1e-9 if T <= 20 else 5e-6
    assert np.abs(a['u'] - b['u']).max() < tol_ab and np.abs(a['x'] - b['x']).max() < 10 * tol_ab
    assert np.array_equal(a['u'][:n], a['u'][n:2 * n])                         # same problem, another group/lane: same bits
    po = orc.MpcParams(T=T)
    for i in range(0, n, 7):
        r = orc.qp_solve(po, st[i], xref[i], xbar[i], re[i])
        assert r.status == 0 and np.abs(r.u - a['u'][i]).max() < 1e-9 and np.abs(r.x - a['x'][i]).max() < 1e-9


def test_qp_order_hint_changes_schedule_not_results(ctx):
    """mpcx_qp_set_order_hint only reorders the work queue: any hint (zeros, random, adversarial) gives bit-identical outputs"""
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    g = H.gold('mpc_pre.npz')
    T, reps = 20, 50
    tile = lambda a: np.concatenate([a] * reps)
    ctx.set_mpc_params(MpcParams(T=T))
    dev = [ctx.f64(tile(g['T20/state'])), ctx.f64(tile(g['T20/xref'])), ctx.f64(tile(g['T20/xbar'])), ctx.u8(tile(g['T20/reaches_end']))]
    B = dev[0].shape[0]
    ctx.set_qp_order_hint(None)
    ref = {k: v.cpu().numpy() for k, v in ctx.qp_solve(*dev).items()}
    rng = np.random.default_rng(0)
    for hint in (np.zeros(B), rng.integers(0, 20, B), np.full(B, 500), np.arange(B) % 9, -np.ones(B)):
        h = ctx.i32(hint)
        jump = (ctx.i32(rng.integers(0, 2, B)), ctx.i32(np.zeros(B))) if hint[0] != 0 else (None, None)
        ctx.set_qp_order_hint(h, *jump)
        out = {k: v.cpu().numpy() for k, v in ctx.qp_solve(*dev).items()}
        for k in ref:
            assert np.array_equal(ref[k], out[k]), k
    # the hint may alias the iteration-count output (what the closed loop does)
    out = ctx.qp_solve(*dev)
    ctx.set_qp_order_hint(out['iters'])
    out2 = ctx.qp_solve(*dev, out=out)
    ctx.synchronize()
    assert np.array_equal(ref['u'], out2['u'].cpu().numpy()) and np.array_equal(ref['iters'], out2['iters'].cpu().numpy())
    ctx.set_qp_order_hint(None)
