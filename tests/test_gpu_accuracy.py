"""Accuracy of the QP solve AT SCALE: both HIP solvers against the exact minimiser of the literal problem of main/lib/mpc.py:138-208
(tests/qp_literal.exact_solution: active-set KKT solve in numpy, no code shared with oracle.c or the kernels) on QPs harvested from
the benchmark's own closed loop -- every >= 10-iteration problem of the start-up and steady-state windows plus a random sample, 4096
in all -- and on the golden cold starts at T = 10 / 13 / 20.

Round 2 measured this distance on 180 problems only, and the interior-point iterate turned out to sit up to 1e-3 from the optimum on
the hard closed-loop problems (weakly active rows: s ~ lam ~ sqrt(mu), amplified by the low curvature 2R = 0.02 of the input cost).
Round 3's active-set polish (csrc/mpcx_qp_stage.h: MPCX_POLISH) ends every constrained solve on a verified KKT point instead.
Bar: max < 5e-6 (the north star allows 1e-4 against the reference's optimum; VERDICT r2 asked for < 5e-5)."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

BAR = 5e-6


@pytest.fixture(scope='module')
def ctx():
    from mpc_for_av_at_intersection_amd.runtime import Context
    from mpc_for_av_at_intersection_amd.lib import _session
    c = Context(0)
    yield c
    if _session._ctx is c:          # synthetic_batch() made this context the drop-in classes' session context: do not leave a closed one behind
        _session.set_context(None)
    c.close()


def _distances(ctx, T, x0, xref, xbar, re, uw, solvers=('condensed', 'stage')):
    """{solver: |z_gpu - z_exact| per problem}, plus the iteration counts of the last solver"""
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    from oracle import oracle_py as orc
    from tests import qp_literal as QL
    ctx.set_mpc_params(MpcParams(T=T))
    po = orc.MpcParams(T=T)
    res, exact = {}, [None] * len(x0)
    for name in solvers:
        ctx.set_qp_solver(name)
        try:
            out = ctx.qp_solve(ctx.f64(x0), ctx.f64(xref), ctx.f64(xbar), ctx.u8(re), None if uw is None else ctx.f64(uw))
            ctx.synchronize()
        finally:
            ctx.set_qp_solver('auto')
        assert (out['status'].cpu().numpy() == 0).all()
        u, x, it = out['u'].cpu().numpy(), out['x'].cpu().numpy(), out['iters'].cpu().numpy()
        dist = np.empty(len(x0))
        for k in range(len(x0)):
            z = QL.pack(po, x[k], u[k])
            if exact[k] is None:
                ex = QL.exact_solution(po, x0[k], xref[k], xbar[k], re[k], z)
                assert ex['eq'] < 1e-9 and (ex['lam'] >= -1e-7).all() and ex['slack'].min() > -1e-9, k
                exact[k] = ex['z']
            dist[k] = np.abs(z - exact[k]).max()
        res[name] = dist
    return res, it


def test_closed_loop_corpus_vs_exact_minimiser(ctx):
    """>= 4096 QPs out of the benchmark's closed loop (4096 instances x 8 agents, T = 20): steps 3-8 and 100-105, every problem
    with >= 10 interior-point iterations plus a random sample of the rest"""
    c = H.harvest_closed_loop_qps(ctx, total=4096, hard_iters=10)
    n_hard = int((c['iters'] >= 10).sum())
    assert len(c['iters']) >= 4096 and n_hard >= 200, (len(c['iters']), n_hard)
    res, it = _distances(ctx, 20, c['x0'], c['xref'], c['xbar'], c['re'], c['uw'])
    for name, d in res.items():
        print('%s: %d problems (%d with >= 10 iterations in the closed loop; up to %d): |z_gpu - z_exact| max %.2e  p99 %.2e  median %.2e'
              % (name, len(d), n_hard, c['iters'].max(), d.max(), np.quantile(d, .99), np.median(d)))
        assert d.max() < BAR, (name, d.max(), int(d.argmax()))
        assert np.median(d) < 1e-8


@pytest.mark.parametrize('T', [10, 13, 20])
def test_golden_cold_starts_vs_exact_minimiser(ctx, T):
    """the 60 golden problems per horizon (cold starts: the hardest kind, 10-17 iterations), both solvers"""
    g = H.gold('mpc_pre.npz')
    res, it = _distances(ctx, T, g['T%d/state' % T], g['T%d/xref' % T], g['T%d/xbar' % T], g['T%d/reaches_end' % T], None)
    for name, d in res.items():
        print('T=%d %s: |z_gpu - z_exact| max %.2e median %.2e (iterations up to %d)' % (T, name, d.max(), np.median(d), it.max()))
        assert d.max() < BAR, (name, d.max())


def test_committed_corpus_vs_exact_minimiser(ctx):
    """tests/golden/qp_corpus.npz: the hardest and the formerly least accurate problems of a harvest (up to 22 iterations, up to 1e-3 from
    the optimum before the polish), kept as a fixture so that the CPU suite sees them too (tests/test_oracle_qp.py)"""
    g = H.gold('qp_corpus.npz')
    res, it = _distances(ctx, 20, g['x0'], g['xref'], g['xbar'], g['re'], g['uw'])
    for name, d in res.items():
        print('%s: max %.2e (before the polish: %.2e)' % (name, d.max(), g['dist_before_polish'].max()))
        assert d.max() < BAR, (name, d.max())


def test_config2_workload_vs_exact_minimiser(ctx):
    """SURVEY 8(d) config 2 (batch.config2_batch, 256 independent perturbed-state instances, T = 20): every QP of the four steps after
    the burn-in, inputs pulled off the device as the kernel saw them, both solvers against the exact minimiser of the literal problem"""
    from mpc_for_av_at_intersection_amd.batch import config2_batch
    sim = config2_batch(ctx, B=256, T=20, seed=0)
    rows = {k: [] for k in ('x0', 'xref', 'xbar', 're', 'uw', 'it')}
    for _ in range(4):
        uw = sim.sol['u'].clone()
        sim.step()
        rows['x0'].append(sim.sol['x'][:, :, 0].cpu().numpy()); rows['xref'].append(sim.pre['xref'].cpu().numpy())
        rows['xbar'].append(sim.pre['xbar'].cpu().numpy()); rows['re'].append(sim.pre['reaches_end'].cpu().numpy())
        rows['uw'].append(uw.cpu().numpy()); rows['it'].append(sim.sol['iters'].cpu().numpy())
    c = {k: np.concatenate(v) for k, v in rows.items()}
    assert (c['it'] > 0).mean() > 0.2                  # constrained problems are a real share of this workload
    res, it = _distances(ctx, 20, c['x0'], c['xref'], c['xbar'], c['re'], c['uw'])
    for name, d in res.items():
        print('config 2, %s: %d problems (%.0f %% constrained, up to %d iterations): |z_gpu - z_exact| max %.2e median %.2e'
              % (name, len(d), 100.0 * (c['it'] > 0).mean(), c['it'].max(), d.max(), np.median(d)))
        assert d.max() < BAR, (name, d.max(), int(d.argmax()))


# ---------------------------------------------------------------------------------------------------------------------------------------
# round 4: the accuracy evidence must not lean on the GPU's own answer, and must cover every controller / horizon

def test_exact_minimiser_does_not_depend_on_where_it_starts(ctx):
    """`exact_solution` above is started from the active set of the GPU's answer.  Its result is a verified KKT point of a strictly
    convex problem -- THE minimiser whatever the start -- and this shows it: 256 CONSTRAINED harvested problems solved again from the
    EMPTY active set (z_start = 0: no row is tight there) land on the same z to 1e-9, with the same active rows carrying multipliers."""
    from oracle import oracle_py as orc
    from tests import qp_literal as QL
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    c = H.harvest_closed_loop_qps(ctx, B=1024, total=1024, hard_iters=8, windows=((3, 3), (60, 3)))
    sel = np.nonzero(c['iters'] > 0)[0][:320]
    assert len(sel) >= 280
    ctx.set_mpc_params(MpcParams(T=20))
    po = orc.MpcParams(T=20)
    out = ctx.qp_solve(ctx.f64(c['x0'][sel]), ctx.f64(c['xref'][sel]), ctx.f64(c['xbar'][sel]), ctx.u8(c['re'][sel]), ctx.f64(c['uw'][sel]))
    u, x = out['u'].cpu().numpy(), out['x'].cpu().numpy()
    worst, n_act, rounds, unsettled = 0.0, [], [], 0
    held = lambda e: {i for i, l in zip(e['active'], e['lam']) if l > 1e-7}
    for j, k in enumerate(sel):
        z_gpu = QL.pack(po, x[j], u[j])
        a = QL.exact_solution(po, c['x0'][k], c['xref'][k], c['xbar'][k], c['re'][k], z_gpu)
        try:
            b = QL.exact_solution(po, c['x0'][k], c['xref'][k], c['xbar'][k], c['re'][k], np.zeros_like(z_gpu), max_rounds=400)
        except RuntimeError:        # the plain add / drop iteration can cycle from a start this far away: no answer, not a different answer
            unsettled += 1
            continue
        worst = max(worst, float(np.abs(a['z'] - b['z']).max()))
        assert held(a) == held(b), (k, sorted(held(a) ^ held(b)))
        n_act.append(len(a['active'])); rounds.append(b['rounds'])
    print('exact minimiser from the GPU\'s set vs from the empty set: max |dz| %.2e over %d constrained problems (%d more did not settle from the '
          'empty set; active rows: mean %.1f, max %d; rounds from empty: mean %.1f, max %d)'
          % (worst, len(rounds), unsettled, np.mean(n_act), max(n_act), np.mean(rounds), max(rounds)))
    assert len(rounds) >= 256 and unsettled <= len(sel) // 10
    assert worst < 1e-9


@pytest.mark.parametrize('T', [10, 13])
def test_closed_loop_harvest_short_horizons_vs_exact_minimiser(ctx, T):
    """the stock horizons (mpc_config.json: T = 13; config 1 of SURVEY 8(d): T = 10) in the coupled closed loop, harvested like the T = 20 corpus"""
    c = H.harvest_closed_loop_qps(ctx, B=1024, T=T, total=1024, hard_iters=8, windows=((3, 4), (60, 4)))
    assert len(c['iters']) >= 1024 and (c['iters'] > 0).sum() >= 100
    res, it = _distances(ctx, T, c['x0'], c['xref'], c['xbar'], c['re'], c['uw'])
    for name, d in res.items():
        print('T=%d closed loop, %s: %d problems (%d constrained, up to %d iterations): |z_gpu - z_exact| max %.2e median %.2e'
              % (T, name, len(d), int((c['iters'] > 0).sum()), c['iters'].max(), d.max(), np.median(d)))
        assert d.max() < BAR, (name, d.max(), int(d.argmax()))


def test_jerk_closed_loop_harvest_vs_exact_minimiser(ctx):
    """the five-state controller (main/lib/mpc_jerk.py:143-208, T = 13) in the coupled closed loop: 1024 harvested problems (every one with
    >= 8 iterations + a sample) against the exact minimiser of the literal five-state problem (tests/qp_literal.py, jerk rows :190,193).
    Round 3 had 82 golden-derived problems only (tests/test_gpu_jerk.py)."""
    from oracle import oracle_py as orc
    from tests import qp_literal as QL
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    c = H.harvest_closed_loop_qps(ctx, B=1024, total=1024, hard_iters=8, windows=((3, 4), (60, 4)), mpc=MpcParams.jerk())
    assert len(c['iters']) >= 1024 and (c['iters'] > 0).sum() >= 100
    ctx.set_mpc_params(MpcParams.jerk())
    po = orc.MpcParams.jerk()
    assert po.T == c['xref'].shape[2] - 1 == 13
    out = ctx.qp_solve(ctx.f64(c['x0']), ctx.f64(c['xref']), ctx.f64(c['xbar']), ctx.u8(c['re']), ctx.f64(c['uw']))
    ctx.synchronize()
    assert (out['status'].cpu().numpy() == 0).all()
    u, x = out['u'].cpu().numpy(), out['x'].cpu().numpy()
    dist = np.empty(len(u))
    for k in range(len(u)):
        # the reference returns rows 0..3; the fifth follows from the dynamics: x4_0 = (v_1 - v_0) / dt - a_0, x4' = x4 + dt a
        x4 = (x[k, 2, 1] - x[k, 2, 0]) / po.dt - u[k, 0, 0] + po.dt * np.concatenate([[0.0], np.cumsum(u[k, 0])])
        z = QL.pack(po, np.vstack([x[k], x4]), u[k])
        ex = QL.exact_solution(po, c['x0'][k], c['xref'][k], c['xbar'][k], c['re'][k], z)
        assert ex['eq'] < 1e-9 and (ex['lam'] >= -1e-7).all() and ex['slack'].min() > -1e-9, k
        dist[k] = np.abs(z - ex['z']).max()
    print('five-state controller, closed loop: %d problems (%d constrained, up to %d iterations): |z_gpu - z_exact| max %.2e p99 %.2e median %.2e'
          % (len(dist), int((c['iters'] > 0).sum()), c['iters'].max(), dist.max(), np.quantile(dist, .99), np.median(dist)))
    assert dist.max() < BAR, (dist.max(), int(dist.argmax()))
