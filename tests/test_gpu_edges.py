"""GPU edge cases of the C ABI: empty batches, invalid arguments, horizons that use the padded / largest kernel
instantiations, degenerate paths, no obstacles, size limits."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from mpc_for_av_at_intersection_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


def _random_problems(T, n, seed):
    """synthetic but realistic pre-QP tensors: a path window + a rollout of random controls (oracle builds them)"""
    from oracle import oracle_py as orc
    rng = np.random.default_rng(seed)
    full = H.smoothed_path(1, 1)
    dl = float(np.linalg.norm(full[0, :2] - full[1, :2]))
    p = orc.MpcParams(T=T)
    out = []
    for k in range(n):
        i0 = int(rng.integers(0, len(full) - 10))
        cut = len(full) if k % 2 else min(len(full), i0 + int(rng.integers(5, 200)))
        st = np.array([full[i0, 0] + rng.normal(0, .3), full[i0, 1] + rng.normal(0, .3), rng.uniform(0, 8.3), full[i0, 2] + rng.normal(0, .05)])
        xref, s, re = orc.calc_ref_trajectory(p, st, full[:cut, 0], full[:cut, 1], full[:cut, 2], dl, max(0, i0 - 1))
        oa = rng.uniform(-3, 2, T); od = rng.uniform(-.5, .5, T)
        out.append((st, xref, orc.predict_motion(p, st, oa, od), re, np.stack([oa, od])))
    return p, out


@pytest.mark.parametrize('T', [1, 2, 5, 11, 16, 21, 27, 32])
def test_qp_all_horizons_vs_oracle(ctx, T):
    """T=1..10 -> qp_kernel<10> with padding unknowns, 11..13 -> <13>, 14..20 -> <20>, 21..32 -> <32> (64 unknowns)"""
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    from oracle import oracle_py as orc
    p, probs = _random_problems(T, 24, seed=T)
    ctx.set_mpc_params(MpcParams(T=T))
    st = np.array([q[0] for q in probs]); xref = np.array([q[1] for q in probs]); xbar = np.array([q[2] for q in probs])
    re = np.array([q[3] for q in probs]); uw = np.array([q[4] for q in probs])
    out = ctx.qp_solve(ctx.f64(st), ctx.f64(xref), ctx.f64(xbar), ctx.u8(re), ctx.f64(uw))
    ctx.synchronize()
    u = out['u'].cpu().numpy(); x = out['x'].cpu().numpy(); status = out['status'].cpu().numpy()
    worst = 0.0
    for k in range(len(probs)):
        o = orc.qp_solve(p, st[k], xref[k], xbar[k], re[k], uw[k])
        assert o.status == status[k]
        if o.status == 0:
            worst = max(worst, np.abs(o.u - u[k]).max(), np.abs(o.x - x[k]).max())
    assert (status == 0).all() and worst < 1e-6, (status, worst)


def test_empty_batches_and_invalid_arguments(ctx):
    from mpc_for_av_at_intersection_amd.runtime import InteractionParams, MpcParams, MpcxError
    ctx.set_mpc_params(MpcParams(T=13))
    f = torch.float64
    z = lambda *s, dt=f: torch.zeros(s, dtype=dt, device=ctx.device)
    out = ctx.qp_solve(z(0, 4), z(0, 4, 14), z(0, 4, 14), z(0, 14, dt=torch.uint8))            # B = 0: no launch, no error
    assert out['u'].shape == (0, 2, 13)
    ctx.prepare(z(0, 4), None, z(5, 3), z(0, dt=torch.int32), z(0, dt=torch.int32), 0.1, z(0, dt=torch.int32))
    with pytest.raises(MpcxError):
        ctx.set_mpc_params(MpcParams(T=33))                                                     # beyond MPCX_T_MAX
    with pytest.raises(MpcxError):
        ctx.set_mpc_params(MpcParams(T=0))
    with pytest.raises(MpcxError):
        ctx.qp_solve(z(2, 4), z(2, 4, 13), z(2, 4, 14), z(2, 14, dt=torch.uint8))               # wrong shape caught on the host
    lib = ctx.lib
    rc = lib.mpcx_qp_solve_batch(ctx._ctx, 1, None, None, None, None, None, None, None, None, None, None)
    assert rc == -1 and b'null' in lib.mpcx_last_error(ctx._ctx)                                # MPCX_E_INVALID from the C side
    assert lib.mpcx_set_mpc_params(ctx._ctx, None) == -1
    with pytest.raises(MpcxError):
        ctx.search_model([np.zeros((600, 3))], np.zeros((1, 3)), np.zeros(1), np.zeros((4, 3)), np.array([0, 4]))  # > 512 template points
    with pytest.raises(MpcxError):
        ctx.prepare(z(1, 4), None, z(5, 3), z(1, dt=torch.int32), z(1, dt=torch.int32) + 5, 0.0, z(1, dt=torch.int32))   # dl <= 0


def test_degenerate_paths_and_no_obstacles(ctx):
    from mpc_for_av_at_intersection_amd.runtime import InteractionParams, MpcParams
    from oracle import oracle_py as orc
    T = 13
    ctx.set_mpc_params(MpcParams(T=T))
    po = orc.MpcParams(T=T)
    full = H.smoothed_path(2, 1)
    dl = float(np.linalg.norm(full[0, :2] - full[1, :2]))
    # paths of length 1, 2, 3 and a start index beyond the (cut) path: trajectories.py:122-126 and mpc.py:100
    cases = [(1, 0), (2, 0), (3, 0), (3, 2), (50, 60), (len(full), len(full) - 1)]
    st = np.array([[full[0, 0], full[0, 1], 1.0, full[0, 2]]] * len(cases))
    tind = ctx.i32([c[1] for c in cases])
    pre = ctx.prepare(ctx.f64(st), None, ctx.f64(full), ctx.i32([0] * len(cases)), ctx.i32([c[0] for c in cases]), dl, tind)
    ctx.synchronize()
    for k, (n, s0) in enumerate(cases):
        xref, s, re = orc.calc_ref_trajectory(po, st[k], full[:n, 0], full[:n, 1], full[:n, 2], dl, s0)
        assert int(tind.cpu()[k]) == s
        assert np.array_equal(pre['xref'].cpu().numpy()[k], xref) and np.array_equal(pre['reaches_end'].cpu().numpy()[k], re)
    # interaction with an empty obstacle pool -> None (collision_avoidance.py:69-70), path untouched
    car = H.car()
    ip = InteractionParams(L=car['L'], radius=car['radius'], circle_centers=np.array(car['circle_centers']).ravel())
    cs = np.column_stack([np.cos(full[:, 2]), np.sin(full[:, 2])])
    ti = ctx.i32([3, 10])
    out = ctx.interaction(ip, ctx.f64(st[:2]), ctx.f64(full), ctx.f64(cs), ctx.i32([0, 0]), ctx.i32([len(full)] * 2), None,
                          None, ctx.i32([0, 0]), ctx.i32([0, 0]), None, ti)
    ctx.synchronize()
    assert (out['hit_idx'].cpu().numpy() == -1).all() and (out['cut_len'].cpu().numpy() == len(full)).all()
    # more obstacles than MPCX_MAX_OBS -> per-problem limit code -2, not a crash
    pool = np.tile([[100.0, 100.0, 0.0, 0.0, 0.0, 0.0]], (20, 1))
    out = ctx.interaction(ip, ctx.f64(st[:1]), ctx.f64(full), ctx.f64(cs), ctx.i32([0]), ctx.i32([len(full)]), None,
                          ctx.f64(pool), ctx.i32([0]), ctx.i32([20]), None, ctx.i32([0]))
    ctx.synchronize()
    assert int(out['hit_idx'].cpu()[0]) == -2


def test_closed_loop_run_statistics():
    """mpcx_closed_loop_stats: the device-side counters of mpcx_closed_loop_run (agent-steps, iterations, failures, max iterations)
    equal what the per-step outputs add up to"""
    from mpc_for_av_at_intersection_amd.batch import synthetic_batch
    from mpc_for_av_at_intersection_amd.runtime import Context
    ctx = Context(0)
    sim = synthetic_batch(ctx, B=40, A=8, T=20, seed=9)
    ctx.closed_loop_stats(reset=True)
    it_sum = fails = it_max = 0
    for _ in range(7):
        sim.step()
        it = sim.sol['iters'].cpu().numpy(); st = sim.sol['status'].cpu().numpy()
        it_sum += int(it.sum()); fails += int((st != 0).sum()); it_max = max(it_max, int(it.max()))
    s = ctx.closed_loop_stats(reset=True)
    assert s == dict(agent_steps=7 * sim.P, iterations=it_sum, failures=fails, max_iterations=it_max), s
    sim.run(3)
    s = ctx.closed_loop_stats(reset=False)
    assert s['agent_steps'] == 3 * sim.P and ctx.closed_loop_stats()['agent_steps'] == 3 * sim.P
    assert ctx.closed_loop_stats()['agent_steps'] == 0


def test_context_on_a_stream_of_its_own():
    """A Context bound to its own HIP stream, used while torch's CURRENT stream is another one (inputs uploaded and outputs read on
    the default stream): every call is ordered against the current stream on both sides (runtime._ordered), so the results equal
    the single-stream run bit for bit -- route planning (hundreds of small launches with host round trips), a batched solve and a
    coupled closed loop."""
    from mpc_for_av_at_intersection_amd.batch import stock_routes, synthetic_batch
    from mpc_for_av_at_intersection_amd.runtime import Context, MpcParams
    ref = Context(0)
    own = Context(0, stream=torch.cuda.Stream())
    assert own.stream != torch.cuda.current_stream()
    try:
        out = []
        for c in (ref, own):
            routes, dl, cd = stock_routes(c)
            sim = synthetic_batch(c, B=96, A=8, T=20, seed=2, routes=routes, dl=dl, cd=cd)
            sim.run(6)
            sim.step()
            snap = sim.snapshot()
            g = H.gold('mpc_pre.npz')
            c.set_mpc_params(MpcParams(T=20))
            sol = c.qp_solve(c.f64(g['T20/state']), c.f64(g['T20/xref']), c.f64(g['T20/xbar']), c.u8(g['T20/reaches_end']))
            out.append((routes, snap, sol['u'].cpu().numpy(), sol['iters'].cpu().numpy()))
        (r0, s0, u0, i0), (r1, s1, u1, i1) = out
        assert len(r0) == len(r1) and all(np.array_equal(a, b) for a, b in zip(r0, r1))
        for k in s0:
            assert np.array_equal(s0[k], s1[k]), k
        assert np.array_equal(u0, u1) and np.array_equal(i0, i1)
    finally:
        own.close(); ref.close()


def test_linearisation_passes_batch_equals_staged(ctx):
    """lib/mpc.py MAX_ITER > 1 in the batched closed loop: mpcx_closed_loop_run with mpcx_set_linearisation_passes(2) is bit-identical
    to driving (interaction, [window with the previous pass's speeds, rollout, QP] x 2, plant) through the per-stage entry points, and
    differs from the one-pass loop"""
    from mpc_for_av_at_intersection_amd.batch import synthetic_batch
    sims = [synthetic_batch(ctx, B=16, A=8, T=13, seed=5) for _ in range(3)]
    sims[0].lin_passes = sims[1].lin_passes = 2
    for _ in range(6):
        sims[0].run(1)
        sims[1].step_staged()
        sims[2].run(1)
    a, b, c = (s.snapshot() for s in sims)
    for k in ('state', 'u', 'x', 'target_ind', 'cut_len', 'status', 'xref'):
        assert np.array_equal(a[k], b[k]), k
    assert (a['status'] == 0).all() and not np.array_equal(a['u'], c['u'])


def test_one_call_of_n_steps_equals_n_calls_of_one_step(ctx):
    """mpcx_closed_loop_run(n) enqueues n steps back to back (what bench.py times): every output -- states, applied inputs, solutions, cut
    lengths, indices, run statistics -- must be bit-identical to stepping one call at a time, for the stage solver and for the condensed
    one (with its second-chance launch), with two linearisation passes, and for calls of mixed lengths."""
    from mpc_for_av_at_intersection_amd.batch import synthetic_batch
    for B, kw in ((64, {}), (1536, {}), (48, dict(lin=2))):
        sims = [synthetic_batch(ctx, B=B, A=8, T=20, seed=9) for _ in range(2)]
        for s in sims:
            s.lin_passes = kw.get('lin', 1)
        ctx.closed_loop_stats(reset=True)
        sims[0].run(3); sims[0].run(11); sims[0].run(1); sims[0].run(6)
        st_a = ctx.closed_loop_stats(reset=True)
        for _ in range(21):
            sims[1].run(1)
        st_b = ctx.closed_loop_stats(reset=True)
        a, b = sims[0].snapshot(), sims[1].snapshot()
        for k in a:
            assert np.array_equal(a[k], b[k]), (B, k)
        assert st_a == st_b and st_a['agent_steps'] == 21 * B * 8, (st_a, st_b)
        assert (a['status'] == 0).all()


def test_arc_length_table_changes_nothing(ctx):
    """mpcx_interaction_params.path_cum (round 3): the conflict search takes its resampling buckets from the caller's arc-length table where
    that is safe.  Same integer outputs, bit for bit, as deriving the step lengths from the points -- on a batch in the middle of its run
    (paths cut, cars at standstill and under way), and with a deliberately USELESS error bound (every agent then takes the sequential
    fallback) as well as with the true one."""
    import dataclasses
    from mpc_for_av_at_intersection_amd.batch import stock_routes, synthetic_batch
    routes, dl, cd = stock_routes(ctx)
    sim = synthetic_batch(ctx, B=96, A=8, T=20, seed=5, routes=routes, dl=dl, cd=cd)
    assert sim.ip.path_cum is not None and 0 < sim.ip.path_cum_err < 1e-9 and sim.ip.plan is not None
    for burn in (3, 40):
        sim.run(burn)
        ctx.synchronize()
        outs = []
        for ip in (dataclasses.replace(sim.ip, path_cum=None, path_first_within=None, plan=None), sim.ip, dataclasses.replace(sim.ip, path_cum_err=1.0),
                   dataclasses.replace(sim.ip, path_first_within=None), dataclasses.replace(sim.ip, plan=None),
                   dataclasses.replace(sim.ip, plan=None, path_cum_err=1.0)):
            tr = sim.traj_idx.clone()
            o = ctx.interaction(ip, sim.state, sim.path, sim.path_cs, sim.path_off, sim.path_len, sim.inter['cut_len'].clone(), sim.obs6,
                                sim.obs_off, sim.obs_cnt, sim.obs_skip, tr)
            ctx.synchronize()
            outs.append((tr.cpu().numpy(), o['hit_idx'].cpu().numpy(), o['cut_len'].cpu().numpy(), o['hit_xy'].cpu().numpy()))
        for other in outs[1:]:
            for x, y in zip(outs[0], other):
                assert np.array_equal(x, y)
        assert (outs[0][1] >= 0).any() and (outs[0][1] == -1).any()          # conflicts and free agents both occur


def test_resampling_by_search_on_unevenly_sampled_paths(ctx):
    """Round 4: with the arc-length table the kept poses of `resample_curve` (trajectories.py:58-86) come from one probe per bucket boundary
    at the index a straight line through the table predicts, and from a binary search where that misses.  Planner paths are evenly
    sampled (the guess always hits); here two thirds of the path points are dropped at random, so steps of 1 .. 8 spacings alternate and
    the guess misses most of the time: same integer outputs as the point-by-point scan without the table and as the sequential fallback."""
    import dataclasses
    from mpc_for_av_at_intersection_amd.batch import stock_routes, synthetic_batch
    routes, dl, cd = stock_routes(ctx)
    rng = np.random.default_rng(11)
    thin = []
    for r in routes:
        keep = np.sort(np.concatenate([[0, len(r) - 1], rng.choice(np.arange(1, len(r) - 1), (len(r) - 2) // 3, replace=False)]))
        keep = np.sort(np.concatenate([keep, rng.choice(keep[1:-1], len(keep) // 10, replace=False)]))      # and every tenth kept point twice in a row:
        thin.append(np.ascontiguousarray(r[keep]))                                                           # duplicate points (the cut-index table is not the identity)
    sim = synthetic_batch(ctx, B=64, A=8, T=20, seed=7, routes=thin, dl=dl, cd=cd)
    assert sim.ip.path_cum is not None and sim.ip.path_first_within is not None
    fw = sim.ip.path_first_within.cpu().numpy(); offs = np.cumsum([0] + [len(r) for r in thin])
    assert sum(int((fw[a:b] != np.arange(b - a)).sum()) for a, b in zip(offs[:-1], offs[1:])) > 50           # the duplicates point at their first copy
    seen_conflict = False
    for burn in (2, 12, 25):
        sim.run(burn)
        ctx.synchronize()
        outs = []
        for ip in (dataclasses.replace(sim.ip, path_cum=None, path_first_within=None, plan=None), sim.ip, dataclasses.replace(sim.ip, path_cum_err=1.0),
                   dataclasses.replace(sim.ip, path_first_within=None), dataclasses.replace(sim.ip, plan=None),
                   dataclasses.replace(sim.ip, plan=None, path_cum_err=1.0)):
            tr = sim.traj_idx.clone()
            o = ctx.interaction(ip, sim.state, sim.path, sim.path_cs, sim.path_off, sim.path_len, sim.inter['cut_len'].clone(), sim.obs6,
                                sim.obs_off, sim.obs_cnt, sim.obs_skip, tr)
            ctx.synchronize()
            outs.append((tr.cpu().numpy(), o['hit_idx'].cpu().numpy(), o['cut_len'].cpu().numpy(), o['hit_xy'].cpu().numpy()))
        for other in outs[1:]:
            for x, y in zip(outs[0], other):
                assert np.array_equal(x, y)
        assert (outs[0][1] != -2).all()                                      # nobody beyond the kernel's capacity (-3: the reference's 'something wrong' on a duplicate point)
        seen_conflict |= bool((outs[0][1] >= 0).any())
    assert seen_conflict


@pytest.mark.parametrize('n_prim,n_obst,pts', [(5, 40, 20), (16, 70, 9), (1, 3, 40), (9, 24, 14)])
def test_bulk_expansion_equals_per_lane_and_oracle_on_synthetic_models(ctx, n_prim, n_obst, pts):
    """The bulk expansion kernel (>= 4096 nodes: pairs of a wavefront worked off together, record boxes from the template's box, obstacle boxes
    four at a time by scalar loads) on models that exercise what the stock ones do not: other primitive counts (the node / primitive split of
    a block), more than 32 obstacles (second chunk of the cull), templates of more than 32 points (second mask word), obstacles packed so
    densely that a wavefront's pair queue overflows (per-lane fallback), boxes and octagons mixed.  Against the per-lane kernel of the small
    launches and against the oracle, bit for bit (flags, successor poses, costs)."""
    from oracle import oracle_py as orc
    rng = np.random.default_rng(100 * n_prim + n_obst)
    templates = [np.column_stack([rng.uniform(-1, 6, pts if k % 2 == 0 else max(1, pts // 3)), rng.uniform(-2, 2, pts if k % 2 == 0 else max(1, pts // 3))]) for k in range(n_prim)]
    last = np.column_stack([rng.uniform(1, 6, n_prim), rng.uniform(-2, 2, n_prim), rng.uniform(-0.8, 0.8, n_prim)])
    cost = rng.uniform(1, 6, n_prim)
    hp, off = [], [0]
    for o in range(n_obst):
        cx, cy, hx, hy = rng.uniform(-30, 30), rng.uniform(-30, 30), rng.uniform(0.5, 9), rng.uniform(0.5, 9)
        rows = [[1, 0, -(cx + hx)], [-1, 0, cx - hx], [0, 1, -(cy + hy)], [0, -1, cy - hy]]
        if o % 3 == 0:          # an octagon: four diagonal rows behind the axis-aligned ones (obstacles.py:140-148 order)
            r = 1.2 * max(hx, hy)
            rows += [[a, b, -(a * cx + b * cy + r)] for a, b in ((0.7071, 0.7071), (-0.7071, 0.7071), (0.7071, -0.7071), (-0.7071, -0.7071))]
        hp += rows; off.append(len(hp))
    hp, off = np.array(hp, dtype=np.float64), np.array(off, dtype=np.int32)
    model = ctx.search_model(templates, last, cost, hp, off)
    om = orc.SearchModel(templates, last, cost, hp, off)
    n = 9000
    nodes = np.column_stack([rng.uniform(-35, 35, n), rng.uniform(-35, 35, n), rng.uniform(-np.pi, np.pi, n)])
    nodes[7] = [0.0, 0.0, 0.3]                      # linalg.py:13-17: rotation only
    dev = ctx.f64(nodes)
    cs = ctx.f64(np.column_stack([np.cos(nodes[:, 2]), np.sin(nodes[:, 2])]))
    bulk = ctx.expand(model, dev, nodes_cs=cs)
    ctx.synchronize()
    col, nbr, cst = bulk['collide'].cpu().numpy(), bulk['nbr'].cpu().numpy(), bulk['cost'].cpu().numpy()
    for lo in (0, 3000, 6000):
        part = ctx.expand(model, dev[lo:lo + 3000].contiguous(), nodes_cs=cs[lo:lo + 3000].contiguous())
        assert np.array_equal(part['collide'].cpu().numpy(), col[lo:lo + 3000])
        assert np.array_equal(part['nbr'].cpu().numpy(), nbr[lo:lo + 3000]) and np.array_equal(part['cost'].cpu().numpy(), cst[lo:lo + 3000])
    onbr, ocol = orc.expand(om, nodes, host_trig=True)
    assert np.array_equal(ocol, col) and np.array_equal(onbr, nbr)
    assert 0.02 < col.mean() < 0.98
