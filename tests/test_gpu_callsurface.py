"""GPU tests of the reference's Python call surface (mpc_for_av_at_intersection_amd.lib.*) against golden runs of the
reference: full A* searches (cost, primitive ids and expansion ORDER exact, coordinates <= 1e-12) and the stock
closed loop of main/scenarios/mpc_intersection.py (integer decisions exact, states within a stated tolerance)."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu
COORD_TOL = 1e-12


def _setup(version='bicycle_model'):
    from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions, PriusDimensions
    from mpc_for_av_at_intersection_amd.lib.motion_primitive import load_motion_primitives
    cd = BicycleModelDimensions() if version == 'bicycle_model' else PriusDimensions()
    return cd, load_motion_primitives(version)


def _check_run(search, runs, pre, names_sorted):
    cost, path, traj = search.run(debug=True)
    assert cost == float(runs[pre + 'cost'])
    gp = runs[pre + 'path']
    assert len(path) == len(gp) and np.abs(np.array(path) - gp).max() < COORD_TOL
    seq = [names_sorted.index(search._points_to_mp_names[a, b]) for a, b in zip(path[:-1], path[1:])]
    assert seq == runs[pre + 'seq'].tolist()                                  # primitive ids: exact
    dbg = search.debug_data
    assert len(dbg) == len(runs[pre + 'dbg_g'])                               # same number of expansions ...
    assert np.abs(np.array([d.node for d in dbg]) - runs[pre + 'dbg_node']).max() < COORD_TOL     # ... in the same order
    assert np.array_equal(np.array([d.g for d in dbg]), runs[pre + 'dbg_g'])
    assert np.abs(np.array([d.h for d in dbg]) - runs[pre + 'dbg_h']).max() < 1e-11
    assert traj.shape == runs[pre + 'traj'].shape and np.abs(traj - runs[pre + 'traj']).max() < COORD_TOL
    return search


@pytest.mark.parametrize('sp,ti', [(sp, ti) for sp in (1, 2, 3, 4) for ti in (1, 2, 3)])
def test_search_modified_all_stock_routes(sp, ti):
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search_modified import MotionPrimitiveSearch
    from mpc_for_av_at_intersection_amd.lib.scenario import intersection
    cd, mps = _setup()
    s = MotionPrimitiveSearch(intersection(turn_indicator=ti, start_pos=sp), cd, mps, margin=cd.radius)
    _check_run(s, H.gold('astar_runs.npz'), 'mod_bic_%d_%d/' % (sp, ti), sorted(mps))
    assert s.kernel_launches < len(s.debug_data) + 2          # batching: fewer launches than expansions


@pytest.mark.parametrize('variant,pre,sp,ti,kw', [
    ('base', 'base_bic_4_1/', 4, 1, {}), ('base', 'base_bic_3_2/', 3, 2, {}), ('base', 'base_bic_2_3/', 2, 3, {}),
    ('multi_lane', 'ml_bic_4_1/', 4, 1, {}), ('multi_lane', 'ml_bic_1_2/', 1, 2, {}),
    ('multi_lane', 'mlw_bic_4_1/', 4, 1, dict(wh_obstacle=0.2, wh_center=0.1, wc_center=0.05))])
def test_search_other_variants(variant, pre, sp, ti, kw):
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import MotionPrimitiveSearch
    from mpc_for_av_at_intersection_amd.lib.scenario import intersection
    cd, mps = _setup()
    s = MotionPrimitiveSearch(intersection(turn_indicator=ti, start_pos=sp), cd, mps, margin=cd.radius, variant=variant, **kw)
    _check_run(s, H.gold('astar_runs.npz'), pre, sorted(mps))


def test_search_prius_and_geometry_helpers():
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search_modified import MotionPrimitiveSearch
    from mpc_for_av_at_intersection_amd.lib.scenario import intersection
    cd, mps = _setup('prius')
    runs = H.gold('astar_runs.npz')
    if 'mod_pri_4_1/cost' in runs.files:
        s = MotionPrimitiveSearch(intersection(turn_indicator=1, start_pos=4), cd, mps, margin=cd.radius)
        _check_run(s, runs, 'mod_pri_4_1/', sorted(mps))
    cd, mps = _setup()
    s = MotionPrimitiveSearch(intersection(turn_indicator=1, start_pos=4), cd, mps, margin=cd.radius)
    ex = H.gold('expand.npz')
    node = tuple(ex['bic/nodes'][7].tolist())
    pts = s.motion_primitive_at('left2', node)
    assert pts.shape == mps['left2'].points.shape
    k = sorted(mps).index('left2')
    assert np.abs(pts[-1, :2] - ex['bic/nbr'][7, k, :2]).max() < 1e-12       # last point == successor pose
    cc = s.collision_checking_points_at('left2', node)
    assert cc.shape == (10, 3)
    with pytest.raises(Exception, match='No solution found'):
        from mpc_for_av_at_intersection_amd.lib.obstacles import BoxObstacle
        sc = intersection(turn_indicator=1, start_pos=4)
        sc.obstacles.append(BoxObstacle(xy_width=(200, 200), height=1, xy_center=(0, 0)))   # everything blocked
        MotionPrimitiveSearch(sc, cd, mps, margin=cd.radius).run()


def test_check_collision_and_linalg_functions():
    """lib.obstacles.check_collision / lib.linalg.transform_2d_pts (rows a10, a11) through the product call surface against
    the COMMITTED reference outputs: the half-plane tables of tests/golden/scenarios.npz (to_convex of the reference's own
    obstacle objects) and the per-(node, primitive) collide flags / successor poses of tests/golden/expand.npz
    (neighbor_function of the reference on its own nodes)."""
    from mpc_for_av_at_intersection_amd.lib.linalg import create_2d_transform_mtx, transform_2d_pts
    from mpc_for_av_at_intersection_amd.lib.obstacles import check_collision
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search_modified import MotionPrimitiveSearch
    from mpc_for_av_at_intersection_amd.lib.scenario import intersection
    cd, mps = _setup()
    ex, sc = H.gold('expand.npz'), H.gold('scenarios.npz')
    sp, ti = (int(v) for v in ex['bic/scenario'])          # (start_pos, turn_indicator) of the scenario the flags were recorded on
    tag = 'int_%d_%d' % (sp, ti)
    hp, off = sc[tag + '/hp_bic'], sc[tag + '/hp_off']
    scen = intersection(start_pos=sp, turn_indicator=ti)
    # the product's obstacle objects reproduce the reference's half-plane rows bit for bit
    mine = np.concatenate([o.to_convex(margin=cd.radius) for o in scen.obstacles], axis=0)
    assert np.array_equal(mine, hp)
    s = MotionPrimitiveSearch(scen, cd, mps, margin=cd.radius)
    names = sorted(mps)
    rng = np.random.default_rng(1)
    col = ex['bic/collide']
    # nodes with both outcomes among their primitives are the informative ones
    mixed = np.nonzero((col.min(axis=1) == 0) & (col.max(axis=1) == 1))[0]
    checked = hits = 0
    for i in rng.choice(mixed, 6, replace=False):
        node = tuple(ex['bic/nodes'][i].tolist())
        for k in rng.choice(len(names), 3, replace=False):
            pts = s.collision_checking_points_at(names[k], node)                     # transform_2d_pts of the collision template
            flag = any(check_collision(hp[off[o]:off[o + 1]], pts[:, :2].T) for o in range(len(off) - 1))
            assert flag == bool(col[i, k]), (i, names[k])
            checked += 1; hits += flag
            # successor pose = transform of the primitive's last point (motion_primitive_search.py:110-113)
            m = create_2d_transform_mtx(*node)
            last = transform_2d_pts(node[2], m, mps[names[k]].points[-1:].copy())
            assert np.abs(last[0, :2] - ex['bic/nbr'][i, k, :2]).max() < 1e-12
    assert 0 < hits < checked
    # degenerate pose (0, 0, theta): the reference builds a 2x2 rotation-only matrix (linalg.py:13-17)
    m = create_2d_transform_mtx(0.0, 0.0, -1.1)
    assert m.shape == (2, 2)
    pts = mps['straight'].points[:5].copy()
    out = transform_2d_pts(-1.1, m, pts)
    c, sn = np.cos(-1.1), np.sin(-1.1)
    assert np.abs(out[:, 0] - (c * pts[:, 0] - sn * pts[:, 1])).max() < 1e-14 and np.abs(out[:, 2] - (pts[:, 2] - 1.1)).max() < 1e-14
    with pytest.raises(AssertionError):
        check_collision(hp[:4, :2], pts[:, :2].T)                                    # shape assertions of obstacles.py:166-170


@pytest.mark.parametrize('T,max_iter', [(10, 1), (13, 1), (20, 1), (13, 2)])
def test_stock_closed_loop_matches_reference_run(T, max_iter):
    """main/scenarios/mpc_intersection.py:95-159 driven with the product classes; golden = the reference's own loop with
    the oracle QP substituted for ECOS (tests/golden/make_golden.py --stage closedloop).
    max_iter = 2 (closedloop_iter2.npz, 60 steps): the reference's successive linearisation (lib/mpc.py:226-237, MAX_ITER passes per
    step, the second one's reference window spaced by the first one's speeds) -- the stock mpc_config.json has MAX_ITER = 1."""
    import mpc_for_av_at_intersection_amd.lib.mpc as pmpc
    from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions
    from mpc_for_av_at_intersection_amd.lib.collision_avoidance import check_collision_moving_cars, get_cutoff_curve_by_position_idx
    from mpc_for_av_at_intersection_amd.lib.moving_obstacles_prediction import MovingObstaclesPrediction
    from mpc_for_av_at_intersection_amd.lib.simulation import HistorySimulation, Simulation, State
    from mpc_for_av_at_intersection_amd.lib.trajectories import calc_nearest_index_in_direction, resample_curve
    g = H.gold('closedloop.npz' if max_iter == 1 else 'closedloop_iter%d.npz' % max_iter)
    tape = H.gold('moving.npz')['traffic/tape']
    pmpc.T = T
    pmpc.Qf = np.diag([1.0, 1.0, 0.0, 0.5]) * T
    pmpc.MAX_ITER = max_iter
    try:
        DT = 0.2
        cd = BicycleModelDimensions()
        full = H.gold('mpc_pre.npz')['path_4_1'].copy()
        dl = np.linalg.norm(full[0, :2] - full[1, :2])
        mpc = pmpc.MPC(cx=full[:, 0], cy=full[:, 1], cyaw=full[:, 2], dl=dl, dt=DT, car_dimensions=cd)
        assert np.array_equal(full, g['T%d/full' % T])            # smooth_yaw mutated the caller's array like the reference
        state = State(x=full[0, 0], y=full[0, 1], yaw=full[0, 2], v=0.0)
        sim = HistorySimulation(car_dimensions=cd, sample_time=DT, initial_state=state)
        margin = 4 * int(np.ceil(cd.radius / dl))
        tidx, tmp = 0, None
        n_steps = int(g['T%d/steps' % T])
        worst = 0.0
        for i in range(n_steps):
            assert not mpc.is_goal(state)
            assert np.abs(np.array([state.x, state.y, state.v, state.yaw]) - g['T%d/state' % T][i]).max() < 1e-6
            worst = max(worst, np.abs(np.array([state.x, state.y, state.v, state.yaw]) - g['T%d/state' % T][i]).max())
            if tmp is None or np.any(tmp[tidx, :] != tmp[-1, :]):
                tidx = calc_nearest_index_in_direction(state, full[:, 0], full[:, 1], start_index=tidx, forward=True)
            assert tidx == g['T%d/tidx' % T][i]
            tres = traj = full[tidx:]
            if state.v < Simulation.MAX_SPEED:
                rdl = DT * np.minimum(np.cumsum(np.zeros(tres.shape[0]) + pmpc.MAX_ACCEL) + state.v, Simulation.MAX_SPEED)
                tres = resample_curve(tres, dl=rdl)
            else:
                tres = resample_curve(tres, dl=DT * Simulation.MAX_SPEED)
            trajs = [np.vstack(MovingObstaclesPrediction(*six, sample_time=DT, car_dimensions=cd).state_prediction(7.)).T for six in tape[i]]
            hit = check_collision_moving_cars(cd, tres, traj, trajs, frame_window=20)
            if hit is not None:
                cut = max(tidx + 1, get_cutoff_curve_by_position_idx(full, hit[0], hit[1]) - margin)
                tmp = full[:cut]
                assert hit[2] == g['T%d/hit' % T][i][2] and cut == g['T%d/cut' % T][i]
            else:
                tmp = full
                assert g['T%d/hit' % T][i][2] < 0
            mpc.set_trajectory_fromarray(tmp)
            delta, acc = mpc.step(state)
            assert mpc.status == 0 and mpc.target_ind == g['T%d/target' % T][i]
            assert np.abs(np.array([delta, acc]) - g['T%d/ctrl' % T][i]).max() < 1e-6
            assert np.abs(mpc.xref - g['T%d/xref' % T][max_iter * i + max_iter - 1]).max() == 0.0       # the LAST pass's window, bit for bit
            state = sim.step(a=acc, delta=delta, xref_deviation=mpc.get_current_xref_deviation())
        assert mpc.is_goal(state) or max_iter > 1              # (the MAX_ITER = 2 golden stops after 60 steps)
        print('T=%d MAX_ITER=%d closed loop: %d steps, worst state deviation %.2e' % (T, max_iter, n_steps, worst))
    finally:
        pmpc.T = 13
        pmpc.Qf = np.diag([1.0, 1.0, 0.0, 0.5]) * 13
        pmpc.MAX_ITER = 1


def test_mpc_failure_path_commands_max_decel(capsys):
    import mpc_for_av_at_intersection_amd.lib.mpc as pmpc
    from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions
    from mpc_for_av_at_intersection_amd.lib.simulation import State
    full = H.gold('mpc_pre.npz')['path_4_1'].copy()
    mpc = pmpc.MPC(cx=full[:, 0], cy=full[:, 1], cyaw=full[:, 2], dl=0.083, dt=0.2, car_dimensions=BicycleModelDimensions())
    d0, a0 = mpc.step(State(x=full[0, 0], y=full[0, 1], yaw=full[0, 2], v=1.0))
    assert mpc.status == 0
    d1, a1 = mpc.step(State(x=full[3, 0], y=full[3, 1], yaw=full[3, 2], v=9.5))    # v > MAX_SPEED: infeasible (mpc.py:187)
    assert mpc.status == 2 and a1 == pmpc.MAX_DECEL and d1 == d0 and mpc.oa is None and mpc.odelta is None
    assert 'Cannot solve mpc' in capsys.readouterr().err


def test_speed_reference_variant():
    """lib/mpc_with_speed: window with the speed profile (golden from the reference), QP with a speed weight vs the oracle"""
    import mpc_for_av_at_intersection_amd.lib.mpc_with_speed as ws
    from mpc_for_av_at_intersection_amd.lib._session import context
    from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions
    from mpc_for_av_at_intersection_amd.lib.simulation import State
    from oracle import oracle_py as orc
    g = H.gold('mpc_pre.npz')
    ctx = context()
    full = H.smoothed_path(4, 1)
    dl = float(np.linalg.norm(full[0, :2] - full[1, :2]))
    cd = BicycleModelDimensions()
    for T in (13, 20):
        ws.T = T
        ws.Qf = np.diag([1.0, 1.0, 0., 0.5]) * T
        try:
            ctx.set_mpc_params(ws._params(cd, 0.2))
            n = len(g['ws%d/state' % T])
            vmax = float(g['ws/MAX_SPEED'])
            paths, pvs, off, ln, cur = [], [], [], [], 0
            for k in range(n):
                cut, cutoff = int(g['ws%d/cut' % T][k]), int(g['ws%d/cutoff' % T][k])
                cv = np.full(cut, vmax)
                if cutoff != 999:
                    cv[cutoff:] = 0
                paths.append(full[:cut]); pvs.append(cv); off.append(cur); ln.append(cut); cur += cut
            tind = ctx.i32(g['ws%d/start' % T])
            st = ctx.f64(g['ws%d/state' % T])
            pre = ctx.prepare(st, None, ctx.f64(np.concatenate(paths)), ctx.i32(off), ctx.i32(ln), dl, tind, path_v=ctx.f64(np.concatenate(pvs)))
            sol = ctx.qp_solve(st, pre['xref'], pre['xbar'], pre['reaches_end'])
            ctx.synchronize()
            assert np.array_equal(tind.cpu().numpy(), g['ws%d/target_ind' % T])
            assert np.array_equal(pre['xref'].cpu().numpy(), g['ws%d/xref' % T])          # incl. xref[2,:] = cv[idx]
            po = orc.MpcParams(T=T, w_perp=10., w_para=1., Q_v_yaw=(20, 0.5), max_decel=-5)
            xb = pre['xbar'].cpu().numpy(); u = sol['u'].cpu().numpy(); status = sol['status'].cpu().numpy()
            for k in range(0, n, 3):
                o = orc.qp_solve(po, g['ws%d/state' % T][k], g['ws%d/xref' % T][k], xb[k], g['ws%d/reaches_end' % T][k])
                assert o.status == status[k] == 0 and np.abs(o.u - u[k]).max() < 2e-7
        finally:
            ws.T = 13
            ws.Qf = np.diag([1.0, 1.0, 0., 0.5]) * 13
    # object API: constructor with cv, set_trajectory_fromarray(traj, cutoff_idx) rebuilds the profile
    m = ws.MPC(cx=full[:, 0], cy=full[:, 1], cv=np.full(len(full), ws.MAX_SPEED), cyaw=full[:, 2].copy(), dl=dl, car_dimensions=cd)
    d0, a0 = m.step(State(x=full[0, 0], y=full[0, 1], yaw=full[0, 2], v=2.0))
    assert m.status == 0 and a0 > 0.5 and np.all(m.xref[2] == ws.MAX_SPEED)      # below the speed reference: accelerate
    m.set_trajectory_fromarray(full[:200], cutoff_idx=20)
    m.step(State(x=full[5, 0], y=full[5, 1], yaw=full[5, 2], v=5.0))
    assert m.status == 0 and m.ai < 0 and m.xref[2].min() == 0.0                  # zero speed reference ahead: brake


def _world_cases():
    import os
    path = os.path.join(H.GOLD, 'astar_worlds.npz')
    if not os.path.exists(path):
        return []
    return sorted({k.rsplit('/', 1)[0] for k in np.load(path).files})


@pytest.mark.parametrize('case', _world_cases())
def test_search_on_other_worlds_and_remaining_variants(case):
    """Golden runs of the reference on the non-stock worlds (roundabouts, T-intersection, multi-lane intersection) and with
    the `_roundabout` / `_single_lane` search variants: cost, primitive ids and expansion ORDER exact.  The worlds come from
    data/worlds.npz through lib/scenario.py."""
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import MotionPrimitiveSearch
    from mpc_for_av_at_intersection_amd.lib.scenario import world
    tag, key = case.split('|')
    variant = {'round': 'roundabout', 'single': 'single_lane', 'base': 'base', 'ml': 'multi_lane'}[tag]
    kw = dict(wh_obstacle=0.2, wc_center=0.02) if (tag == 'ml' and key.endswith('3_2_1_2_3')) else {}
    cd, mps = _setup()
    s = MotionPrimitiveSearch(world(key), cd, mps, margin=cd.radius, variant=variant, **kw)
    runs = H.gold('astar_worlds.npz')
    pre = case + '/'
    cost, path, traj = s.run(debug=True)
    assert cost == float(runs[pre + 'cost'])
    assert len(path) == len(runs[pre + 'path']) and np.abs(np.array(path) - runs[pre + 'path']).max() < COORD_TOL
    names = sorted(mps)
    assert [names.index(s._points_to_mp_names[a, b]) for a, b in zip(path[:-1], path[1:])] == runs[pre + 'seq'].tolist()
    dbg = s.debug_data
    assert len(dbg) == len(runs[pre + 'dbg_g'])
    assert np.abs(np.array([d.node for d in dbg]) - runs[pre + 'dbg_node']).max() < COORD_TOL
    assert np.array_equal(np.array([d.g for d in dbg]), runs[pre + 'dbg_g'])
    assert np.abs(traj - runs[pre + 'traj']).max() < COORD_TOL


def test_plan_many_runs_searches_concurrently_and_exactly():
    """plan_many: the 12 stock routes (12 different obstacle sets) + two `base`-heuristic searches advanced in lock-step, one
    mpcx_expand_multi_batch launch per level for all of them: every search replays its golden run (cost, primitive ids,
    expansion order) exactly, with far fewer launches than running them one after the other."""
    import time
    from mpc_for_av_at_intersection_amd.lib import _session
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import MotionPrimitiveSearch, plan_many
    from mpc_for_av_at_intersection_amd.lib.scenario import intersection
    cd, mps = _setup()
    names = sorted(mps)
    runs = H.gold('astar_runs.npz')
    cases = [('modified', 'mod_bic_%d_%d/' % (sp, ti), sp, ti) for sp in (1, 2, 3, 4) for ti in (1, 2, 3)]
    cases += [('base', 'base_bic_4_1/', 4, 1), ('base', 'base_bic_3_2/', 3, 2)]
    make = lambda: [MotionPrimitiveSearch(intersection(turn_indicator=ti, start_pos=sp), cd, mps, margin=cd.radius, variant=v) for v, _, sp, ti in cases]
    searches = make()
    ctx = _session.context()
    ctx.synchronize(); t0 = time.perf_counter()
    results = plan_many(searches, debug=True)
    t_many = time.perf_counter() - t0
    for s, (v, pre, sp, ti), (cost, path, traj) in zip(searches, cases, results):
        assert cost == float(runs[pre + 'cost'])
        assert len(path) == len(runs[pre + 'path']) and np.abs(np.array(path) - runs[pre + 'path']).max() < COORD_TOL
        assert [names.index(s._points_to_mp_names[a, b]) for a, b in zip(path[:-1], path[1:])] == runs[pre + 'seq'].tolist()
        dbg = s.debug_data
        assert len(dbg) == len(runs[pre + 'dbg_g'])
        assert np.abs(np.array([d.node for d in dbg]) - runs[pre + 'dbg_node']).max() < COORD_TOL
        assert np.array_equal(np.array([d.g for d in dbg]), runs[pre + 'dbg_g'])
        assert np.abs(traj - runs[pre + 'traj']).max() < COORD_TOL
    rounds = max(s.kernel_launches for s in searches)              # every search counts the launches it took part in
    solo = make()
    t0 = time.perf_counter()
    for s in solo:
        s.run(debug=True)
    t_solo = time.perf_counter() - t0
    # the longest search sets the number of rounds; all the others ride along in its launches
    assert rounds <= 1.1 * max(s.kernel_launches for s in solo) + 4 and rounds < sum(s.kernel_launches for s in solo)
    print('plan_many: %d searches, %d launch rounds, %.1f ms; one after the other: %d launches, %.1f ms' % (
        len(searches), rounds, 1e3 * t_many, sum(s.kernel_launches for s in solo), 1e3 * t_solo))
