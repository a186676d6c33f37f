"""CPU tests of the host-side logic of the product (no GPU needed): C-ABI library exports, the reference call surface's
host parts (A* queue, set-up geometry, heuristics) against the golden vectors, and the multi-GPU sharding rule."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mpc_for_av_at_intersection_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    hdr = open(os.path.join(ROOT, 'include', 'mpcx.h')).read()
    declared = sorted(set(re.findall(r'\b(mpcx_[a-z_0-9]+)\s*\(', hdr)))
    assert len(declared) >= 16
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib.EXPORTS) == declared
    lib.mpcx_version.restype = ctypes.c_char_p
    assert b'gfx950' in lib.mpcx_version()


def test_struct_layouts_match_header():
    from mpc_for_av_at_intersection_amd import _lib
    assert ctypes.sizeof(_lib.MpcParamsC) == 8 + 8 * 23 + 8 + 8  # 2 int32 + 23 doubles + (model, reserved) + jerk_weight, no padding surprises
    assert ctypes.sizeof(_lib.InteractionParamsC) == 16 + 8 * 9 + 8 + 8 + 8 + 4 * 8 + 8 + 16  # + path_cum pointer and its error bound (round 3), path_first_within, plan_* (round 4)
    assert ctypes.sizeof(_lib.AstarSearchC) == 8 * 20 + 8 + 16 == _lib.ASTAR_SEARCH_DTYPE.itemsize    # 20 doubles, hp_norm, 4 int32 (round 4)
    assert [_lib.ASTAR_SEARCH_DTYPE.fields[n][1] for n, _ in _lib.AstarSearchC._fields_] == [getattr(_lib.AstarSearchC, n).offset for n, _ in _lib.AstarSearchC._fields_]
    from oracle import oracle_py as orc
    assert ctypes.sizeof(orc._CParams) == ctypes.sizeof(_lib.MpcParamsC)


def test_no_gpu_means_loud_failure():
    import torch
    from mpc_for_av_at_intersection_amd.runtime import Context, MpcxError
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(MpcxError):
        Context(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'mpc_for_av_at_intersection_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dp, f)).read()
                assert 'oracle' not in src.replace('oracle/', '').lower() or f in (), (dp, f)


def test_a_star_class_matches_reference_unit_vectors():
    from mpc_for_av_at_intersection_amd.lib.a_star import AStar
    edges = {'Start': [('A', 3), ('B', 1), ('D', 4)], 'A': [('C', 2)], 'B': [('A', 4), ('H', 1), ('E', 6)],
             'C': [('F', 1), ('Goal', 9)], 'D': [('L', 2)], 'E': [('J', 4)], 'F': [('G', 1)], 'G': [('K', 3)],
             'H': [('G', 4), ('I', 6), ('O', 2)], 'I': [('J', 5)], 'J': [('G', 3), ('Goal', 3)], 'K': [('N', 1)],
             'M': [('K', 1), ('Goal', 2)], 'N': [('M', 2)], 'O': [('L', 2)], 'L': [], 'Goal': []}
    a = AStar(lambda n: ((w, m) for m, w in edges[n]))
    g, path = a.run('Start', lambda n: n == 'Goal', lambda n: 0, debug=True)
    assert g == 14 and path == ['Start', 'A', 'C', 'Goal']
    assert [(d.node, d.g, d.predecessor) for d in a.debug_data] == [
        ('Start', 0, 'Start'), ('B', 1, 'Start'), ('H', 2, 'B'), ('A', 3, 'Start'), ('D', 4, 'Start'), ('O', 4, 'H'),
        ('C', 5, 'A'), ('F', 6, 'C'), ('G', 6, 'H'), ('L', 6, 'D'), ('E', 7, 'B'), ('I', 8, 'H'), ('K', 9, 'G'),
        ('N', 10, 'K'), ('J', 11, 'E'), ('M', 12, 'N'), ('Goal', 14, 'C')]
    b = AStar(lambda n: ((1., n + d) for d in (-1, 1)))
    b.run(0, lambda n: n == 10, lambda n: 0, debug=True)
    assert len(b.debug_data) == 21
    b.run(0, lambda n: n == 10, lambda n: abs(n - 10), debug=True)
    assert len(b.debug_data) == 11
    with pytest.raises(Exception, match='No solution found'):
        AStar(lambda n: iter(())).run(0, lambda n: False, lambda n: 0)
    # peek_open never changes results
    c = AStar(lambda n: (c.peek_open(5) and None) or ((1., n + d) for d in (-1, 1)))
    assert c.run(0, lambda n: n == 4, lambda n: abs(n - 4))[0] == 4


def test_setup_geometry_matches_golden():
    from mpc_for_av_at_intersection_amd.lib import car_dimensions as cdm
    from mpc_for_av_at_intersection_amd.lib.motion_primitive import load_motion_primitives
    from mpc_for_av_at_intersection_amd.lib.scenario import intersection
    from mpc_for_av_at_intersection_amd.lib.trajectories import car_trajectory_to_collision_point_trajectories, resample_curve
    tm, sc, prim = H.gold('templates.npz'), H.gold('scenarios.npz'), H.gold('primitives.npz')
    for version, cd in (('bicycle_model', cdm.BicycleModelDimensions()), ('prius', cdm.PriusDimensions())):
        ref = H.car(version)
        assert cd.radius == ref['radius'] and cd.distance_back_to_front_wheel == ref['L']
        assert np.array_equal(cd.circle_centers, np.array(ref['circle_centers']))
        mps = load_motion_primitives(version)
        assert sorted(mps) == H.prim_meta(version)['names']
        for n, mp in mps.items():
            assert np.array_equal(mp.points, prim['%s/%s' % (version, n)])
            pts = resample_curve(mp.points.copy(), dl=cd.radius, keep_last_point=True)
            tpl = np.concatenate(car_trajectory_to_collision_point_trajectories(pts, cd), axis=0)
            assert np.array_equal(tpl, tm['%s/%s' % (version, n)])            # G2: collision templates bit-exact
        tag = 'bic' if version == 'bicycle_model' else 'pri'
        for sp in (1, 2, 3, 4):
            for ti in (1, 2, 3):
                s = intersection(turn_indicator=ti, start_pos=sp)
                key = 'int_%d_%d' % (sp, ti)
                hp = np.concatenate([o.to_convex(margin=cd.radius) for o in s.obstacles], axis=0)
                assert np.array_equal(hp, sc[key + '/hp_' + tag])             # G3: half-plane rows bit-exact
                assert s.start == tuple(sc[key + '/start']) and s.goal_point == tuple(sc[key + '/goal_point'])


def test_host_path_utilities_match_golden():
    from mpc_for_av_at_intersection_amd.lib.simulation import State
    from mpc_for_av_at_intersection_amd.lib.trajectories import calc_nearest_index_in_direction, resample_curve
    from mpc_for_av_at_intersection_amd.lib.mpc import smooth_yaw
    from mpc_for_av_at_intersection_amd.lib.maths import normalize_angle
    mv, pre = H.gold('moving.npz'), H.gold('mpc_pre.npz')
    full = pre['path_4_1']
    for (x, y, s), o in zip(mv['nearest/in'], mv['nearest/out']):
        assert calc_nearest_index_in_direction(State(x=x, y=y), full[:, 0], full[:, 1], start_index=int(s)) == o
    for k in range(6):
        v0, i0 = mv['resample/%d/in' % k]
        tr = full[int(i0):]
        assert np.array_equal(resample_curve(tr, dl=H.ego_resample_dl(len(tr), v0)), mv['resample/%d/out' % k])
    assert np.array_equal(smooth_yaw(pre['smooth_yaw/in'].copy()), pre['smooth_yaw/out'])
    assert normalize_angle(-1e-17) == 0.0 and normalize_angle(np.pi) == -np.pi and abs(normalize_angle(7.0) - (7.0 - 2 * np.pi)) < 1e-15


def test_plant_matches_oracle():
    from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions
    from mpc_for_av_at_intersection_amd.lib.simulation import Simulation, State
    from oracle import oracle_py as orc
    rng = np.random.default_rng(0)
    p = orc.MpcParams(T=5)
    for _ in range(50):
        st = [rng.uniform(-30, 30), rng.uniform(-30, 30), rng.uniform(-5, 8.3), rng.uniform(-4, 4)]
        a, d = rng.uniform(-12, 4), rng.uniform(-1.2, 1.2)
        sim = Simulation(BicycleModelDimensions(), 0.2, State(x=st[0], y=st[1], v=st[2], yaw=st[3]))
        o = sim.step(a, d)
        ref = orc.plant_step(p, st, a, d)
        assert np.abs(np.array([o.x, o.y, o.v, o.yaw]) - ref).max() < 1e-13


def _shard_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mpc_for_av_at_intersection_amd.sharding import shard_instances, gather_agent_states
    import torch
    lo, hi = shard_instances(10, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float64).reshape(-1, 1, 1).repeat(1, 3, 6) + 0.5 * rank * 0
    allst = gather_agent_states(local, 10, rank, world)
    q.put((rank, lo, hi, allst[:, 0, 0].tolist()))
    dist.destroy_process_group()


def test_sharding_world_size_2_gloo():
    """instances shard contiguously; the optional agent-state exchange (north-star's all-gather) reassembles them"""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p_ in procs:
        p_.join(60)
    assert res[0][1:3] == (0, 5) and res[1][1:3] == (5, 10)
    for r in res:
        assert r[3] == [float(i) for i in range(10)]


def test_sensitivity_config_to_tuning_row(tmp_path, monkeypatch):
    """lib/mpc_sensitivity.py: the JSON keys of config/mpc_config_sensitivity.json map onto MpcParams / one mpcx_qp_tuning row"""
    import json
    cfg = {"NX": 4, "NU": 2, "T": 13, "w_perp": 7.0, "w_para": 0.5, "R": [0.02, 0.03], "Rd": [0.04, 2.0], "Q_v_yaw": [0.1, 0.6],
           "Qf": [1.0, 2.0, 0.0, 0.5], "GOAL_DIS": 1.5, "STOP_SPEED": 0.1389, "MAX_TIME": 13.0, "MAX_ITER": 1, "DU_TH": 0.1,
           "MAX_DSTEER": 45.0, "MAX_ACCEL": 1.5, "MAX_DECEL": -6}
    path = tmp_path / 'mpc_config_sensitivity.json'
    path.write_text(json.dumps(cfg))
    monkeypatch.setenv('MPCX_MPC_SENSITIVITY_CONFIG', str(path))
    from mpc_for_av_at_intersection_amd.lib import mpc_sensitivity as ms
    from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions
    p = ms.params_from_config(ms._load(), BicycleModelDimensions(), 0.2, horizon=13)
    row = p.tuning_row()
    assert row.shape == (16,)
    assert np.allclose(row, [7.0, 0.5, 0.02, 0.03, 0.04, 2.0, 0.1, 0.6, 13.0, 26.0, 0.0, 6.5, 1.5, -6.0, np.deg2rad(45.0), 0.0])
    # live reload: editing the file changes the next parameter set
    cfg['w_perp'] = 33.0
    path.write_text(json.dumps(cfg))
    assert ms.params_from_config(ms._load(), BicycleModelDimensions(), 0.2).w_perp == 33.0


def test_scripted_traffic_matches_reference_tapes():
    """lib/moving_obstacles.py against get()/step() tapes recorded from the reference's classes (tests/golden/traffic.npz,
    34 configurations x 150 steps: both directions, turning or not, start delays, plant dt 0.1/0.2): bit-identical"""
    import contextlib
    import io
    import json
    import os
    from mpc_for_av_at_intersection_amd.lib import moving_obstacles as mo
    from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions
    tapes = H.gold('traffic.npz')
    meta = json.load(open(os.path.join(H.GOLD, 'traffic_meta.json')))
    bic = BicycleModelDimensions()
    assert len(meta) == 34
    turned = 0
    for key, kw in meta.items():
        kw = dict(kw)
        cls = getattr(mo, kw.pop('cls'))
        with contextlib.redirect_stdout(io.StringIO()):
            got = cls(bic, **kw).tape(150)
        assert np.array_equal(got, tapes[key]), key
        turned += int(np.ptp(got[:, 3]) > 1.0)
    assert turned >= 8                                            # the turning branches really turn
    assert np.array_equal(tapes['arc_angle'], [mo.calculate_steering_angle_for_radius(r) for r in (4.0, 5.0, 7.5)])
    # the stock scenario's traffic (mpc_intersection.py:49-52) equals the tape the interaction tests replay
    mov = H.gold('moving.npz')
    obs = [mo.MovingObstacleTIntersection(bic, direction=1, offset=2., turning=False, speed=25 / 3.6, dt=0.2),
           mo.MovingObstacleTIntersection(bic, direction=-1, offset=4., turning=True, speed=25 / 3.6, dt=0.2)]
    tape = np.stack([o.tape(120) for o in obs], axis=1)
    assert np.array_equal(tape, mov['traffic/tape'])


def test_world_tables_reproduce_reference_half_planes():
    """lib/scenario.py + data/worlds.npz: the stock intersection worlds rebuilt from the tables give the reference's
    half-plane arrays bit for bit (tests/golden/scenarios.npz holds `to_convex` outputs of the reference's own objects);
    every tabulated world builds; closed-form worlds equal their tabulated twins."""
    from mpc_for_av_at_intersection_amd.lib import scenario as sc
    from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions, PriusDimensions
    gold = H.gold('scenarios.npz')
    for cd, tag in ((BicycleModelDimensions(), 'bic'), (PriusDimensions(), 'pri')):
        for sp in (1, 2, 3, 4):
            for ti in (1, 2, 3):
                w = sc.intersection(turn_indicator=ti, start_pos=sp)
                hp = np.concatenate([o.to_convex(margin=cd.radius) for o in w.obstacles], axis=0)
                assert np.array_equal(hp, gold['int_%d_%d/hp_%s' % (sp, ti, tag)])
                assert np.array_equal(np.array(w.start), gold['int_%d_%d/start' % (sp, ti)])
    names = sc.available_worlds()
    assert len(names) >= 200
    kinds = {n.split('/')[0] for n in names}
    assert {'intersection', 't_intersection', 'roundabout', 'roundabout_big', 'intersection_multi_lanes', 'arterial'} <= kinds
    n_rows = {}
    for n in names:
        w = sc.world(n)
        rows = sum(len(o.to_convex(0.5)) for o in w.obstacles)
        assert rows <= 512 and len(w.obstacles) <= 128          # device table limits of mpcx_search_model_create
        n_rows.setdefault(n.split('/')[0], set()).add(rows)
        assert len(w.start) == 3 and len(w.goal_point) == 3 and w.goal_area.distance_to_point(w.goal_point[:2]) == 0
    assert 128 in n_rows['intersection'] and max(n_rows['roundabout_big']) >= 152 and min(n_rows['t_intersection']) >= 80
    for nl in (1, 2, 3, 4):
        for gl in range(1, nl + 1):
            a = sc.ArterialMultiLanes(num_lanes=nl, goal_lane=gl).create_scenario()
            b = sc.world('arterial/%d_%d' % (nl, gl))
            assert a.start == b.start and a.goal_point == b.goal_point
            assert [o.xy1 + o.xy2 for o in a.obstacles] == [o.xy1 + o.xy2 for o in b.obstacles]
            assert a.goal_area.xy1 + a.goal_area.xy2 == b.goal_area.xy1 + b.goal_area.xy2
    assert sc.ArterialMultiLanes(num_lanes=2, goal_lane=3).create_scenario() is None
    f = sc.free_area(test_no=2, angle=0.3, start_pos=1.0, goal_distance=10)
    assert f.obstacles == [] and f.goal_point[2] == 0 and abs(f.goal_point[0] - (1.0 + 10 * np.cos(0.3))) < 1e-15
    with pytest.raises(KeyError):
        sc.t_intersection(turn_indicator=4, start_pos=4)        # the reference raises KeyError for this combination too


def test_arc_length_table_error_bound_holds():
    """runtime.path_tables: a difference of two table entries is within the stated bound of the reference's own running sum over the same
    steps (np.cumsum restarted at the first of the two points, trajectories.py:72-79) -- the premise of the conflict search's fast pass"""
    from mpc_for_av_at_intersection_amd.runtime import path_tables
    rng = np.random.default_rng(3)
    routes = [np.cumsum(rng.uniform(0.01, 0.4, (n, 3)), axis=0) * rng.uniform(0.5, 3.0) for n in (720, 37, 2, 1500)]
    table = np.concatenate(routes)
    offs = np.cumsum([0] + [len(r) for r in routes])
    cum, err = path_tables(table, offs)
    assert 0 < err < 1e-9
    worst = 0.0
    for a, b in zip(offs[:-1], offs[1:]):
        assert cum[a] == 0.0 and (np.diff(cum[a:b]) >= 0).all()
        for start in rng.integers(a, b, 12):
            pts = table[start:b, :2]
            steps = np.append(0.0, np.linalg.norm(pts[1:] - pts[:-1], axis=1))
            worst = max(worst, np.abs((cum[start:b] - cum[start]) - steps.cumsum()).max())
    assert worst <= err, (worst, err)


def test_vectorised_reference_expressions_equal_the_scalar_call_surface():
    """plan_many_device checks every heuristic / edge value the device search used against `_reference_h` / `_reference_edge` (numpy over
    arrays).  Those must carry the bits of the scalar methods `distance_to_goal` / `neighbor_function` -- the ones the golden runs of the
    reference pin on the GPU (tests/test_gpu_callsurface.py) -- for every variant with computed terms.  No GPU needed: a stub stands in for
    the device model."""
    import numpy as np
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import MotionPrimitiveSearch, _DeviceCheck, _py_square, _POW_WITNESSES
    from mpc_for_av_at_intersection_amd.lib.scenario import world
    from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions
    from mpc_for_av_at_intersection_amd.lib.motion_primitive import load_motion_primitives

    class StubModel:
        n_prim = 9

    class StubCtx:
        def search_model(self, *a):
            return StubModel()
    # the squares: libm pow, not x * x (and the witnesses really are witnesses on this machine)
    assert np.array_equal(_py_square(_POW_WITNESSES), np.array([v ** 2 for v in _POW_WITNESSES.tolist()]))
    assert not np.array_equal(_POW_WITNESSES * _POW_WITNESSES, _py_square(_POW_WITNESSES))
    cd, mps = BicycleModelDimensions(), load_motion_primitives('bicycle_model')
    runs = np.load(os.path.join(ROOT, 'tests', 'golden', 'astar_worlds.npz'))
    rng = np.random.default_rng(0)
    seen = set()
    for case in sorted({k.rsplit('/', 1)[0] for k in runs.files}):
        tag, key = case.split('|')
        if tag == 'base':
            continue
        variant = {'round': 'roundabout', 'single': 'single_lane', 'ml': 'multi_lane'}[tag]
        kw = dict(wh_obstacle=0.2, wh_center=0.1, wc_center=0.02) if (tag == 'ml' and key.endswith('3_2_1_2_3')) else {}
        s = MotionPrimitiveSearch(world(key), cd, mps, margin=cd.radius, variant=variant, ctx=StubCtx(), **kw)
        s._dev_check = _DeviceCheck(s)
        nodes = runs[case + '/dbg_node']
        assert np.array_equal(s._reference_h(nodes), np.array([float(s.distance_to_goal(tuple(n))) for n in nodes.tolist()]))
        children = nodes + rng.normal(0, 1, nodes.shape)
        kk = rng.integers(0, 9, len(nodes))
        scalar = []
        for p, c, k in zip(nodes.tolist(), children.tolist(), kk.tolist()):
            s._cache = {tuple(p): [(k, tuple(c))]}
            scalar.append(float(next(iter(s.neighbor_function(tuple(p))))[0]))
        assert np.array_equal(s._reference_edge(nodes, children, kk), np.array(scalar))
        seen.add((variant, bool(kw)))
    assert seen == {('multi_lane', False), ('multi_lane', True), ('roundabout', False), ('single_lane', False)}


def test_path_first_within_is_the_cutoff_function_of_the_reference():
    """runtime.path_first_within tabulates get_cutoff_curve_by_position_idx(path, *path[k, :2]) (collision_avoidance.py:107-119) for every
    point k of every path: checked here against that function written out point by point, on paths with duplicate and near-duplicate
    points (where the answer is not k itself)."""
    import numpy as np
    from mpc_for_av_at_intersection_amd.runtime import path_first_within
    rng = np.random.default_rng(3)
    paths = []
    for n in (5, 40, 300, 700):
        p = np.cumsum(rng.normal(0.0, 0.05, (n, 2)), axis=0)
        dup = rng.choice(np.arange(1, n), max(1, n // 8), replace=False)
        p[dup] = p[dup - 1] + rng.choice([0.0, 4e-4, 9.9e-4, 1.1e-3], (len(dup), 1)) * np.array([[0.6, 0.8]])   # copies, near copies, just outside
        paths.append(np.column_stack([p, np.zeros(n)]))
    table = np.concatenate(paths)
    offs = np.cumsum([0] + [len(p) for p in paths])
    got = path_first_within(table, offs)
    differs = 0
    for a, b in zip(offs[:-1], offs[1:]):
        pts = table[a:b]
        for k in range(b - a):
            d = pts[:, :2].copy()
            d[:, 0] -= pts[k, 0]; d[:, 1] -= pts[k, 1]
            want = int(np.argmax(np.linalg.norm(d, axis=1) <= 0.001))
            assert got[a + k] == want, (a, k, got[a + k], want)
            differs += want != k
    assert differs > 20


def test_path_plan_is_the_scenario_loops_ego_prediction():
    """runtime.path_plan tabulates, per path point t, the ego prediction of main/scenarios/mpc_intersection.py:107-116 for a saturated speed
    (resample_curve(trajectory_full[t:], dl = DT * MAX_SPEED), trajectories.py:58-86), the disc centres of the kept poses
    (trajectories.py:11-37) and conservative boxes of eight runs of frames: checked against lib.trajectories' own functions (the mirror
    of the reference's that the golden vectors pin) on a stock route and on an unevenly sampled one."""
    import numpy as np
    from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions
    from mpc_for_av_at_intersection_amd.lib.trajectories import car_trajectory_to_collision_point_trajectories, resample_curve
    from mpc_for_av_at_intersection_amd.runtime import path_plan
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'mpc_pre.npz'))
    full = g['path_4_1']
    rng = np.random.default_rng(5)
    thin = full[np.sort(np.concatenate([[0, len(full) - 1], rng.choice(np.arange(1, len(full) - 1), len(full) // 3, replace=False)]))]
    cd = BicycleModelDimensions()
    table = np.concatenate([full, thin])
    offs = np.array([0, len(full), len(full) + len(thin)])
    dt, vmax, steps = 0.2, 30.0 / 3.6, 35
    pl = path_plan(table, np.column_stack([np.cos(table[:, 2]), np.sin(table[:, 2])]), offs, dt, vmax, np.asarray(cd.circle_centers).ravel(), cd.radius, steps)
    assert pl['disc'].shape == (len(table), 64, 4) and pl['box'].shape == (len(table), 8, 4)
    for a, b in zip(offs[:-1], offs[1:]):
        for t in list(range(a, b, 37)) + [b - 2, b - 1]:
            res = resample_curve(table[t:b], dl=dt * vmax)
            k = len(res)
            assert pl['cnt'][t] == k
            front, rear = car_trajectory_to_collision_point_trajectories(res, cd)
            assert np.array_equal(pl['disc'][t, :k, 0:2], front[:, :2]) and np.array_equal(pl['disc'][t, :k, 2:4], rear[:, :2])
            # every disc of every (padded) frame lies inside its run's box with room for 2 * radius
            F = max(k, steps); SL = (F + 7) // 8
            for f in range(F):
                e = pl['disc'][t, min(f, k - 1)]
                bx = pl['box'][t, f // SL]
                assert bx[0] <= e[0::2].min() - 2 * cd.radius and bx[1] >= e[0::2].max() + 2 * cd.radius
                assert bx[2] <= e[1::2].min() - 2 * cd.radius and bx[3] >= e[1::2].max() + 2 * cd.radius
    full_disc = car_trajectory_to_collision_point_trajectories(table, cd)
    assert np.array_equal(pl['path_disc'][:, 0:2], full_disc[0][:, :2]) and np.array_equal(pl['path_disc'][:, 2:4], full_disc[1][:, :2])
