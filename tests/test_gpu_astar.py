"""Device-resident best-first search (csrc/mpcx_astar.hip, lib.motion_primitive_search.plan_many_device): open list, closed set
and successor generation on the GPU, one wavefront per search, against the golden runs of the reference's own search
(main/lib/a_star.py:31-78 + motion_primitive_search*.py; tests/golden/astar_runs.npz from make_golden.py): cost, path, primitive ids
and THE EXPANSION ORDER, node for node."""
import time

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _setup(version='bicycle_model'):
    from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions, PriusDimensions
    from mpc_for_av_at_intersection_amd.lib.motion_primitive import load_motion_primitives
    cd = BicycleModelDimensions() if version == 'bicycle_model' else PriusDimensions()
    return cd, load_motion_primitives(version)


def _golden_cases():
    cases = [('modified', 'mod_bic_%d_%d/' % (sp, ti), sp, ti, 'bicycle_model') for sp in (1, 2, 3, 4) for ti in (1, 2, 3)]
    cases += [('base', 'base_bic_1_1/', 1, 1, 'bicycle_model'), ('base', 'base_bic_2_3/', 2, 3, 'bicycle_model'),
              ('base', 'base_bic_3_2/', 3, 2, 'bicycle_model'), ('base', 'base_bic_4_1/', 4, 1, 'bicycle_model')]
    return cases


def _make(cases):
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import MotionPrimitiveSearch
    from mpc_for_av_at_intersection_amd.lib.scenario import intersection
    out = []
    for v, _, sp, ti, ver in cases:
        cd, mps = _setup(ver)
        out.append(MotionPrimitiveSearch(intersection(turn_indicator=ti, start_pos=sp), cd, mps, margin=cd.radius, variant=v))
    return out


def _check(s, runs, pre, cost, path, traj):
    names = sorted(s._mps)
    assert cost == float(runs[pre + 'cost'])
    gp = runs[pre + 'path']
    assert len(path) == len(gp) and np.array_equal(np.array(path), gp)                         # bit for bit: the host's cos / sin table
    assert [names.index(s._points_to_mp_names[a, b]) for a, b in zip(path[:-1], path[1:])] == runs[pre + 'seq'].tolist()
    dbg = s.debug_data
    assert len(dbg) == len(runs[pre + 'dbg_g']), (pre, len(dbg), len(runs[pre + 'dbg_g']))     # same number of expansions ...
    assert np.array_equal(np.array([d.node for d in dbg]), runs[pre + 'dbg_node'])             # ... of the same nodes in the same order
    assert np.array_equal(np.array([d.predecessor for d in dbg]), runs[pre + 'dbg_pred'])
    assert np.array_equal(np.array([d.g for d in dbg]), runs[pre + 'dbg_g'])
    assert np.abs(np.array([d.h for d in dbg]) - runs[pre + 'dbg_h']).max() < 1e-11            # h is logged as f - g
    assert traj.shape == runs[pre + 'traj'].shape and np.abs(traj - runs[pre + 'traj']).max() < 1e-12


def test_device_search_replays_the_golden_runs():
    """16 golden searches (12 stock routes with the `modified` heuristic, 4 with the base one incl. the 1911-expansion tie-heavy one) in
    ONE launch + the few re-runs the heuristic check asks for"""
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import plan_many_device
    runs = H.gold('astar_runs.npz')
    cases = _golden_cases()
    searches = _make(cases)
    results, info = plan_many_device(searches, debug=True, first_expansions=4096)     # (capacity escalation has its own test below)
    for s, c, (cost, path, traj) in zip(searches, cases, results):
        _check(s, runs, c[1], cost, path, traj)
    print('device search: %d searches, %d launches, rounds per search %s, %d heuristic overrides, %d table headings, expansions %s'
          % (len(searches), info['launches'], info['rounds'], info['overrides'], info['table_headings'], info['expansions']))
    assert info['launches'] <= 4 and max(info['rounds']) <= 3
    assert all(r == 1 for r, c in zip(info['rounds'], cases) if c[0] == 'base')      # the base heuristic is exact on the device: never re-run


def test_device_search_prius():
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import MotionPrimitiveSearch, plan_many_device
    from mpc_for_av_at_intersection_amd.lib.scenario import intersection
    runs = H.gold('astar_runs.npz')
    cd, mps = _setup('prius')
    searches = [MotionPrimitiveSearch(intersection(turn_indicator=ti, start_pos=sp), cd, mps, margin=cd.radius, variant='modified') for sp, ti in ((4, 1), (1, 2))]
    results, info = plan_many_device(searches, debug=True)
    for s, pre, (cost, path, traj) in zip(searches, ('mod_pri_4_1/', 'mod_pri_1_2/'), results):
        _check(s, runs, pre, cost, path, traj)


def test_1024_concurrent_searches_beat_the_host_queues():
    """1024 searches (the stock routes and base-heuristic searches, replicated) in one device launch against plan_many's host queues on a
    sample of them: same answers, and the whole batch in less time than the host needs for its sample scaled up"""
    from mpc_for_av_at_intersection_amd.lib import _session
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import plan_many, plan_many_device
    cases = [c for c in _golden_cases() if c[1] != 'base_bic_3_2/']          # (the 1911-expansion search once is enough)
    many = [cases[i % len(cases)] for i in range(1024)]
    searches = _make(many)
    ctx = _session.context()
    plan_many_device(_make(cases[:2]))                                        # warm-up (module load)
    ctx.synchronize(); t0 = time.perf_counter()
    results, info = plan_many_device(searches)
    ctx.synchronize(); t_dev = time.perf_counter() - t0
    sample = _make(cases)
    t0 = time.perf_counter()
    ref = plan_many(sample)
    t_host = time.perf_counter() - t0
    for i, (cost, path, traj) in enumerate(results):
        rc, rp, rt = ref[i % len(cases)]
        assert cost == rc and path == rp and np.array_equal(traj, rt)
    print('1024 device-resident searches: %.1f ms in %d launches (%d heuristic overrides; heading table %.1f ms, device %.1f ms, host check + results %.1f ms); '
          'plan_many on %d of them: %.1f ms => %.1f ms for 1024'
          % (1e3 * t_dev, info['launches'], info['overrides'], 1e3 * info['t_closure'], 1e3 * info['t_device'], 1e3 * info['t_check'], len(cases), 1e3 * t_host,
             1e3 * t_host * 1024 / len(cases)))
    assert t_dev < t_host * 1024 / len(cases)


def test_heuristic_override_round_trip():
    """The override mechanism (host finds heuristic values that differ from the reference's bits -> override table -> the searches concerned
    run again) forced: the 'reference' heuristic is bent by a node-dependent amount, on the host check and in the host search alike; the
    device search must then reproduce the HOST search under the bent heuristic, expansion for expansion."""
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import plan_many_device
    cases = [c for c in _golden_cases() if c[0] == 'modified'][:4]
    dev, host = _make(cases), _make(cases)

    def bend(x, y):              # a few % of the nodes, by much more than an ulp so that the pop order really changes
        return np.where(np.floor(np.abs(x * 3.0 + y * 5.0)) % 7 == 0, 0.05, 0.0)
    for s in dev + host:
        ref_h, d2g = s._reference_h, s.distance_to_goal
        s._reference_h = (lambda nodes, f=ref_h: f(nodes) + bend(nodes[:, 0], nodes[:, 1]))
        s.distance_to_goal = (lambda node, f=d2g: float(f(node) + bend(np.float64(node[0]), np.float64(node[1]))))
    results, info = plan_many_device(dev, debug=True)
    assert info['overrides'] > 0 and max(info['rounds']) >= 2
    changed = 0
    for sd, sh, c, (cost, path, traj) in zip(dev, host, cases, results):
        hc, hp, _ = sh.run(debug=True)
        assert cost == hc and path == hp
        dd, hd = sd.debug_data, sh.debug_data
        assert [d.node for d in dd] == [d.node for d in hd] and [d.g for d in dd] == [d.g for d in hd]
        changed += len(hd) != len(H.gold('astar_runs.npz')[c[1] + 'dbg_g'])
    assert changed > 0           # the bent heuristic did change the searches, i.e. the overrides mattered
    print('override round trip: %d overrides, rounds %s' % (info['overrides'], info['rounds']))


def test_heading_outside_the_table_is_added_and_the_search_rerun():
    """A heading table too shallow for the search (closure depth 3): the kernel reports the missing heading, the host adds its closure and
    runs the search again -- until the golden run comes out."""
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import plan_many_device
    runs = H.gold('astar_runs.npz')
    cases = [c for c in _golden_cases() if c[1] in ('mod_bic_1_2/', 'mod_bic_2_1/')]
    searches = _make(cases)
    results, info = plan_many_device(searches, closure_depth=3, max_rounds=40, debug=True)
    assert max(info['rounds']) >= 2
    for s, c, (cost, path, traj) in zip(searches, cases, results):
        _check(s, runs, c[1], cost, path, traj)


# ---------------------------------------------------------------------------------------------------------------------------------------
# round 4: the three remaining variants on the device, per-search overrides, capacities that grow

def _world_cases():
    runs = H.gold('astar_worlds.npz')
    return sorted({k.rsplit('/', 1)[0] for k in runs.files})


def _world_search(case):
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import MotionPrimitiveSearch
    from mpc_for_av_at_intersection_amd.lib.scenario import world
    tag, key = case.split('|')
    variant = {'round': 'roundabout', 'single': 'single_lane', 'base': 'base', 'ml': 'multi_lane'}[tag]
    kw = dict(wh_obstacle=0.2, wc_center=0.02) if (tag == 'ml' and key.endswith('3_2_1_2_3')) else {}
    cd, mps = _setup()
    return MotionPrimitiveSearch(world(key), cd, mps, margin=cd.radius, variant=variant, **kw)


def _check_world(s, runs, pre, cost, path, traj):
    names = sorted(s._mps)
    assert cost == float(runs[pre + 'cost'])
    gp = runs[pre + 'path']
    assert len(path) == len(gp) and np.array_equal(np.array(path), gp)
    assert [names.index(s._points_to_mp_names[a, b]) for a, b in zip(path[:-1], path[1:])] == runs[pre + 'seq'].tolist()
    dbg = s.debug_data
    assert len(dbg) == len(runs[pre + 'dbg_g']), (pre, len(dbg), len(runs[pre + 'dbg_g']))
    assert np.array_equal(np.array([d.node for d in dbg]), runs[pre + 'dbg_node'])             # the reference's expansion order, node for node
    assert np.array_equal(np.array([d.g for d in dbg]), runs[pre + 'dbg_g'])                   # g = chain of EDGE VALUES: bit for bit
    assert traj.shape == runs[pre + 'traj'].shape and np.abs(traj - runs[pre + 'traj']).max() < 1e-12


def test_device_search_all_variants_on_the_other_worlds():
    """multi_lane (default weights AND wh_obstacle / wc_center switched on), roundabout, single_lane and base searches on the roundabouts, the
    T-intersection and the multi-lane intersection -- the reference's golden runs (tests/golden/astar_worlds.npz) -- in ONE device launch
    + re-runs: cost, path, primitive ids, expansion order and every g exact.  Reference: motion_primitive_search_multi_lane.py:155-181,226-237,
    _roundabout.py:131-157,212, _single_lane.py:145-162,218."""
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import plan_many_device
    runs = H.gold('astar_worlds.npz')
    cases = _world_cases()
    assert {c.split('|')[0] for c in cases} >= {'ml', 'round', 'single', 'base'}
    searches = [_world_search(c) for c in cases]
    results, info = plan_many_device(searches, debug=True)
    for s, c, (cost, path, traj) in zip(searches, cases, results):
        _check_world(s, runs, c + '/', cost, path, traj)
    print('device search, other worlds: %d searches, %d launches, rounds %s, %d overrides, expansions %s'
          % (len(searches), info['launches'], info['rounds'], info['overrides'], info['expansions']))


def test_device_search_multi_lane_on_the_stock_intersection():
    """the multi_lane goldens of astar_runs.npz (default weights, and mlw: wh_obstacle = 0.2, wh_center = 0.1, wc_center = 0.05 -- every term of
    the heuristic and of the edge value switched on) replayed on the device"""
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import MotionPrimitiveSearch, plan_many_device
    from mpc_for_av_at_intersection_amd.lib.scenario import intersection
    runs = H.gold('astar_runs.npz')
    cd, mps = _setup()
    cases = [('ml_bic_4_1/', 4, 1, {}), ('ml_bic_1_2/', 1, 2, {}), ('ml_bic_2_3/', 2, 3, {}),
             ('mlw_bic_4_1/', 4, 1, dict(wh_obstacle=0.2, wh_center=0.1, wc_center=0.05))]
    searches = [MotionPrimitiveSearch(intersection(turn_indicator=ti, start_pos=sp), cd, mps, margin=cd.radius, variant='multi_lane', **kw) for _, sp, ti, kw in cases]
    results, info = plan_many_device(searches, debug=True)
    for s, c, (cost, path, traj) in zip(searches, cases, results):
        _check(s, runs, c[0], cost, path, traj)
    print('multi_lane on the device: rounds %s, %d overrides' % (info['rounds'], info['overrides']))


def test_edge_value_override_round_trip():
    """The override mechanism for EDGE values forced (they enter g and therefore the key): the host's reference edge value and the host
    search's edge value are bent alike by an amount that depends on the successor; the device search must reproduce the bent HOST search
    expansion for expansion, through overrides keyed by (parent, primitive)."""
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import plan_many_device
    cases = ['round|roundabout/1_3', 'single|intersection/4_1', 'ml|intersection_multi_lanes/1_1_2_1_2']

    def bend(x, y):
        return np.where(np.floor(np.abs(x * 3.0 + y * 5.0)) % 5 == 0, 0.07, 0.0)
    dev, host = [_world_search(c) for c in cases], [_world_search(c) for c in cases]
    for s in dev:
        s._reference_edge = (lambda par, ch, kk, f=type(s)._reference_edge, s=s: f(s, par, ch, kk) + bend(ch[:, 0], ch[:, 1]))
    for s in host:
        nf = s.neighbor_function
        s._a_star.neighbor_function = (lambda node, nf=nf: ((float(c + bend(np.float64(nb[0]), np.float64(nb[1]))), nb) for c, nb in nf(node)))
    results, info = plan_many_device(dev, debug=True)
    assert info['overrides'] > 0 and max(info['rounds']) >= 2
    for sd, sh, (cost, path, traj) in zip(dev, host, results):
        hc, hp, _ = sh.run(debug=True)
        assert cost == hc and path == hp
        assert [d.node for d in sd.debug_data] == [d.node for d in sh.debug_data] and [d.g for d in sd.debug_data] == [d.g for d in sh.debug_data]
    print('edge override round trip: %d overrides, rounds %s' % (info['overrides'], info['rounds']))


def test_overrides_are_per_search():
    """searches that share their start but not their goal must not hand each other heuristic overrides.  The three stock routes
    out of every arm share a start pose; the heuristic is bent with an amount that depends on the GOAL, so that the same node needs a
    different override in every search: all of them settle in two or three rounds (the re-run with overrides can meet new nodes; one
    table for the whole batch needed one launch per contradicting search and gave up after six), and a `base` search in the same batch
    runs once, untouched."""
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import plan_many_device
    cases = [c for c in _golden_cases() if c[0] == 'modified'] + [c for c in _golden_cases() if c[1] == 'base_bic_4_1/']
    dev, host = _make(cases), _make(cases)
    for s in dev + host:
        if s.variant == 'base':
            continue
        amp = 0.02 + 0.01 * (abs(s._goal_point[0]) % 3 + abs(s._goal_point[1]) % 5)
        bend = (lambda x, y, amp=amp: np.where(np.floor(np.abs(x * 3.0 + y * 5.0)) % 4 == 0, amp, 0.0))
        s._reference_h = (lambda nodes, f=type(s)._reference_h, s=s, bend=bend: f(s, nodes) + bend(nodes[:, 0], nodes[:, 1]))
        s.distance_to_goal = (lambda node, f=s.distance_to_goal, bend=bend: float(f(node) + bend(np.float64(node[0]), np.float64(node[1]))))
    results, info = plan_many_device(dev, debug=True, first_expansions=4096)
    assert 2 <= max(info['rounds']) <= 3 and info['launches'] <= 3, info['rounds']
    assert info['rounds'][-1] == 1                                   # the base search
    for sd, sh, (cost, path, traj) in zip(dev, host, results):
        hc, hp, _ = sh.run(debug=True)
        assert cost == hc and path == hp
        assert [d.node for d in sd.debug_data] == [d.node for d in sh.debug_data]


def test_capacities_grow_on_demand():
    """nothing is truncated silently.  A path longer than path_cap ends in MPCX_ASTAR_PATH_CAPACITY and runs again with more room;
    a search that needs more expansions than the first launch holds runs again with four times as many; beyond max_expansions it raises."""
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import plan_many_device
    runs = H.gold('astar_runs.npz')
    cases = [c for c in _golden_cases() if c[1] in ('mod_bic_1_1/', 'mod_bic_2_2/', 'base_bic_3_2/')]       # 293, 13 and 1911 expansions; paths of 13 nodes
    searches = _make(cases)
    results, info = plan_many_device(searches, debug=True, first_expansions=64, path_cap=4)
    for s, c, (cost, path, traj) in zip(searches, cases, results):
        _check(s, runs, c[1], cost, path, traj)
    assert info['rounds'][1] >= 3 and info['rounds'][2] >= 4        # 4 -> 8 -> 16 path nodes; 64 -> 256 -> 1024 -> 4096 expansions
    with pytest.raises(RuntimeError, match='exceeds'):
        plan_many_device(_make(cases[2:]), max_expansions=1024)


def test_c_abi_rejects_bad_search_rows():
    """mpcx_astar_batch validates what it can on the host: unknown variant, override slice outside the table, obstacle-term variant without
    the row norms"""
    import torch
    from mpc_for_av_at_intersection_amd import _lib
    from mpc_for_av_at_intersection_amd.lib import _session
    from mpc_for_av_at_intersection_amd.runtime import MpcxError
    ctx = _session.context()
    s = _make(_golden_cases()[:1])[0]
    cs_t = ctx.f64(np.array([0.0])); cs_v = ctx.f64(np.array([[1.0, 0.0]]))
    base = dict(start=s._start, goal_box=(*s._goal_area.xy1, *s._goal_area.xy2), goal_point=s._goal_point, allowed_dtheta=0.2, variant=_lib.ASTAR_MODIFIED)
    for bad in (dict(variant=7), dict(ov_off=0, ov_cnt=3), dict(variant=_lib.ASTAR_ROUNDABOUT)):
        with pytest.raises(MpcxError):
            ctx.astar_batch([s._model], [dict(base, **bad)], cs_t, cs_v, max_expansions=16)


def test_device_search_equals_the_host_queue_across_worlds_and_variants():
    """Beyond the golden runs: 48 (world, variant, weights) combinations drawn (seeded) from the 228 tabulated worlds of main/envs/*.py --
    roundabouts (octagon half-planes), T-intersection, arterial road, multi-lane intersections -- each searched on the device AND through
    the exact host queue (`MotionPrimitiveSearch.run`, the implementation the golden runs pin): identical cost, path, primitive ids,
    expansion order and g of every expansion, or the same "No solution found." / capacity outcome."""
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import MotionPrimitiveSearch, plan_many_device
    from mpc_for_av_at_intersection_amd.lib.scenario import available_worlds, world
    rng = np.random.default_rng(4)
    keys = available_worlds()
    cd, mps = _setup()
    variants = ['base', 'modified', 'multi_lane', 'roundabout', 'single_lane']
    combos = []
    for n, k in enumerate(rng.choice(len(keys), 48, replace=False)):
        v = variants[n % 5]
        kw = {}
        if v == 'multi_lane' and n % 2:
            kw = dict(wh_obstacle=float(rng.choice([0.1, 0.2])), wh_center=float(rng.choice([0.0, 0.1])), wc_center=float(rng.choice([0.0, 0.03])),
                      wc_steering=float(rng.choice([5.0, 2.0])))
        combos.append((keys[k], v, kw))
    make = lambda: [MotionPrimitiveSearch(world(k), cd, mps, margin=cd.radius, variant=v, **kw) for k, v, kw in combos]
    host = make()
    LIMIT = 1500                                            # expansions: the host queue needs ~0.1 ms .. 2 ms each
    ref = []
    for s in host:
        orig = s._a_star.neighbor_function
        count = [0]

        def limited(node, orig=orig, count=count):
            count[0] += 1
            if count[0] > LIMIT:
                raise OverflowError
            return orig(node)
        s._a_star.neighbor_function = limited
        try:
            ref.append(s.run(debug=True))
        except OverflowError:
            ref.append('capacity')
        except Exception as e:
            assert 'No solution' in str(e)
            ref.append('none')
    keep = [i for i, r in enumerate(ref) if r not in ('capacity', 'none')]
    assert len(keep) >= 25, [r if isinstance(r, str) else 'ok' for r in ref]
    dev = make()
    results, info = plan_many_device([dev[i] for i in keep], debug=True, max_expansions=4096)
    for i, (cost, path, traj) in zip(keep, results):
        hc, hp, ht = ref[i]
        assert cost == hc and path == hp, combos[i]
        dd, hd = dev[i].debug_data, host[i].debug_data
        assert [d.node for d in dd] == [d.node for d in hd], combos[i]
        assert [d.g for d in dd] == [d.g for d in hd] and [d.predecessor for d in dd] == [d.predecessor for d in hd], combos[i]
        assert np.array_equal(traj, ht)
    # searches the host gave up on: the device reports the same outcome
    for i, r in enumerate(ref):
        if r == 'none':
            with pytest.raises(Exception, match='No solution'):
                plan_many_device(make()[i:i + 1], max_expansions=4096)
    print('device == host queue on %d searches (%s); %d without solution, %d beyond %d expansions; launches %d, overrides %d, expansions up to %d'
          % (len(keep), {v: sum(1 for i in keep if combos[i][1] == v) for v in variants}, sum(r == 'none' for r in ref), sum(r == 'capacity' for r in ref),
             LIMIT, info['launches'], info['overrides'], max(info['expansions'])))
