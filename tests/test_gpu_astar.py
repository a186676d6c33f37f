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
    results, info = plan_many_device(searches, debug=True)
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
