"""GPU box: the device-resident search (plan_many_device) against the exact host queue (MotionPrimitiveSearch.run) on EVERY tabulated world of
main/envs/*.py, every variant in turn (multi_lane also with non-default weights): cost, path, expansion order and every g must be identical,
or both must end without a solution / beyond the expansion limit.  Summary for profiles/."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions
from mpc_for_av_at_intersection_amd.lib.motion_primitive import load_motion_primitives
from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import MotionPrimitiveSearch, plan_many_device
from mpc_for_av_at_intersection_amd.lib.scenario import available_worlds, world

LIMIT = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
cd, mps = BicycleModelDimensions(), load_motion_primitives('bicycle_model')
rng = np.random.default_rng(0)
variants = ['base', 'modified', 'multi_lane', 'roundabout', 'single_lane']
combos = []
for n, k in enumerate(available_worlds()):
    for v in (variants[n % 5], variants[(n + 2) % 5]):
        kw = {}
        if v == 'multi_lane' and n % 3 == 0:
            kw = dict(wh_obstacle=float(rng.choice([0.1, 0.2])), wh_center=float(rng.choice([0.0, 0.1])), wc_center=float(rng.choice([0.0, 0.03])))
        combos.append((k, v, kw))
make = lambda cs: [MotionPrimitiveSearch(world(k), cd, mps, margin=cd.radius, variant=v, **kw) for k, v, kw in cs]
host = make(combos)
ref = []
t0 = time.perf_counter()
for s in host:
    orig, count = s._a_star.neighbor_function, [0]
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
t_host = time.perf_counter() - t0
keep = [i for i, r in enumerate(ref) if not isinstance(r, str)]
dev = make([combos[i] for i in keep])
t0 = time.perf_counter()
results, info = plan_many_device(dev, debug=True, max_expansions=4096)
t_dev = time.perf_counter() - t0
bad = 0
for j, i in enumerate(keep):
    cost, path, traj = results[j]
    hc, hp, ht = ref[i]
    dd, hd = dev[j].debug_data, host[i].debug_data
    ok = (cost == hc and path == hp and [d.node for d in dd] == [d.node for d in hd] and [d.g for d in dd] == [d.g for d in hd]
          and [d.predecessor for d in dd] == [d.predecessor for d in hd] and np.array_equal(traj, ht))
    if not ok:
        bad += 1
        print('MISMATCH', combos[i])
per_variant = {v: sum(1 for i in keep if combos[i][1] == v) for v in variants}
print('%d searches (%d worlds x 2 variants): %d compared node for node %s, %d without solution on the host, %d beyond %d expansions; mismatches: %d'
      % (len(combos), len(combos) // 2, len(keep), per_variant, sum(r == 'none' for r in ref), sum(r == 'capacity' for r in ref), LIMIT, bad))
print('expansions up to %d (total %d); device path %.2f s in %d launches with %d overrides; host queues %.1f s'
      % (max(info['expansions']), sum(info['expansions']), t_dev, info['launches'], info['overrides'], t_host))
none_idx = [i for i, r in enumerate(ref) if r == 'none'][:6]
for i in none_idx:
    try:
        plan_many_device(make([combos[i]]), max_expansions=4096)
        print('MISMATCH: device found a solution where the host did not', combos[i]); bad += 1
    except Exception as e:
        assert 'No solution' in str(e) or 'exceeds' in str(e), e
print('"No solution found." reproduced on the device for %d searches' % len(none_idx))
sys.exit(1 if bad else 0)
