"""Wall time of the 12 stock route searches through lib.MotionPrimitiveSearch (host queue + GPU expansion)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_for_av_at_intersection_amd.runtime import Context
from mpc_for_av_at_intersection_amd.lib import _session
from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions
from mpc_for_av_at_intersection_amd.lib.motion_primitive import load_motion_primitives
from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import MotionPrimitiveSearch
from mpc_for_av_at_intersection_amd.lib.scenario import intersection
ctx = Context(0); _session.set_context(ctx)
cd = BicycleModelDimensions(); mps = load_motion_primitives('bicycle_model')
for variant in ('modified', 'base'):
    tot = 0.0
    for sp in (1, 2, 3, 4):
        for ti in (1, 2, 3):
            s = MotionPrimitiveSearch(intersection(ti, sp), cd, mps, margin=cd.radius, variant=variant)
            t0 = time.perf_counter(); c, p, tr = s.run(debug=True); dt = time.perf_counter() - t0
            tot += dt
            print('%s sp%d ti%d: %.1f ms, %d expansions, %d launches' % (variant, sp, ti, dt * 1e3, len(s.debug_data), s.kernel_launches), flush=True)
    print(variant, 'total %.3f s' % tot)
