"""dev aid (GPU box): where plan_many_device spends its host time for 1024 searches"""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_for_av_at_intersection_amd.runtime import Context
from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions
from mpc_for_av_at_intersection_amd.lib.motion_primitive import load_motion_primitives
from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import MotionPrimitiveSearch, plan_many_device
from mpc_for_av_at_intersection_amd.lib.scenario import intersection
from mpc_for_av_at_intersection_amd.lib import _session
ctx = _session.context()
cd, mps = BicycleModelDimensions(), load_motion_primitives('bicycle_model')
pairs = [(sp, ti) for sp in (1, 2, 3, 4) for ti in (1, 2, 3)]
mk = lambda n: [MotionPrimitiveSearch(intersection(turn_indicator=pairs[i % 12][1], start_pos=pairs[i % 12][0]), cd, mps, margin=cd.radius, variant='modified', ctx=ctx) for i in range(n)]
plan_many_device(mk(12))
ss = mk(1024)
torch.cuda.synchronize(); t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
res, inf = plan_many_device(ss)
pr.disable()
torch.cuda.synchronize(); print('total %.1f ms; closure %.1f device %.1f check %.1f' % (1e3 * (time.perf_counter() - t0), 1e3 * inf['t_closure'], 1e3 * inf['t_device'], 1e3 * inf['t_check']))
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
