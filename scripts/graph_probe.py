"""dev aid (GPU): ms per step of the bench workload, mpcx_closed_loop_run enqueued kernel by kernel against the replayed hipGraph"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpc_for_av_at_intersection_amd.batch import stock_routes, synthetic_batch
from mpc_for_av_at_intersection_amd.runtime import Context
s = torch.cuda.Stream()
ctx = Context(0, stream=s) if 'stream' in Context.__init__.__code__.co_varnames else Context(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for graph in (False, True, False, True):
    sim = synthetic_batch(ctx, B=B, A=8, T=20, seed=1000)
    try:
        sim.run(8, graph=graph)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sim.run(20, graph=graph)
        torch.cuda.synchronize(); t = time.perf_counter() - t0
        print('graph=%s: %.4f ms per step' % (graph, 1e3 * t / 20), flush=True)
    except Exception as e:
        print('graph=%s failed: %r' % (graph, e))
