"""Per-step latency of the closed loop at small batch sizes: staged host calls vs mpcx_closed_loop_run vs graph replay."""
import sys, time
sys.path.insert(0, '.')
import torch
from mpc_for_av_at_intersection_amd.batch import stock_routes, synthetic_batch
from mpc_for_av_at_intersection_amd.runtime import Context

ctx = Context(0)
side = Context(0, stream=torch.cuda.Stream(device=0))
routes, dl, cd = stock_routes(ctx)
for B in (1, 8, 64, 512, 4096):
    row = []
    for mode in ('staged', 'fused', 'graph'):
        c = side if mode == 'graph' else ctx
        sim = synthetic_batch(c, B=B, A=8, T=20, seed=5, routes=routes, dl=dl, cd=cd)
        n = 40
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode == 'staged':
                for _ in range(n):
                    sim.step_staged()
            else:
                sim.run(n, graph=(mode == 'graph'))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        row.append('%s %.3f ms' % (mode, dt * 1e3))
    print('B=%5d x 8 agents, T=20: ' % B + ' | '.join(row), flush=True)
