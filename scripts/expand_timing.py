"""Timing of the expansion kernel (2^20-node Prius frontier, SURVEY 8d config 5: free-space frontier and round 2's uniform one) and
consistency of sliced launches (dev aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_for_av_at_intersection_amd.batch import prius_frontier
from mpc_for_av_at_intersection_amd.runtime import Context
ctx = Context(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for free in (True, False):
    model, nodes = prius_frontier(ctx, 1 << 20, seed=0, free_space=free)
    out = ctx.expand(model, nodes)
    ok = True
    for lo in (0, 5000, 1 << 19):
        part = ctx.expand(model, nodes[lo:lo + 4000].contiguous())      # < 4096 nodes: the per-lane kernel
        for k in ('nbr', 'cost', 'collide'):
            ok &= bool(torch.equal(part[k], out[k][lo:lo + 4000]))
    print('%s frontier: free records %.3f; slices agree: %s; checksum %d' % ('free-space' if free else 'uniform', (out['collide'] == 0).double().mean().item(), ok,
                                                                             int(out['collide'].long().sum().item())))
    for n in (1 << 20, 1 << 17):
        sub = nodes[:n].contiguous(); o = ctx.expand(model, sub)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): ctx.expand(model, sub, out=o)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('   %8d nodes: %.3f ms = %.0f M nodes/s, %.0f GB/s algorithmic = %.1f %% of HBM peak' % (n, ms, n / ms / 1e3, n * 321 / ms / 1e6, n * 321 / ms / 1e6 / 80))
