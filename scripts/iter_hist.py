"""dev aid (GPU box): per-step histogram of interior-point iteration counts of the benchmark workload and how well the previous
step's count predicts the current one (the work queue of the stage solver is ordered by that prediction)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mpc_for_av_at_intersection_amd.runtime import Context
from mpc_for_av_at_intersection_amd.batch import synthetic_batch

ctx = Context(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sim = synthetic_batch(ctx, B=B, A=8, T=20, seed=0)
prev = None
for step in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    sim.step()
    it = sim.snapshot()['iters']
    h = np.bincount(it, minlength=1)
    line = 'step %2d mean %.2f zero %.1f%% max %d  hist %s' % (step, it.mean(), 100 * (it == 0).mean(), it.max(), h[:24].tolist())
    if prev is not None:
        hard = it >= 10
        line += ' | hard now: %d, of which prev==0: %d, prev>=8: %d' % (hard.sum(), (hard & (prev == 0)).sum(), (hard & (prev >= 8)).sum())
    print(line, flush=True)
    prev = it
