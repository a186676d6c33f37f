"""Per-phase shader-clock profile of interaction_kernel (needs the MPCX_INTER_PROFILE dev build installed as libmpcx.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_for_av_at_intersection_amd.batch import synthetic_batch
from mpc_for_av_at_intersection_amd.runtime import Context
ctx = Context(0)
sim = synthetic_batch(ctx, B=4096, A=8, T=20, seed=1000)
P = sim.P
big = torch.zeros((P + 8 * P, 2), dtype=torch.float64, device=ctx.device)   # 16 slots per ego behind hit_xy
sim.inter['hit_xy'] = big
sim._desc = None
for _ in range(12): sim.step_staged()
torch.cuda.synchronize(); big[P:].zero_()
for _ in range(1): sim.step_staged()
torch.cuda.synchronize()
allp = big[P:].view(torch.int64).cpu().numpy().reshape(P, 16).astype(np.float64)
per = allp[:, :7]
prof = per.sum(axis=0)
names = ['distance / step-length pass', 'three-smallest selection', 'sequential cumsum', 'resample', 'ego discs', 'conflict search', 'cut index']
for n, v in zip(names, prof): print('  %-30s %5.1f %%  %8.0f cycles per ego' % (n, 100 * v / prof.sum(), v / P))
print('  total %.0f cycles per ego' % (prof.sum() / P))
for k, nme in zip(range(8, 13), ['run boxes', 'candidate loads', 'box tests + compaction (all runs visited)', 'queue work (all runs visited)', 'earliest pose + winner']):
    print('    conflict search / %-45s %8.0f cycles per ego' % (nme, allp[:, k].sum() / P))

print('  per-ego total: median %.0f  p90 %.0f  max %.0f cycles' % (np.median(per.sum(1)), np.quantile(per.sum(1), 0.9), per.sum(1).max()))

hi = sim.inter['hit_idx'][:P].cpu().numpy(); hq = big[:P, 0].cpu().numpy()
free = hi == -1
print('  egos without a conflict: %.1f %%; of those, no candidate in any run box: %.1f %%; candidates queued (mean over conflict-free egos): %.1f'
      % (100 * free.mean(), 100 * (hq[free] == 0).mean(), hq[free].mean()))
