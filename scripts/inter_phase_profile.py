"""Per-phase shader-clock profile of interaction_kernel (needs the MPCX_INTER_PROFILE dev build installed as libmpcx.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_for_av_at_intersection_amd.batch import synthetic_batch
from mpc_for_av_at_intersection_amd.runtime import Context
ctx = Context(0)
sim = synthetic_batch(ctx, B=4096, A=8, T=20, seed=1000)
P = sim.P
big = torch.zeros((P + 8, 2), dtype=torch.float64, device=ctx.device)
sim.inter['hit_xy'] = big
sim._desc = None
for _ in range(12): sim.step_staged()
torch.cuda.synchronize(); big[P:].zero_()
for _ in range(5): sim.step_staged()
torch.cuda.synchronize()
prof = big[P:].view(torch.int64).cpu().numpy().ravel()[:7].astype(np.float64)
names = ['distance / step-length pass', 'three-smallest selection', 'sequential cumsum', 'resample', 'ego discs', 'conflict search', 'cut index']
for n, v in zip(names, prof): print('  %-30s %5.1f %%' % (n, 100 * v / prof.sum()))
