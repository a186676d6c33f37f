"""Per-phase shader-clock profile of the stage QP kernel (needs the MPCX_STAGE_PROFILE dev build installed as libmpcx.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import helpers as H
from mpc_for_av_at_intersection_amd.runtime import Context, MpcParams
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 32768
g = H.gold('mpc_pre.npz')
ctx = Context(0); ctx.set_mpc_params(MpcParams(T=T))
reps = (B + 59) // 60
tile = lambda a: np.concatenate([a] * reps)[:B]
st, xref, xbar, re = (ctx.f64(tile(g['T%d/state' % T])), ctx.f64(tile(g['T%d/xref' % T])), ctx.f64(tile(g['T%d/xbar' % T])), ctx.u8(tile(g['T%d/reaches_end' % T])))
f = torch.float64
big = torch.zeros((B + 4, 4), dtype=f, device=ctx.device)
out = dict(x=torch.empty((B, 4, T + 1), dtype=f, device=ctx.device), u=torch.empty((B, 2, T), dtype=f, device=ctx.device),
           status=torch.empty(B, dtype=torch.int32, device=ctx.device), iters=torch.empty(B, dtype=torch.int32, device=ctx.device), kkt=big)
ctx.qp_solve(st, xref, xbar, re, out=out); torch.cuda.synchronize()
big.zero_()
ctx.qp_solve(st, xref, xbar, re, out=out); torch.cuda.synchronize()
prof = big[B:].view(torch.int64).cpu().numpy().ravel()[:10].astype(np.float64)
names = ['refill/set-up', 'local A (rows, gradient)', 'costate sweep + norms', 'Riccati sweep', 'forward 1', 'local C (affine step)',
         'local D + corrector sweep', 'forward 2', 'local E + safeguard', 'update']
tot = prof.sum()
it = out['iters'].float().mean().item()
print('T=%d mean iters %.2f; shader-clock ticks summed over wavefronts: %.3g' % (T, it, tot))
for n, v in zip(names, prof):
    print('  %-28s %5.1f %%' % (n, 100 * v / tot))
print('relative to the Riccati sweep (unchanged code: a round counter):', ' '.join('%.3f' % (v / prof[3]) for v in prof), ' sum %.3f' % (tot / prof[3]))
