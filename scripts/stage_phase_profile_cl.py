"""Per-phase shader-clock profile of the stage QP kernel on the BENCHMARK workload (closed loop, 4096 x 8, T = 20); needs the
MPCX_STAGE_PROFILE dev build via MPCX_LIB.  Ticks are summed over wavefronts and over the profiled steps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_for_av_at_intersection_amd.runtime import Context
from mpc_for_av_at_intersection_amd.batch import synthetic_batch
ctx = Context(0)
sim = synthetic_batch(ctx, B=4096, A=8, T=20, seed=1000)
P = sim.P
big = torch.zeros((P + 4, 4), dtype=torch.float64, device=ctx.device)
sim.sol['kkt'] = big          # the dev build adds its ten counters behind the kkt rows
sim.run(8)
torch.cuda.synchronize()
big[P:].zero_()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sim.run(n)
torch.cuda.synchronize()
prof = big[P:].view(torch.int64).cpu().numpy().ravel()[:10].astype(np.float64)
names = ['refill/set-up', 'local A (rows, gradient)', 'costate sweep + norms', 'Riccati sweep', 'forward 1', 'local C (affine step)',
         'local D + corrector sweep', 'forward 2', 'local E + safeguard', 'update']
tot = prof.sum()
print('%d steps; shader-clock ticks summed over wavefronts: %.4g (%.3g per step)' % (n, tot, tot / n))
for nm, v in zip(names, prof):
    print('  %-28s %5.1f %%   %.3f of the Riccati sweep' % (nm, 100 * v / tot, v / prof[3]))
print('sum relative to the Riccati sweep %.3f' % (tot / prof[3]))
