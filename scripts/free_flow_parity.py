"""GPU box: SURVEY 8(d) config 2's generator at 32 768 independent single-ego instances (bench.py's free_flow workload: perturbed poses, speeds
over the whole range, 80 % of the QPs constrained), 60 steps, EVERY agent of every 10th step replayed on the oracle.  Summary for profiles/."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_for_av_at_intersection_amd.batch import config2_batch
from mpc_for_av_at_intersection_amd.runtime import Context
from tests.test_gpu_fullsize import _replay_all_on_oracle
ctx = Context(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
sim = config2_batch(ctx, B=B, T=20, seed=0)
worst, n_rep, itd, con, t0 = 0.0, 0, 0, 0, time.perf_counter()
dead = np.zeros(sim.P, bool)
for k in range(6):
    sim.run(9)
    before = sim.snapshot()
    sim.step()
    after = sim.snapshot()
    w, it_diff, failed = _replay_all_on_oracle(sim, before, after, threads=16, dead=dead)
    dead |= sim.raised
    worst = max(worst, w); n_rep += sim.P; itd += it_diff; con += int((after['iters'] > 0).sum())
    print('step %3d: worst so far %.2e, iteration counts differing %d, failed %d, raised so far %d, mean speed %.2f' % (sim.steps_done, worst, it_diff, failed, int(dead.sum()), float(after['state'][:, 2].mean())), flush=True)
print('%d agent-steps replayed (%d constrained QPs): every integer decision and status identical, worst |GPU - oracle| %.2e, %d iteration counts differ; %.0f s'
      % (n_rep, con, worst, itd, time.perf_counter() - t0))
