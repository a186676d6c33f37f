"""GPU box: the benchmark's closed loop (4096 instances x 8 agents, T = 20) for 200 steps, EVERY agent of every 10th step replayed on the
oracle from the device state of the step before (tests/test_gpu_fullsize.py::_replay_all_on_oracle: path indices, cut lengths, conflict
indices, statuses identical for every agent; solutions within 2e-7).  Summary for profiles/."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_for_av_at_intersection_amd.batch import synthetic_batch
from mpc_for_av_at_intersection_amd.runtime import Context
from tests.test_gpu_fullsize import _replay_all_on_oracle
ctx = Context(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sim = synthetic_batch(ctx, B=B, A=8, T=20, seed=1000)
worst, n_rep, it_diff_all, cut, standing, free, t0 = 0.0, 0, 0, 0, 0, 0, time.perf_counter()
dead = np.zeros(sim.P, bool)
for k in range(20):
    sim.run(9)
    before = sim.snapshot()
    sim.step()
    after = sim.snapshot()
    w, it_diff, failed = _replay_all_on_oracle(sim, before, after, threads=16, dead=dead)
    dead |= sim.raised
    assert failed == 0
    worst = max(worst, w); n_rep += sim.P; it_diff_all += it_diff
    cut += int((after['cut_len'] < sim.path_len.cpu().numpy()).sum()); standing += int((after['traj_idx'] == before['traj_idx']).sum()); free += int((after['hit_idx'] == -1).sum())
    print('step %3d: worst so far %.2e, iteration counts differing %d, mean speed %.2f' % (sim.steps_done, worst, it_diff, float(after['state'][:, 2].mean())), flush=True)
sim.check()
print('%d agent-steps replayed on the oracle over %d closed-loop steps of %d x 8 agents: every integer decision and status identical, worst |GPU - oracle| %.2e, '
      '%d iteration counts differ; agent-steps with a cut path %d, standing %d, conflict-free %d; %.0f s'
      % (n_rep, sim.steps_done, B, worst, it_diff_all, cut, standing, free, time.perf_counter() - t0))
