"""dev aid (GPU box): which cheap features of a QP's inputs predict that it will be constrained (trial pass rejected) or hard
(>= 10 iterations), beyond the previous step's iteration count the work queue is ordered by today"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_for_av_at_intersection_amd.runtime import Context
from mpc_for_av_at_intersection_amd.batch import synthetic_batch

ctx = Context(0)
sim = synthetic_batch(ctx, B=4096, A=8, T=20, seed=1000)
sim.run(5)
prev = sim.snapshot()
p = sim.params
for step in range(5):
    sim.step()
    cur = sim.snapshot()
    it, pit = cur['iters'], prev['iters']
    uw = prev['u']                      # the warm start of this step = last step's solution (unshifted, as the kernels use it)
    xbar, xref, re = cur['xbar'], cur['xref'], cur['reaches_end']
    moved = cur['cut_len'] != prev['cut_len']
    f_abound = (uw[:, 0].max(1) >= 0.98 * p.max_accel) | (uw[:, 0].min(1) <= 0.98 * p.max_decel)
    f_sbound = np.abs(uw[:, 1]).max(1) >= 0.98 * p.max_steer
    f_rate = np.abs(np.diff(uw[:, 1], axis=1)).max(1) >= 0.98 * p.max_dsteer * p.dt
    f_v = (xbar[:, 2].max(1) >= 0.97 * p.max_speed) | (xbar[:, 2].min(1) <= 0.2)
    f_end = re.any(1)
    f_err = np.abs(xbar[:, :2, 1:] - xref[:, :2, 1:]).max((1, 2)) > 1.0
    con, hard = it > 0, it >= 10
    print('step %d: constrained %d (prev unconstrained: %d), hard %d (prev unconstrained: %d)' % (step, con.sum(), (con & (pit == 0)).sum(), hard.sum(), (hard & (pit == 0)).sum()))
    base = pit == 0
    for name, f in (('moved', moved), ('accel at bound', f_abound), ('steer at bound', f_sbound), ('steer rate at bound', f_rate),
                    ('v near a speed bound', f_v), ('window reaches path end', f_end), ('rollout > 1 m off the window', f_err),
                    ('any of them', moved | f_abound | f_sbound | f_rate | f_v | f_end | f_err)):
        sel = base & f
        print('   prev==0 & %-28s: %5d problems, constrained now %5.1f%%, hard now %5.1f%%  | covers %4.1f%% of the unforeseen constrained, %4.1f%% of the unforeseen hard'
              % (name, sel.sum(), 100 * con[sel].mean() if sel.any() else 0, 100 * hard[sel].mean() if sel.any() else 0,
                 100 * (sel & con).sum() / max(1, (base & con).sum()), 100 * (sel & hard).sum() / max(1, (base & hard).sum())))
    prev = cur
