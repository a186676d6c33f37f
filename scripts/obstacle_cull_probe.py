"""dev aid (GPU box): how many (ego, obstacle) pairs of the benchmark could an obstacle-level bounding-box cull drop in front of the conflict
search?  ego box = remaining path points, obstacle box = its 35 predicted poses, both inflated by the car's reach (2 m disc offset + 2 radii)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_for_av_at_intersection_amd.runtime import Context
from mpc_for_av_at_intersection_amd.batch import synthetic_batch
ctx = Context(0)
sim = synthetic_batch(ctx, B=512, A=8, T=20, seed=1000)
for burn in (12, 40):
    sim.run(burn); ctx.synchronize()
    path = sim.path.cpu().numpy(); off = sim.path_off.cpu().numpy(); ln = sim.path_len.cpu().numpy(); ti = sim.traj_idx.cpu().numpy()
    st = sim.state.cpu().numpy()
    pred = ctx.predict_obstacles(sim.obs6, sim.ip.pred_steps, sim.ip.dt, sim.ip.L).cpu().numpy()      # (P, 35, 3)
    P = len(st); A = 8
    infl = 2.2 + 2 * sim.ip.radius
    ob = np.stack([pred[:, :, 0].min(1), pred[:, :, 0].max(1), pred[:, :, 1].min(1), pred[:, :, 1].max(1)], 1)
    keep = tot = 0
    for p in range(P):
        pts = path[off[p] + ti[p]:off[p] + ln[p], :2]
        e = (pts[:, 0].min() - infl, pts[:, 0].max() + infl, pts[:, 1].min() - infl, pts[:, 1].max() + infl)
        inst = p // A
        for q in range(inst * A, inst * A + A):
            if q == p: continue
            tot += 1
            keep += not (ob[q, 0] > e[1] or ob[q, 1] < e[0] or ob[q, 2] > e[3] or ob[q, 3] < e[2])
    print('after %d more steps: %d of %d (ego, obstacle) pairs survive an obstacle-level box test (%.1f %%)' % (burn, keep, tot, 100.0 * keep / tot))
