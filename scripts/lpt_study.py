"""Would ordering the QP work queue by the previous step's iteration counts (longest first) shorten the launch? (dev study)"""
import sys, os, heapq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_for_av_at_intersection_amd.batch import synthetic_batch
from mpc_for_av_at_intersection_amd.runtime import Context
ctx = Context(0)
sim = synthetic_batch(ctx, B=4096, A=8, T=20, seed=1000)
its = []
for k in range(25):
    sim.step(); its.append(sim.sol['iters'].cpu().numpy().copy())
def makespan(order, dur, ngroups=8192):
    h = [0] * ngroups; heapq.heapify(h)
    end = 0
    for i in order:
        t = heapq.heappop(h) + dur[i] + 1          # +1 round for hand-in / set-up
        end = max(end, t); heapq.heappush(h, t)
    return end
for k in (6, 12, 18, 24):
    prev, cur = its[k - 1], its[k]
    n = len(cur)
    fifo = makespan(np.arange(n), cur)
    lpt_prev = makespan(np.argsort(-prev, kind='stable'), cur)
    lpt_true = makespan(np.argsort(-cur, kind='stable'), cur)
    ideal = (cur + 1).sum() / 8192
    two = {th: makespan(np.concatenate([np.nonzero(prev >= th)[0], np.nonzero(prev < th)[0]]), cur) for th in (7, 8, 10)}
    ema = 0.5 * its[k - 1] + 0.3 * its[k - 2] + 0.2 * its[k - 3]
    lpt_ema = makespan(np.argsort(-ema, kind='stable'), cur)
    mx = np.maximum(its[k - 1], its[k - 2])
    lpt_mx = makespan(np.argsort(-mx, kind='stable'), cur)
    print('   two-class thresholds', two, 'EMA', lpt_ema, 'max of last two', lpt_mx)
    print('step %d: corr(prev, cur) = %.2f; rounds: ideal %.1f, FIFO %d, LPT by previous iters %d, LPT by true iters %d' % (
        k, np.corrcoef(prev, cur)[0, 1], ideal, fifo, lpt_prev, lpt_true), flush=True)
