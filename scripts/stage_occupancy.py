"""dev aid (GPU box, MPCX_LIB = the MPCX_STAGE_PROFILE build): how many lane groups of the stage QP kernel are at work in every round of a benchmark
launch -- per wavefront (8 groups each) and summed over the 1024 wavefronts -- i.e. where the launch loses its time: work, imbalance, tail."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_for_av_at_intersection_amd.runtime import Context
from mpc_for_av_at_intersection_amd.batch import synthetic_batch
ctx = Context(0)
sim = synthetic_batch(ctx, B=4096, A=8, T=20, seed=1000)
P = sim.P
G = 1024
big = torch.zeros((P + 4 + 4 * G + 8 + P // 16 + 8, 4), dtype=torch.float64, device=ctx.device)
sim.sol['kkt'] = big
sim.run(int(sys.argv[1]) if len(sys.argv) > 1 else 10)
for step in range(4):
    big[P:].zero_()
    prev_it = sim.sol['iters'].cpu().numpy().copy(); prev_cut = sim.inter['cut_len'].cpu().numpy().copy()
    sim.run(1)
    torch.cuda.synchronize()
    raw = big[P:].view(torch.uint8).cpu().numpy().ravel()[16 * 8:16 * 8 + 128 * G].reshape(G, 64, 2)
    occ, spec = raw[:, :, 0].astype(int), raw[:, :, 1].astype(int)
    alive = (occ > 0)
    last = np.where(alive.any(1), 64 - np.argmax(alive[:, ::-1], axis=1), 0)          # rounds each wavefront ran
    per_round = occ.sum(0)
    print('step %d: rounds per wavefront mean %.1f max %d; group-rounds %d (of which trial / polish %d) = %.1f per group; its max %d'
          % (step, last.mean(), last.max(), occ.sum(), spec.sum(), occ.sum() / (8.0 * G), int(sim.sol['iters'].max().item())))
    print('   groups at work per round (of %d): %s' % (8 * G, ' '.join('%d' % v for v in per_round[:last.max()])))
    print('   wavefronts alive per round:        %s' % ' '.join('%d' % v for v in alive.sum(0)[:last.max()]))
    life = big[P:].view(torch.uint8).cpu().numpy().ravel()[(16 + 16 * G) * 8:(16 + 16 * G) * 8 + 2 * P].reshape(P, 2).astype(int)
    its = sim.sol['iters'].cpu().numpy()
    start, end = life[:, 0], life[:, 1] % 100
    late = end >= 12
    print('   problems handed in at round >= 12: %d; their start rounds: %s; their iteration counts: %s; handed over: %d'
          % (late.sum(), np.bincount(start[late], minlength=12)[:14].tolist(), np.bincount(its[late], minlength=16)[:18].tolist(), int((life[:, 1] >= 100).sum())))
    c = its > 0
    print('   constrained problems: %d; start rounds %s' % (c.sum(), np.bincount(start[c], minlength=12)[:14].tolist()))
    key = np.minimum(63, np.maximum(prev_it, 0) + 11 * (sim.inter['cut_len'].cpu().numpy() != prev_cut))
    for lo, hi in ((0, 0), (1, 5), (6, 10), (11, 11), (12, 16), (17, 63)):
        m = (key >= lo) & (key <= hi)
        if m.any():
            print('   key %2d..%2d: %5d problems, constrained %5d; start rounds %s' % (lo, hi, m.sum(), (m & c).sum(), np.bincount(start[m], minlength=12)[:13].tolist()))
    light = alive & (spec == occ)
    lastw = np.argmax(last)      # a wavefront that ran the longest
    print('   wavefront-rounds in which every working group is in a trial / polish round: %.1f %% of the live ones; per round: %s; in the longest-running wavefront: %d of %d'
          % (100.0 * light.sum() / alive.sum(), ' '.join('%d' % v for v in light.sum(0)[:last.max()]), int(light[lastw].sum()), int(last[lastw])))
    busy = occ[alive]
    print('   mean groups at work in a live wavefront-round: %.2f of 8' % busy.mean())
