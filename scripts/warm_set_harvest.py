"""GPU box: QPs of the benchmark's closed loop WITH their work-queue key (previous step's iteration count, "the path cut moved"),
for scripts/warm_set_lab.py (acceptance of an active-set solve started from the set the warm start sits on,
by queue key).   python scripts/warm_set_harvest.py [out.npz] [per_step]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mpc_for_av_at_intersection_amd.batch import synthetic_batch
from mpc_for_av_at_intersection_amd.runtime import Context

out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/warm_set_corpus.npz'
per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctx = Context(0)
sim = synthetic_batch(ctx, B=4096, A=8, T=20, seed=1000)
rng = np.random.default_rng(0)
rows = []
sim.run(5)
for step in (6, 7, 8, 12, 20, 40):
    if step - 1 > sim.steps_done:
        sim.run(step - 1 - sim.steps_done)
    uw = sim.sol['u'].clone(); prev_it = sim.sol['iters'].clone(); prev_cut = sim.inter['cut_len'].clone()
    sim.step()
    it = sim.sol['iters']
    con = torch.nonzero(it > 0).flatten().cpu().numpy()
    idx = torch.as_tensor(np.sort(rng.choice(con, min(per_step, len(con)), replace=False)), device=it.device)
    rows.append(dict(x0=sim.sol['x'][idx][:, :, 0].cpu().numpy(), xref=sim.pre['xref'][idx].cpu().numpy(), xbar=sim.pre['xbar'][idx].cpu().numpy(),
                     re=sim.pre['reaches_end'][idx].cpu().numpy(), uw=uw[idx].cpu().numpy(), iters=it[idx].cpu().numpy(), u=sim.sol['u'][idx].cpu().numpy(),
                     prev_iters=prev_it[idx].cpu().numpy(), moved=(sim.inter['cut_len'][idx] != prev_cut[idx]).cpu().numpy(),
                     step=np.full(len(idx), sim.steps_done, np.int32)))
    print('step %d: %d constrained of %d; sample %d' % (sim.steps_done, len(con), len(it), len(idx)), flush=True)
np.savez_compressed(out, **{'s0/' + k: np.concatenate([r[k] for r in rows]) for k in rows[0]})
