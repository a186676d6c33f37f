"""Solve the QPs saved by capture_qp_failures.py with both HIP solvers and the oracle (dev aid). usage: solve_saved_qp.py file.npz"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mpc_for_av_at_intersection_amd.runtime import Context, MpcParams
from oracle import oracle_py as orc
d = np.load(sys.argv[1]); T = 20
ctx = Context(0); ctx.set_mpc_params(MpcParams(T=T))
n = len(d['x0'])
for reps in (1, 8, 64):
    tile = lambda a: np.concatenate([a] * reps)
    for solver in ('stage', 'condensed'):
        ctx.set_qp_solver(solver)
        out = ctx.qp_solve(ctx.f64(tile(d['x0'])), ctx.f64(tile(d['xref'])), ctx.f64(tile(d['xbar'])), ctx.u8(tile(d['re'])), ctx.f64(tile(d['uw'])))
        ctx.synchronize()
        print('copies %2d %-9s status %s iters %s kkt0 %s' % (reps, solver, out['status'].cpu().numpy()[:n], out['iters'].cpu().numpy()[:n], out['kkt'].cpu().numpy()[:n, 0]))
for i in range(n):
    r = orc.qp_solve(orc.MpcParams(T=T), d['x0'][i], d['xref'][i], d['xbar'][i], d['re'][i], d['uw'][i])
    print('oracle', r.status, r.iters, r.kkt[0])
