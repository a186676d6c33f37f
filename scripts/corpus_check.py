"""GPU box: both HIP solvers on a saved QP corpus (scripts/harvest_qp.py) against the oracle: iteration counts and distances."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_for_av_at_intersection_amd.runtime import Context, MpcParams
from oracle import oracle_py as orc

path = sys.argv[1] if len(sys.argv) > 1 else 'build/qp_corpus.npz'
d = np.load(path); d = {k[3:]: d[k] for k in d.files}
N = int(sys.argv[2]) if len(sys.argv) > 2 else len(d['iters'])
T = d['xref'].shape[2] - 1
ctx = Context(0)
ctx.set_mpc_params(MpcParams(T=T))
po = orc.MpcParams(T=T)
sols = [orc.qp_solve(po, d['x0'][k], d['xref'][k], d['xbar'][k], d['re'][k], d['uw'][k]) for k in range(N)]
oit = np.array([s.iters for s in sols]); ou = np.stack([s.u for s in sols])
for name in ('stage', 'condensed'):
    ctx.set_qp_solver(name)
    o = ctx.qp_solve(ctx.f64(d['x0'][:N]), ctx.f64(d['xref'][:N]), ctx.f64(d['xbar'][:N]), ctx.u8(d['re'][:N]), ctx.f64(d['uw'][:N]))
    ctx.synchronize()
    u = o['u'].cpu().numpy(); it = o['iters'].cpu().numpy(); st = o['status'].cpu().numpy()
    du = np.abs(u - ou).max((1, 2))
    print('%s: status!=0 %d | iters == oracle %.4f, |d|<=1 %.4f | max du %.2e p99 %.2e | mean its %.2f max %d' % (
        name, (st != 0).sum(), (it == oit).mean(), (np.abs(it - oit) <= 1).mean(), du.max(), np.quantile(du, .99), it[it > 0].mean(), it.max()))
    w = np.argsort(-du)[:8]
    print('   worst', [(int(k), int(it[k]), int(oit[k]), '%.1e' % du[k]) for k in w])
    np.savez('gpurun_out/cc_%s.npz' % name, u=u, x=o['x'].cpu().numpy(), it=it, kkt=o['kkt'].cpu().numpy(), ou=ou, oit=oit)
