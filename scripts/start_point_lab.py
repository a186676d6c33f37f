"""dev aid (CPU): does the interior-point iteration need fewer iterations on the HARD problems (moved path cut: the warm start is the
previous step's solution of a different problem) when it starts from the trial pass's unconstrained minimiser, clipped into the input
box, instead of the warm start?  numpy replica of the iteration (scripts/ipm_lab.py); corpus = tests/golden/qp_corpus.npz or a harvest
of scripts/warm_set_harvest.py.    python scripts/start_point_lab.py [corpus.npz] [N]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle_py as orc
from scripts import ipm_lab as L

T = 20


def run(path, N):
    p = orc.MpcParams(T=T)
    raw = np.load(path)
    if 's0/x0' in raw.files:
        d = {k: raw['s0/' + k] for k in ('x0', 'xref', 'xbar', 're', 'uw', 'iters')}
        moved = raw['s0/moved'] if 's0/moved' in raw.files else None
    else:
        d = {k: raw[k] for k in ('x0', 'xref', 'xbar', 're', 'uw', 'iters')}
        moved = None
    n = min(N, len(d['x0']))
    rng = np.random.default_rng(0)
    sel = rng.permutation(len(d['x0']))[:n]
    res = {k: [] for k in ('base', 'clip', 'raw', 'half', 'best2')}
    mv = []
    for k in sel:
        H, g, G, h, u0 = L.dense(p, d['x0'][k], d['xref'][k], d['xbar'][k], d['re'][k], d['uw'][k])
        w = u0 - np.linalg.solve(H, H @ u0 + g)
        if (G @ w - h <= 0).all():
            continue
        lo = np.empty(2 * T); hi = np.empty(2 * T)
        lo[0::2], hi[0::2] = p.max_decel, p.max_accel
        lo[1::2], hi[1::2] = -p.max_steer, p.max_steer
        wc = np.clip(w, lo, hi)
        pol = dict(when='exit', rho=1e8)

        def its(start):
            try:
                return L.ipm(H, g, G, h, start, polish=pol)[1]
            except np.linalg.LinAlgError:
                return 60
        res['base'].append(its(u0))
        res['clip'].append(its(wc))
        res['raw'].append(its(w))
        res['half'].append(its(0.5 * (wc + u0)))
        # the start with the smaller violation of the rows
        viol = lambda u: np.maximum(G @ u - h, 0).sum()
        res['best2'].append(its(wc if viol(wc) < viol(u0) else u0))
        mv.append(bool(moved[k]) if moved is not None else False)
    mv = np.array(mv)
    for name, v in res.items():
        v = np.array(v)
        line = '%-6s n %4d  mean %.2f  p90 %d  p99 %d  max %d' % (name, len(v), v.mean(), np.quantile(v, .9), np.quantile(v, .99), v.max())
        if moved is not None and mv.any():
            line += '   | moved: mean %.2f max %d   | not moved: mean %.2f max %d' % (v[mv].mean(), v[mv].max(), v[~mv].mean(), v[~mv].max())
        print(line)
    b, c = np.array(res['base']), np.array(res['clip'])
    print('clip better %d, equal %d, worse %d; on base >= 10: base mean %.2f clip mean %.2f' % ((c < b).sum(), (c == b).sum(), (c > b).sum(), b[b >= 10].mean() if (b >= 10).any() else 0, c[b >= 10].mean() if (b >= 10).any() else 0))


if __name__ == '__main__':
    run(sys.argv[1] if len(sys.argv) > 1 else 'tests/golden/qp_corpus.npz', int(sys.argv[2]) if len(sys.argv) > 2 else 400)
