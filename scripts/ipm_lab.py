"""dev aid (CPU): numpy replica of the interior-point iteration of oracle.c:orc_ipm_dense on the condensed problems of a harvested
sample (gpurun_out/qp_sample.npz or tests/golden/qp_corpus.npz), used to try changes of the iteration (exit rules, active-set polish) before they go into
oracle.c and the two HIP solvers.  `exact()` is the active-set solution of the same condensed problem (numpy.linalg.solve)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle_py as orc

FRACTION, FLOOR, LAM0 = 0.999, 0.5, 3.0


def load(path='gpurun_out/qp_sample.npz'):
    d = np.load(path)
    keys = sorted(set(k.split('/')[0] for k in d.files))
    cat = lambda n: np.concatenate([d['%s/%s' % (k, n)] for k in keys])
    return dict(x0=cat('x0'), xref=cat('xref'), xbar=cat('xbar'), re=cat('re'), uw=cat('uw'), iters=cat('iters'), u=cat('u'))


def dense(p, x0, xref, xbar, re, uw):
    H, g, G, h, S, c = orc.qp_build(p, x0, xref, xbar, re)
    u0 = np.empty(2 * p.T); u0[0::2] = uw[0]; u0[1::2] = uw[1]
    return H, g, G, h, u0


def exact(H, g, G, h, u_start, tol_act=1e-6):
    act = set(np.where(h - G @ u_start < tol_act)[0].tolist())
    n = len(g)
    for rounds in range(100):
        a = sorted(act)
        K = np.zeros((n + len(a), n + len(a)))
        K[:n, :n] = H; K[:n, n:] = G[a].T; K[n:, :n] = G[a]
        try:
            s = np.linalg.solve(K, np.concatenate([-g, h[a]]))
        except np.linalg.LinAlgError:
            s = np.linalg.lstsq(K, np.concatenate([-g, h[a]]), rcond=None)[0]
        u, lam = s[:n], s[n:]
        slack = h - G @ u
        drop = [(lam[i], a[i]) for i in range(len(a)) if lam[i] < -1e-9]
        add = [i for i in range(len(h)) if i not in act and slack[i] < -1e-11]
        if not drop and not add:
            return u, a, lam
        if drop:
            act.discard(min(drop)[1])
        act.update(add)
    raise RuntimeError('active set did not settle')


def ipm(H, g, G, h, u0, tol=1e-10, max_iter=60, polish=None, trace=None):
    """returns (u, iterations, info).  polish = dict(when=<mu threshold or 'exit'>, rho=...) tries the active-set polish."""
    n, m = len(g), len(h)
    u = u0.copy()
    si = h - G @ u
    s = np.maximum(si, FLOOR); lam = np.full(m, LAM0)
    gnorm = max(1.0, np.abs(g).max()); hnorm = max(1.0, np.abs(h).max())
    info = dict(polished=False, polish_tries=0)
    # trial
    w = u - np.linalg.solve(H, H @ u + g)
    if (G @ w - h <= 0).all():
        return w, 0, info
    tol_loose = max(tol, 1e-7)
    loose_run = 0
    for it in range(max_iter + 1):
        rd = g + H @ u + G.T @ lam
        rp = s - h + G @ u
        mu = (s * lam).sum() / m
        res_d, res_p = np.abs(rd).max(), np.abs(rp).max()
        if trace is not None:
            trace.append((it, res_d / gnorm, res_p / hnorm, mu))
        conv = res_d <= tol * gnorm and res_p <= tol * hnorm and mu <= tol
        if polish is not None and it >= 1:
            go = conv if polish['when'] == 'exit' else (mu <= polish['when'] and res_p <= polish.get('rp', 1e-6) * hnorm)
            if go or conv:
                info['polish_tries'] += 1
                r = try_polish(H, g, G, h, u, s, lam, polish)
                if r is not None:
                    info['polished'] = True
                    return r, it, info
        if conv:
            return u, it, info
        loose = res_d <= tol_loose * gnorm and res_p <= tol_loose * hnorm and mu <= tol_loose
        loose_run = loose_run + 1 if loose else 0
        if loose_run >= 4:
            return u, it, info
        if it == max_iter:
            return u, it, info
        d = lam / s
        M = H + G.T @ (d[:, None] * G)
        L = np.linalg.cholesky(M)
        solve = lambda b: np.linalg.solve(L.T, np.linalg.solve(L, b))
        wv = -lam + d * rp
        du = solve(-rd - G.T @ wv)
        dsa = -rp - G @ du
        dla = -lam - d * dsa
        alpha = 1.0
        neg = dsa < 0
        if neg.any(): alpha = min(alpha, (-s[neg] / dsa[neg]).min())
        neg = dla < 0
        if neg.any(): alpha = min(alpha, (-lam[neg] / dla[neg]).min())
        mu_aff = ((s + alpha * dsa) * (lam + alpha * dla)).sum() / m
        sigma = (mu_aff / mu) ** 3
        smu = max(sigma * mu, 0.1 * tol)
        rc = s * lam + alpha * (dsa * dla) - smu
        wv = (-rc + lam * rp) / s
        du = solve(-rd - G.T @ wv)
        ds = -rp - G @ du
        dl = -(rc + lam * ds) / s
        ap = ad = 1e300
        neg = ds < 0
        if neg.any(): ap = (-s[neg] / ds[neg]).min()
        neg = dl < 0
        if neg.any(): ad = (-lam[neg] / dl[neg]).min()
        ap = min(1.0, FRACTION * ap); ad = min(1.0, FRACTION * ad)
        for tr in range(6):
            pr = (s + ap * ds) * (lam + ad * dl)
            if pr.min() >= 1e-3 * pr.sum() / m:
                break
            ap *= 0.7; ad *= 0.7
        u = u + ap * du; s = s + ap * ds; lam = lam + ad * dl
    return u, max_iter, info


def try_polish(H, g, G, h, u, s, lam, opt):
    """one augmented-Lagrangian solve on the rows with s < lam (penalty rho, multiplier estimate = the iterate's lam); accepted only if
    the new multipliers are >= 0 and no row is violated (then the point is the optimum up to ~ |lam - lam*| / rho)"""
    rho = opt.get('rho', 1e6)
    act = s < lam
    gap = G @ u - h
    nu = np.where(act, lam + rho * gap, 0.0)
    M = H + rho * (G[act].T @ G[act])
    du = np.linalg.solve(M, -(g + H @ u + G.T @ nu))
    un = u + du
    gapn = G @ un - h
    lamn = np.where(act, lam + rho * gapn, 0.0)
    eps_l = opt.get('eps_l', 1e-9) * max(1.0, np.abs(lam).max())
    eps_g = opt.get('eps_g', 1e-9)
    if (lamn >= -eps_l).all() and (gapn[~act] <= eps_g).all():
        return un
    return None


if __name__ == '__main__':
    p = orc.MpcParams(T=20)
    d = load(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/qp_sample.npz')
    N = len(d['x0']) if len(sys.argv) < 3 else int(sys.argv[2])
    sel = np.argsort(-d['iters'])[:N]
    rows = []
    for k in sel:
        H, g, G, h, u0 = dense(p, d['x0'][k], d['xref'][k], d['xbar'][k], d['re'][k], d['uw'][k])
        ub, itb, _ = ipm(H, g, G, h, u0)
        ue, act, lam = exact(H, g, G, h, ub)
        rows.append((d['iters'][k], itb, np.abs(ub - ue).max(), len(act)))
    r = np.array(rows)
    print('gpu iters == replica iters: %d / %d' % ((r[:, 0] == r[:, 1]).sum(), len(r)))
    print('distance to exact: max %.2e p99 %.2e median %.2e' % (r[:, 2].max(), np.quantile(r[:, 2], .99), np.median(r[:, 2])))
