"""dev aid (CPU): how often does an active-set solve started from the set the WARM START sits on end a constrained QP, and in how
many passes?    Corpus: gpurun_out/qp_sample.npz (5 x 1024 problems of the benchmark's closed loop with their
warm starts) or any file of the same layout.  A "pass" is one penalised solve (rho on the guessed rows) + the KKT sign test + the
flips, i.e. one round of the stage solver.

    python scripts/warm_set_lab.py [corpus.npz] [N]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle_py as orc
from scripts.ipm_lab import load, dense, exact

RHO, EPS_L, EPS_G = 1e8, 1e-9, 1e-9
T = 20


def rows_shift(act):
    """the set one stage earlier: row of stage t takes the flag of stage t + 1 (last stage keeps its own)"""
    a = act.copy()
    box = act[:4 * T].reshape(T, 4); a[:4 * T] = np.vstack([box[1:], box[-1:]]).ravel()
    o = 4 * T
    rate = act[o:o + 2 * (T - 1)].reshape(T - 1, 2); a[o:o + 2 * (T - 1)] = np.vstack([rate[1:], rate[-1:]]).ravel()
    o += 2 * (T - 1)
    sp = act[o:].reshape(T, 2); a[o:] = np.vstack([sp[1:], sp[-1:]]).ravel()
    return a


def pdas(H, g, G, h, u, act, lam0=None, max_pass=8, al_second=True):
    """penalised active-set passes from the guess `act`; returns (u, passes, ok).  Multiplier estimate lam0 (zeros if None).
    With al_second an accepted first pass whose multiplier moved by more than 1e-1 is followed by one more solve (counted)."""
    m = len(h)
    lam = np.zeros(m) if lam0 is None else lam0.copy()
    act = act.copy()
    for k in range(1, max_pass + 1):
        gap = G @ u - h
        nu = np.where(act, lam + RHO * gap, 0.0)
        M = H + RHO * (G[act].T @ G[act])
        du = np.linalg.solve(M, -(g + H @ u + G.T @ nu))
        gapn = gap + G @ du
        lamn = np.where(act, lam + RHO * gapn, 0.0)
        drop = act & (lamn < -EPS_L)
        add = (~act) & (gapn > EPS_G)
        if not drop.any() and not add.any():
            un = u + du
            if al_second and np.abs(lamn - lam)[act].max(initial=0.0) / RHO > 1e-9:
                # one more augmented-Lagrangian step with the multipliers just found
                gap2 = G @ un - h
                nu2 = np.where(act, lamn + RHO * gap2, 0.0)
                du2 = np.linalg.solve(M, -(g + H @ un + G.T @ nu2))
                return un + du2, k + 1, True
            return un, k, True
        act = (act & ~drop) | add
    return u, max_pass, False


if __name__ == '__main__':
    p = orc.MpcParams(T=T)
    path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/qp_sample.npz'
    d = load(path)
    raw = np.load(path)
    if 's0/prev_iters' in raw.files:          # scripts/warm_set_harvest.py: the work-queue key of every problem
        d['prev_iters'], d['moved'] = raw['s0/prev_iters'], raw['s0/moved']
    N = len(d['x0']) if len(sys.argv) < 3 else int(sys.argv[2])
    rng = np.random.default_rng(0)
    sel = rng.permutation(len(d['x0']))[:N]
    stats = dict(n=0, uncon=0)
    res = {k: [] for k in ('it', 'warm', 'warm_shift', 'trialviol', 'true_eq_warm', 'true_eq_shift', 'err_warm', 'nact', 'key', 'first_warm', 'first_shift')}
    for k in sel:
        H, g, G, h, u0 = dense(p, d['x0'][k], d['xref'][k], d['xbar'][k], d['re'][k], d['uw'][k])
        w = u0 - np.linalg.solve(H, H @ u0 + g)
        stats['n'] += 1
        if (G @ w - h <= 0).all():
            stats['uncon'] += 1
            continue
        ue, a_true, _ = exact(H, g, G, h, np.concatenate([d['u'][k][0][:, None], d['u'][k][1][:, None]], axis=1).ravel())
        true = np.zeros(len(h), bool); true[a_true] = True
        warm = (G @ u0 - h) >= -1e-7
        sh = rows_shift(warm)
        tv = (G @ w - h) > 0
        res['it'].append(d['iters'][k]); res['nact'].append(true.sum())
        res['key'].append((int(d['prev_iters'][k]) + (11 if d['moved'][k] else 0)) if 'prev_iters' in d else -1)
        res['true_eq_warm'].append((warm == true).all()); res['true_eq_shift'].append((sh == true).all())
        for name, a in (('warm', warm), ('warm_shift', sh), ('trialviol', tv)):
            un, passes, ok = pdas(H, g, G, h, u0, a)
            res[name].append(passes if ok else 99)
            if name != 'trialviol':             # "first try": the very first penalised solve on the guess is a KKT point (sign test passes)
                res['first_' + name.replace('warm_', '')].append(pdas(H, g, G, h, u0, a, max_pass=1, al_second=False)[2])
            if name == 'warm':
                res['err_warm'].append(np.abs(un - ue).max() if ok else np.nan)
    print('%d problems, %d end in the trial pass' % (stats['n'], stats['uncon']))
    it = np.array(res['it'])
    print('constrained: %d; interior-point iterations mean %.2f max %d  => rounds today = 1 + it + ~1.07' % (len(it), it.mean(), it.max()))
    print('guess == true active set: warm %.3f, shifted %.3f; |true set| mean %.1f' % (np.mean(res['true_eq_warm']), np.mean(res['true_eq_shift']), np.mean(res['nact'])))
    for name in ('warm', 'warm_shift', 'trialviol'):
        v = np.array(res[name])
        print('%-11s passes: ' % name + ' '.join('%d:%.3f' % (q, np.mean(v == q)) for q in range(1, 9)) + '  fail %.3f  mean(ok) %.2f' % (np.mean(v == 99), v[v < 99].mean()))
    e = np.array(res['err_warm'])
    print('distance to exact after the warm passes: max %.2e median %.2e' % (np.nanmax(e), np.nanmedian(e)))
    # by iteration count of the closed-loop solve (the queue key of the next step)
    v = np.array(res['warm'])
    for lo, hi in ((0, 5), (5, 7), (7, 10), (10, 99)):
        s = (it >= lo) & (it < hi)
        if s.any():
            print('  it in [%d,%d): n %4d  warm passes <=2: %.3f  <=3: %.3f  fail(8): %.3f' % (lo, hi, s.sum(), np.mean(v[s] <= 2), np.mean(v[s] <= 3), np.mean(v[s] == 99)))
    key = np.array(res['key'])
    if (key >= 0).any():
        print('by work-queue key (previous iterations + 11 if the path cut moved):')
        fw, fs, vs = np.array(res['first_warm']), np.array(res['first_shift']), np.array(res['warm_shift'])
        for lo, hi in ((0, 1), (1, 5), (5, 7), (7, 11), (11, 12), (12, 18), (18, 99)):
            s = (key >= lo) & (key < hi)
            if s.any():
                print('  key in [%2d,%2d): n %4d  first-try acceptance: warm set %.3f, shifted %.3f | accepted within 3 passes: warm %.3f shifted %.3f | '
                      'interior-point iterations of the real solve: mean %.1f' % (lo, hi, s.sum(), fw[s].mean(), fs[s].mean(), np.mean(v[s] <= 3), np.mean(vs[s] <= 3), it[s].mean()))
        print('  all: first-try acceptance warm %.3f shifted %.3f' % (fw.mean(), fs.mean()))
