"""Literal (un-condensed) restatement of the QP of main/lib/mpc.py:138-208, written against the reference source:
variables z = [x(NX,T+1) column-major by time ; u(2,T)], objective exactly as the cvxpy expression sums it, equality and
inequality rows as the constraints list.  Used only to certify oracle/GPU solutions (KKT of THIS problem) and to
cross-check against scipy; it shares no code with oracle.c or the HIP kernel.

p.model == 1 selects the problem of main/lib/mpc_jerk.py:143-208 instead: NX = 5 (the fifth state integrates the
acceleration input, mpc_jerk.py:62-78), the jerk term of line 190, only x[:4, 0] pinned (line 193) -- the fifth state's
initial value is a free variable of the problem."""
import math

import numpy as np


def build(p, x0, xref, xbar, reaches_end):
    """returns P, q, c0 (objective z'Pz + q'z + c0), Aeq, beq, G, h and index helpers ix(i,t), iu(j,t)"""
    T = p.T
    jerk = getattr(p, 'model', 0) == 1
    NX = 5 if jerk else 4
    nz = NX * (T + 1) + 2 * T
    if jerk:                        # xref / xbar of mpc_jerk.py have a fifth row of zeros (lines 91, 115)
        xref = np.vstack([np.asarray(xref)[:4], np.zeros((1, T + 1))])
        xbar = np.vstack([np.asarray(xbar)[:4], np.zeros((1, T + 1))])

    def ix(i, t):
        return NX * t + i

    def iu(j, t):
        return NX * (T + 1) + 2 * t + j
    P = np.zeros((nz, nz)); q = np.zeros(nz); c0 = 0.0
    Aeq, beq, G, h = [], [], [], []
    Qf = np.diag([v * T for v in p.Qf_base] + ([0.0] if jerk else []))
    for t in range(T + 1):
        if t > 0:
            if not reaches_end[t]:
                for ang, w in ((xref[3, t] + 0.5 * np.pi, p.w_perp), (xref[3, t], p.w_para)):
                    c, s = np.cos(ang), np.sin(ang)
                    M = np.array([[c ** 2, c * s], [c * s, s ** 2]]) * w
                    r = xref[:2, t]
                    idx = [ix(0, t), ix(1, t)]
                    P[np.ix_(idx, idx)] += M; q[idx] += -2 * M @ r; c0 += r @ M @ r
                Mq = np.diag(p.Q_v_yaw); r = xref[2:4, t]; idx = [ix(2, t), ix(3, t)]
                P[np.ix_(idx, idx)] += Mq; q[idx] += -2 * Mq @ r; c0 += r @ Mq @ r
            else:
                r = xref[:, t]; idx = [ix(i, t) for i in range(NX)]
                P[np.ix_(idx, idx)] += Qf; q[idx] += -2 * Qf @ r; c0 += r @ Qf @ r
        if t < T:
            v, phi, delta, dt, L = xbar[2, t], xbar[3, t], 0.0, p.dt, p.L
            A = np.eye(NX)
            A[0, 2] = dt * math.cos(phi); A[0, 3] = -dt * v * math.sin(phi)
            A[1, 2] = dt * math.sin(phi); A[1, 3] = dt * v * math.cos(phi)
            A[3, 2] = dt * math.tan(delta) / L
            B = np.zeros((NX, 2)); B[2, 0] = dt; B[3, 1] = dt * v / (L * math.cos(delta) ** 2)
            C = np.zeros(NX)
            C[:4] = [dt * v * math.sin(phi) * phi, -dt * v * math.cos(phi) * phi, 0.0, -dt * v * delta / (L * math.cos(delta) ** 2)]
            if jerk:
                A[2, 4] = dt; B[4, 0] = dt          # mpc_jerk.py:73, 78
            for i in range(NX):     # x[:, t+1] == A x[:, t] + B u[:, t] + C
                row = np.zeros(nz); row[ix(i, t + 1)] = 1.0
                for j in range(NX):
                    row[ix(j, t)] -= A[i, j]
                for j in range(2):
                    row[iu(j, t)] -= B[i, j]
                Aeq.append(row); beq.append(C[i])
            Rt = np.diag(p.R_end) if reaches_end[t] else np.diag(p.R)
            idx = [iu(0, t), iu(1, t)]
            P[np.ix_(idx, idx)] += Rt
        if t < T - 1:
            Rd = np.diag(p.Rd)
            for j in range(2):
                a, b = iu(j, t + 1), iu(j, t)
                P[a, a] += Rd[j, j]; P[b, b] += Rd[j, j]; P[a, b] -= Rd[j, j]; P[b, a] -= Rd[j, j]
            if jerk:                 # jerk_penalty_weight * square(x[4, t+1] - x[4, t]), mpc_jerk.py:190
                a, b = ix(4, t + 1), ix(4, t)
                w = p.jerk_weight
                P[a, a] += w; P[b, b] += w; P[a, b] -= w; P[b, a] -= w
            for sgn in (1.0, -1.0):  # |u[1,t+1] - u[1,t]| <= MAX_DSTEER * dt
                row = np.zeros(nz); row[iu(1, t + 1)] = sgn; row[iu(1, t)] = -sgn
                G.append(row); h.append(p.max_dsteer * p.dt)
    for i in range(4):              # x[:, 0] == x0  (mpc_jerk.py:193: x[:4, 0] == x0)
        row = np.zeros(nz); row[ix(i, 0)] = 1.0
        Aeq.append(row); beq.append(x0[i])
    for t in range(T + 1):
        row = np.zeros(nz); row[ix(2, t)] = 1.0; G.append(row); h.append(p.max_speed)
        row = np.zeros(nz); row[ix(2, t)] = -1.0; G.append(row); h.append(-p.min_speed)
    for t in range(T):
        row = np.zeros(nz); row[iu(0, t)] = 1.0; G.append(row); h.append(p.max_accel)
        row = np.zeros(nz); row[iu(0, t)] = -1.0; G.append(row); h.append(-p.max_decel)
        row = np.zeros(nz); row[iu(1, t)] = 1.0; G.append(row); h.append(p.max_steer)
        row = np.zeros(nz); row[iu(1, t)] = -1.0; G.append(row); h.append(p.max_steer)
    return P, q, c0, np.array(Aeq), np.array(beq), np.array(G), np.array(h), ix, iu


def pack(p, x, u):
    T = p.T
    return np.concatenate([np.asarray(x).T.reshape(-1), np.asarray(u).T.reshape(-1)])


def kkt_certificate(p, x0, xref, xbar, reaches_end, x, u):
    """KKT residuals of the literal problem at (x,u): equality violation, inequality violation, and the best
    multipliers: min || 2Pz + q + Aeq' nu + G' lam ||  over nu free, lam >= 0 (NNLS over ALL inequality rows); a KKT
    point needs stat ~ 0 and comp = max lam_i * slack_i ~ 0."""
    from scipy.optimize import nnls
    P, q, c0, Aeq, beq, G, h, _, _ = build(p, x0, xref, xbar, reaches_end)
    z = pack(p, x, u)
    eq = np.abs(Aeq @ z - beq).max()
    slack = h - G @ z
    ineq = max(0.0, float((-slack).max()))
    grad = 2 * P @ z + q
    # rows with visible slack cannot carry a multiplier at a KKT point: penalise them so NNLS prefers tight rows
    pen = 1e3 * np.maximum(slack, 0.0)
    M = np.concatenate([np.concatenate([Aeq.T, -Aeq.T, G.T], axis=1),
                        np.concatenate([np.zeros((len(h), 2 * len(beq))), np.diag(pen)], axis=1)], axis=0)
    scale = max(1.0, np.abs(grad).max())
    rhs = np.concatenate([-grad / scale, np.zeros(len(h))])
    sol, _ = nnls(M / scale * 1.0, rhs, maxiter=50 * M.shape[1])
    lam = sol[2 * len(beq):] * 1.0
    nu = sol[:len(beq)] - sol[len(beq):2 * len(beq)]
    stat = np.abs(grad + Aeq.T @ nu * 1.0 + G.T @ lam).max()
    comp = float((lam * np.maximum(slack, 0.0)).max())
    return dict(eq=eq, ineq=ineq, stat=stat, comp=comp, obj=float(z @ P @ z + q @ z + c0), grad_scale=scale)


def exact_solution(p, x0, xref, xbar, reaches_end, z_start, max_rounds=60, add_first=False):
    """The minimiser of the literal problem by an active-set iteration started from the active set of `z_start`: solve the
    equality-constrained KKT system [2P Aeq' Ga'; Aeq 0 0; Ga 0 0] with numpy.linalg.solve, drop the most negative multiplier,
    add violated rows, until multipliers >= 0 and no row is violated -- at which point (z, nu, lam) IS a KKT point of the
    literal problem (strictly convex on the feasible set => the unique optimum, the point ECOS converges to).
    Rows on x[2, 0] (mpc.py:187-188 at t = 0) duplicate the initial-state equality and are left to it.
    add_first: violated rows enter BEFORE any multiplier may leave (a start far from the optimum, e.g. the empty set, can make the
    simultaneous rule cycle; the end test -- and therefore the result -- is the same).
    Returns dict(z, active, lam, nu, rounds)."""
    P, q, c0, Aeq, beq, G, h, ix, iu = build(p, x0, xref, xbar, reaches_end)
    n, me = len(q), len(beq)
    skip = set(i for i in range(len(h)) if np.count_nonzero(G[i]) == 1 and abs(G[i][ix(2, 0)]) == 1.0)
    act = set(int(i) for i in np.where(h - G @ z_start < 1e-6)[0] if i not in skip)
    seen_sets, careful = set(), False
    for rounds in range(max_rounds):
        a = sorted(act)
        Ga, ha, ma = G[a], h[a], len(a)
        K = np.zeros((n + me + ma, n + me + ma))
        K[:n, :n] = 2 * P; K[:n, n:n + me] = Aeq.T; K[:n, n + me:] = Ga.T; K[n:n + me, :n] = Aeq; K[n + me:, :n] = Ga
        # a vanishing dual regularisation keeps the system solvable on a degenerate vertex (linearly dependent active rows: the multipliers
        # are then shared out among them); it relaxes an active row by 1e-14 * lam
        K[n + me:, n + me:] = -1e-14 * np.eye(ma)
        s = np.linalg.solve(K, np.concatenate([-q, beq, ha]))
        z, nu, lam = s[:n], s[n:n + me], s[n + me:]
        slack = h - G @ z
        drop = [(lam[i], a[i]) for i in range(ma) if lam[i] < -1e-7]
        add = [i for i in range(len(h)) if i not in act and i not in skip and slack[i] < -1e-10]
        if not drop and not add:
            return dict(z=z, active=a, lam=lam, nu=nu, rounds=rounds, slack=slack, obj=float(z @ P @ z + q @ z + c0),
                        eq=float(np.abs(Aeq @ z - beq).max()), stat=float(np.abs(2 * P @ z + q + Aeq.T @ nu + Ga.T @ lam).max()))
        key = frozenset(act)
        if key in seen_sets:
            careful = True           # the working set came back: from here on ONE row per round (the most violated one first, else the most negative multiplier)
        seen_sets.add(key)
        if careful:
            if add:
                act.add(min(add, key=lambda i: slack[i]))
            else:
                act.discard(min(drop)[1])
            continue
        if drop and not (add_first and add):
            act.discard(min(drop)[1])
        act.update(add)
    raise RuntimeError('active-set iteration did not settle')

