/*
 * oracle.c -- CPU restatement (plain C11, float64) of the reference hot path.
 * TEST INFRASTRUCTURE ONLY -- see oracle.h.  Build: `make -C oracle` (gcc -O2 -ffp-contract=off).
 * Reference paths are relative to /root/reference/main.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NX 4
#define NU 2

/* ------------------------------------------------------------------ lib/mpc.py:58-79 */
void orc_linear_model(double v, double phi, double delta, double dt, double L,
                      double *A, double *B, double *C) {
    memset(A, 0, 16 * sizeof(double));
    memset(B, 0, 8 * sizeof(double));
    memset(C, 0, 4 * sizeof(double));
    A[0] = A[5] = A[10] = A[15] = 1.0;
    A[0 * 4 + 2] = dt * cos(phi);
    A[0 * 4 + 3] = -dt * v * sin(phi);
    A[1 * 4 + 2] = dt * sin(phi);
    A[1 * 4 + 3] = dt * v * cos(phi);
    A[3 * 4 + 2] = dt * tan(delta) / L;
    B[2 * 2 + 0] = dt;
    double cd = cos(delta);
    B[3 * 2 + 1] = dt * v / (L * (cd * cd));
    C[0] = dt * v * sin(phi) * phi;
    C[1] = -dt * v * cos(phi) * phi;
    C[3] = -dt * v * delta / (L * (cd * cd));
}

/* ------------------------------------------------------------------ lib/mpc.py:129-135 */
void orc_xy_cost_mtx(double angle, double *M) {
    double c = cos(angle), s = sin(angle);
    M[0] = c * c; M[1] = c * s; M[2] = c * s; M[3] = s * s;
}

/* ------------------------------------------------------------------ lib/mpc.py:43-55 */
void orc_smooth_yaw(double *yaw, int32_t n) {
    for (int32_t i = 0; i + 1 < n; i++) {
        double d = yaw[i + 1] - yaw[i];
        while (d >= M_PI / 2.0) { yaw[i + 1] -= M_PI * 2.0; d = yaw[i + 1] - yaw[i]; }
        while (d <= -M_PI / 2.0) { yaw[i + 1] += M_PI * 2.0; d = yaw[i + 1] - yaw[i]; }
    }
}

/* ------------------------------------------------------------------ lib/trajectories.py:100-126
 * The three smallest distances in ascending order (numpy argpartition+argsort; ties, which numpy leaves
 * unspecified, are broken here by lower index). */
int32_t orc_nearest_index_in_direction(double x, double y, const double *cx, const double *cy, int32_t n,
                                       int32_t start, int32_t forward) {
    int32_t len = n - start;
    if (len <= 1) return start;
    if (len == 2) return forward ? 1 + start : start;
    double bd[3] = {INFINITY, INFINITY, INFINITY};
    int32_t bi[3] = {-1, -1, -1};
    for (int32_t i = 0; i < len; i++) {
        double dx = cx[start + i] - x, dy = cy[start + i] - y;
        double d = sqrt(dx * dx + dy * dy);
        if (d < bd[2]) {
            int k = 2;
            while (k > 0 && d < bd[k - 1]) { bd[k] = bd[k - 1]; bi[k] = bi[k - 1]; k--; }
            bd[k] = d; bi[k] = i;
        }
    }
    if (abs(bi[1] - bi[2]) == 2) return bi[0] + start;
    if (abs(bi[0] - bi[1]) == 1) {
        int32_t a = bi[0] > bi[1] ? bi[0] : bi[1], b = bi[0] < bi[1] ? bi[0] : bi[1];
        return (forward ? a : b) + start;
    }
    return -1; /* reference raises Exception("something wrong") */
}

/* ------------------------------------------------------------------ lib/mpc.py:86-109 */
int32_t orc_calc_ref_trajectory(const orc_mpc_params *p, const double *st, const double *cx, const double *cy,
                                const double *cyaw, const double *cv /* NULL: lib/mpc.py; else mpc_with_speed.py:103-104 */,
                                int32_t n, double dl, int32_t start_idx, double *xref, uint8_t *reaches_end) {
    return orc_calc_ref_trajectory_ov(p, st, cx, cy, cyaw, cv, NULL, n, dl, start_idx, xref, reaches_end);
}

/* the same with the `ov` argument of mpc.py:86,95-98: the speeds of the previous linearisation pass (T+1 values; mpc.py:226-237 passes
 * them from the second of MAX_ITER passes on) space the reference window instead of max(state.v, 10/3.6) */
int32_t orc_calc_ref_trajectory_ov(const orc_mpc_params *p, const double *st, const double *cx, const double *cy,
                                   const double *cyaw, const double *cv, const double *ov_prev /* T+1 or NULL */,
                                   int32_t n, double dl, int32_t start_idx, double *xref, uint8_t *reaches_end) {
    int32_t T = p->T, W = T + 1;
    int32_t s = orc_nearest_index_in_direction(st[0], st[1], cx, cy, n, start_idx, 1);
    if (s < 0) return -1;
    double ov = st[2] > 10.0 / 3.6 ? st[2] : 10.0 / 3.6;   /* max(state.v, 10/3.6) where ov is None (first pass) */
    double travel = 0.0;
    for (int32_t k = 0; k < W; k++) {
        double step = fabs(ov_prev ? ov_prev[k] : ov) * p->dt;
        travel = (k == 0) ? step : travel + step;          /* np.cumsum: sequential adds */
        long idx = (long)nearbyint(travel / dl);           /* np.rint: half-to-even under the default rounding mode */
        idx += s;
        if (idx > n - 1) idx = n - 1;
        xref[0 * W + k] = cx[idx];
        xref[1 * W + k] = cy[idx];
        xref[2 * W + k] = cv ? cv[idx] : 0.0;
        xref[3 * W + k] = cyaw[idx];
        reaches_end[k] = (idx == n - 1);
    }
    return s;
}

/* ------------------------------------------------------------------ lib/simulation.py:35-47 + bicycle/main.py:28-41 */
void orc_plant_step(const orc_mpc_params *p, double *s, double a, double delta) {
    if (delta > p->max_steer) delta = p->max_steer;
    if (delta < -p->max_steer) delta = -p->max_steer;
    double v = s[2], th = s[3];
    double xd = v * cos(th), yd = v * sin(th), td = (v / p->L) * tan(delta);
    s[0] += xd * p->dt;
    s[1] += yd * p->dt;
    s[3] += td * p->dt;
    v += a * p->dt;
    if (v > p->max_speed) v = p->max_speed;
    if (v < p->min_speed) v = p->min_speed;
    s[2] = v;
}

/* ------------------------------------------------------------------ lib/mpc.py:112-126 */
void orc_predict_motion(const orc_mpc_params *p, const double *x0, const double *oa, const double *od, double *xbar) {
    int32_t T = p->T, W = T + 1;
    double s[4] = {x0[0], x0[1], x0[2], x0[3]};
    for (int i = 0; i < 4; i++) xbar[i * W] = x0[i];
    for (int32_t t = 1; t <= T; t++) {
        orc_plant_step(p, s, oa[t - 1], od[t - 1]);
        for (int i = 0; i < 4; i++) xbar[i * W + t] = s[i];
    }
}

/* ------------------------------------------------------------------ lib/mpc.py:138-208 : QP construction
 * Generic dense condensing: x_t = S_t u + c_t with S_{t+1} = A_t S_t + B_t E_t, c_{t+1} = A_t c_t + C_t, c_0 = x0.
 * u is interleaved [a_0, delta_0, a_1, delta_1, ...].  Objective exactly as cvxpy sums it (no 1/2 factors),
 * expressed as 1/2 u'Hu + g'u (+const). Rows of G: per t (a<=, -a<=, d<=, -d<=), rate rows, speed rows t=1..T. */
int32_t orc_qp_build(const orc_mpc_params *p, const double *x0, const double *xref, const double *xbar,
                     const uint8_t *re, double *H, double *g, double *G, double *h, double *S, double *c) {
    int32_t T = p->T, W = T + 1, n = 2 * T;
    int32_t m = 4 * T + 2 * (T - 1) + 2 * T;
    memset(H, 0, sizeof(double) * n * n);
    memset(g, 0, sizeof(double) * n);
    memset(G, 0, sizeof(double) * m * n);
    memset(S, 0, sizeof(double) * W * 4 * n);
    for (int i = 0; i < 4; i++) c[i] = x0[i];
    for (int32_t t = 0; t < T; t++) {
        double A[16], B[8], C[4];
        orc_linear_model(xbar[2 * W + t], xbar[3 * W + t], 0.0 /* dref is zeroed, mpc.py:93 */, p->dt, p->L, A, B, C);
        const double *St = S + (size_t)t * 4 * n;
        double *Sn = S + (size_t)(t + 1) * 4 * n;
        for (int i = 0; i < 4; i++) {
            for (int k = 0; k < n; k++) {
                double acc = 0.0;
                for (int j = 0; j < 4; j++) acc += A[i * 4 + j] * St[j * n + k];
                Sn[i * n + k] = acc;
            }
            Sn[i * n + 2 * t + 0] += B[i * 2 + 0];
            Sn[i * n + 2 * t + 1] += B[i * 2 + 1];
            double acc = C[i];
            for (int j = 0; j < 4; j++) acc += A[i * 4 + j] * c[t * 4 + j];
            c[(t + 1) * 4 + i] = acc;
        }
    }
    /* state costs, t = 1..T (mpc.py:157-170) */
    for (int32_t t = 1; t <= T; t++) {
        double Wt[16];
        memset(Wt, 0, sizeof Wt);
        if (!re[t]) {
            double Mp[4], Ma[4];
            orc_xy_cost_mtx(xref[3 * W + t] + 0.5 * M_PI, Mp);
            orc_xy_cost_mtx(xref[3 * W + t], Ma);
            Wt[0] = Mp[0] * p->w_perp + Ma[0] * p->w_para;
            Wt[1] = Mp[1] * p->w_perp + Ma[1] * p->w_para;
            Wt[4] = Mp[2] * p->w_perp + Ma[2] * p->w_para;
            Wt[5] = Mp[3] * p->w_perp + Ma[3] * p->w_para;
            Wt[10] = p->Q_v_yaw[0];
            Wt[15] = p->Q_v_yaw[1];
        } else {
            for (int i = 0; i < 4; i++) Wt[i * 4 + i] = p->Qf[i];
        }
        const double *St = S + (size_t)t * 4 * n;
        double e[4], We[4];
        for (int i = 0; i < 4; i++) e[i] = c[t * 4 + i] - xref[i * W + t];
        for (int i = 0; i < 4; i++) { We[i] = 0; for (int j = 0; j < 4; j++) We[i] += Wt[i * 4 + j] * e[j]; }
        for (int a = 0; a < n; a++) {
            double WS[4];
            for (int i = 0; i < 4; i++) { WS[i] = 0; for (int j = 0; j < 4; j++) WS[i] += Wt[i * 4 + j] * St[j * n + a]; }
            for (int b = 0; b < n; b++) {
                double acc = 0;
                for (int i = 0; i < 4; i++) acc += St[i * n + b] * WS[i];
                H[b * n + a] += 2.0 * acc;
            }
            double acc = 0;
            for (int i = 0; i < 4; i++) acc += St[i * n + a] * We[i];
            g[a] += 2.0 * acc;
        }
    }
    /* input costs (mpc.py:177-180) and rate costs (mpc.py:182-183) */
    for (int32_t t = 0; t < T; t++) {
        const double *Rt = re[t] ? p->R_end : p->R;
        H[(2 * t) * n + 2 * t] += 2.0 * Rt[0];
        H[(2 * t + 1) * n + 2 * t + 1] += 2.0 * Rt[1];
    }
    for (int32_t t = 0; t + 1 < T; t++)
        for (int j = 0; j < 2; j++) {
            int a = 2 * t + j, b = 2 * (t + 1) + j;
            H[a * n + a] += 2.0 * p->Rd[j];
            H[b * n + b] += 2.0 * p->Rd[j];
            H[a * n + b] -= 2.0 * p->Rd[j];
            H[b * n + a] -= 2.0 * p->Rd[j];
        }
    /* constraints (mpc.py:184-191) */
    int32_t r = 0;
    for (int32_t t = 0; t < T; t++) {
        G[r * n + 2 * t] = 1.0;      h[r++] = p->max_accel;
        G[r * n + 2 * t] = -1.0;     h[r++] = -p->max_decel;
        G[r * n + 2 * t + 1] = 1.0;  h[r++] = p->max_steer;
        G[r * n + 2 * t + 1] = -1.0; h[r++] = p->max_steer;
    }
    for (int32_t t = 0; t + 1 < T; t++) {
        G[r * n + 2 * t + 3] = 1.0;  G[r * n + 2 * t + 1] = -1.0; h[r++] = p->max_dsteer * p->dt;
        G[r * n + 2 * t + 3] = -1.0; G[r * n + 2 * t + 1] = 1.0;  h[r++] = p->max_dsteer * p->dt;
    }
    for (int32_t t = 1; t <= T; t++) {
        const double *St = S + (size_t)t * 4 * n;
        for (int k = 0; k < n; k++) G[r * n + k] = St[2 * n + k];
        h[r++] = p->max_speed - c[t * 4 + 2];
        for (int k = 0; k < n; k++) G[r * n + k] = -St[2 * n + k];
        h[r++] = c[t * 4 + 2] - p->min_speed;
    }
    return r;
}

#define ORC_STEP_FRACTION 0.999
#ifndef ORC_SLACK_FLOOR
#define ORC_SLACK_FLOOR 0.5      /* starting point of the iteration: s = max(slack, floor), lam = ORC_LAM0 */
#endif
#ifndef ORC_LAM0
#define ORC_LAM0 3.0          /* 1 until round 2: with separate step lengths 3 takes the hardest problems of a batch from 23 to 18 iterations */
#endif

/* dense Cholesky (lower), in place; returns 0 on success */
static int chol(double *M, int n) {
    for (int j = 0; j < n; j++) {
        double d = M[j * n + j];
        for (int k = 0; k < j; k++) d -= M[j * n + k] * M[j * n + k];
        if (!(d > 0.0)) return 1;
        d = sqrt(d);
        M[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = M[i * n + j];
            for (int k = 0; k < j; k++) s -= M[i * n + k] * M[j * n + k];
            M[i * n + j] = s / d;
        }
    }
    return 0;
}
static void chol_solve(const double *Lm, int n, double *b) {
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= Lm[i * n + k] * b[k];
        b[i] = s / Lm[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= Lm[k * n + i] * b[k];
        b[i] = s / Lm[i * n + i];
    }
}

/* Active-set polish (round 3; same rule in both HIP solvers).  An interior-point iterate sits ~sqrt(mu) away from the optimum on
 * weakly active rows (s ~ lam ~ sqrt(mu)), and the low curvature of the input cost (2R = 0.02) amplifies that: on the hard
 * closed-loop problems the iteration's answer was up to 1e-3 from the exact minimiser at its reduced-accuracy exit, 5e-5 at
 * mu = 1e-10.  Once the iterate is close (see `pol_pred` in orc_ipm_dense: the step that produced it predicted mu <= ORC_POLISH_MU and
 * small residuals -- decided at the end of the step so that the HIP solvers can run the polish round in place of the iterate's
 * first row pass), or at any exit, the rows with s < lam are taken as the active
 * set and ONE augmented-Lagrangian solve is made on it: (H + rho Ga'Ga) du = -(H u + g + Ga'(lam_a + rho gap_a)), new multipliers
 * lam_a + rho (gap_a + Ga du).  The point is accepted only if it is a KKT point: multipliers >= 0 on the active rows, no other
 * row violated; then it is the minimiser up to |lam - lam*| / rho.  Otherwise rows with a negative multiplier leave the set, violated
 * rows enter it and the solve is repeated, ORC_POLISH_TRIES times in all; if none is accepted nothing is
 * kept and the iteration goes on (or ends with its own iterate).  A polish solve is not counted as an iteration. */
#ifndef ORC_POLISH
#define ORC_POLISH 1
#endif
#define ORC_POLISH_MU 1e-5
#define ORC_POLISH_RP 1e-6
#define ORC_POLISH_RD 1e-3
#define ORC_POLISH_RHO 1e8
#define ORC_POLISH_TRIES 3
#define ORC_POLISH_EPS_L 1e-9
#define ORC_POLISH_EPS_G 1e-9

/* returns 1 and overwrites (u, s, lam) if a KKT point was found; M (n x n), du (n), w (m), gapn (m) are scratch */
static int polish(int32_t n, int32_t m, const double *H, const double *g, const double *G, const double *h,
                  double *u, double *s, double *lam, double *M, double *du, double *w, double *gapn) {
    unsigned char *act = malloc(m);
    int ok = 0;
    for (int i = 0; i < m; i++) act[i] = s[i] < lam[i];
    for (int tr = 0; tr < ORC_POLISH_TRIES && !ok; tr++) {
        memcpy(M, H, sizeof(double) * n * n);
        for (int i = 0; i < m; i++) {
            double gap = -h[i];
            const double *Gi = G + (size_t)i * n;
            for (int k = 0; k < n; k++) gap += Gi[k] * u[k];
            gapn[i] = gap;
            w[i] = act[i] ? (lam[i] + ORC_POLISH_RHO * gap) : 0.0;
            if (!act[i]) continue;
            for (int a = 0; a < n; a++) {
                if (Gi[a] == 0.0) continue;
                double da = ORC_POLISH_RHO * Gi[a];
                for (int b = 0; b <= a; b++) M[a * n + b] += da * Gi[b];
            }
        }
        if (chol(M, n)) continue;
        for (int k = 0; k < n; k++) {
            double a = g[k];
            for (int j = 0; j < n; j++) a += H[k * n + j] * u[j];
            for (int i = 0; i < m; i++) a += G[i * n + k] * w[i];
            du[k] = -a;
        }
        chol_solve(M, n, du);
        int change = 0;
        for (int i = 0; i < m; i++) {
            double gd = 0; for (int k = 0; k < n; k++) gd += G[i * n + k] * du[k];
            gapn[i] += gd;
            if (act[i]) {
                w[i] = lam[i] + ORC_POLISH_RHO * gapn[i];      /* the new multiplier */
                if (w[i] < -ORC_POLISH_EPS_L) { change = 1; w[i] = -1.0; }       /* marks the row for removal */
            } else if (gapn[i] > ORC_POLISH_EPS_G) { change = 1; w[i] = 1.0; }  /* marks the row for entry */
            else w[i] = 0.0;
        }
        if (!change) { ok = 1; break; }
        for (int i = 0; i < m; i++) {
            if (act[i] && w[i] == -1.0) act[i] = 0;
            else if (!act[i] && w[i] == 1.0) act[i] = 1;
        }
    }
    if (ok) {
        for (int k = 0; k < n; k++) u[k] += du[k];
        for (int i = 0; i < m; i++) {
            lam[i] = act[i] ? (w[i] > 0.0 ? w[i] : 0.0) : 0.0;
            s[i] = -gapn[i] > 0.0 ? -gapn[i] : 0.0;
        }
    }
    free(act);
    return ok;
}

/* ------------------------------------------------------------------ lib/mpc.py:193-206 : the solve.
 * The reference hands the problem to ECOS (an interior-point SOCP code).  The problem is a strictly convex QP
 * (unique minimiser), restated here as a dense Mehrotra predictor-corrector primal-dual interior-point method
 * on the condensed form.  Output: x (4,T+1), u (2,T) laid out as the reference's x.value / u.value. */
/* the interior-point iteration on a dense problem  min 1/2 w'Hw + g'w  s.t. Gw <= h, from the start w (in-out).
 * Shared by the 4-state problem (orc_qp_solve) and the 5-state jerk variant (oracle_jerk.c). */
int32_t orc_ipm_dense(const orc_mpc_params *p, int32_t n, int32_t m, const double *H, const double *g, const double *G,
                      const double *h, double *u, double *lam_out, int32_t *iters, double *kkt4) {
    double *M = malloc(sizeof(double) * n * n);
    double *du = malloc(sizeof(double) * n), *rd = malloc(sizeof(double) * n);
    double *s = malloc(sizeof(double) * m), *lam = malloc(sizeof(double) * m), *rp = malloc(sizeof(double) * m);
    double *ds = malloc(sizeof(double) * m), *dl = malloc(sizeof(double) * m), *rc = malloc(sizeof(double) * m);
    double *dsa = malloc(sizeof(double) * m), *dla = malloc(sizeof(double) * m), *w = malloc(sizeof(double) * m);
    int32_t status = ORC_MAXITER, it = 0;
    double res_d = 0, res_p = 0, mu = 0;
    const double tol_loose = p->tol > 1e-7 ? p->tol : 1e-7;
    int loose = 0, loose_run = 0, pol_pred = 0;
    for (int i = 0; i < m; i++) {
        double gi = 0; for (int k = 0; k < n; k++) gi += G[i * n + k] * u[k];
        double si = h[i] - gi;
        s[i] = si > ORC_SLACK_FLOOR ? si : ORC_SLACK_FLOOR;
        lam[i] = ORC_LAM0;
    }
    double gnorm = 1.0, hnorm = 1.0;
    for (int k = 0; k < n; k++) if (fabs(g[k]) > gnorm) gnorm = fabs(g[k]);
    for (int i = 0; i < m; i++) if (fabs(h[i]) > hnorm) hnorm = fabs(h[i]);

    /* Trial step (round 2, same rule in both HIP solvers): the minimiser of the objective alone, w = u - H^-1 (H u + g).  If it
     * violates no row it IS the solution of the problem (lam = 0 satisfies the KKT conditions exactly) and the interior-point
     * iteration is not needed: 0 iterations.  About two thirds of the closed-loop problems end here (no active constraint);
     * the iteration would spend 4 iterations on each of them walking lam from 1 down to 1e-10.  Otherwise nothing is kept. */
    {
        memcpy(M, H, sizeof(double) * n * n);
        int ok = chol(M, n) == 0;
        if (ok) {
            for (int k = 0; k < n; k++) { double a = g[k]; for (int j = 0; j < n; j++) a += H[k * n + j] * u[j]; du[k] = -a; }
            chol_solve(M, n, du);
            for (int k = 0; k < n; k++) rd[k] = u[k] + du[k];           /* the candidate */
            for (int i = 0; i < m && ok; i++) {
                double a = -h[i];
                for (int k = 0; k < n; k++) a += G[i * n + k] * rd[k];
                if (!(a <= 0.0)) ok = 0;
            }
        }
        if (ok) {
            for (int k = 0; k < n; k++) u[k] = rd[k];
            for (int i = 0; i < m; i++) lam[i] = 0.0;
            status = ORC_OK; it = 0; mu = 0.0;
            goto finish;
        }
    }

    for (it = 0; it <= p->max_iter; it++) {
        res_d = 0; res_p = 0; mu = 0;
        for (int k = 0; k < n; k++) {
            double a = g[k];
            for (int j = 0; j < n; j++) a += H[k * n + j] * u[j];
            for (int i = 0; i < m; i++) a += G[i * n + k] * lam[i];
            rd[k] = a; if (fabs(a) > res_d) res_d = fabs(a);
        }
        for (int i = 0; i < m; i++) {
            double a = s[i] - h[i];
            for (int k = 0; k < n; k++) a += G[i * n + k] * u[k];
            rp[i] = a; if (fabs(a) > res_p) res_p = fabs(a);
            mu += s[i] * lam[i];
        }
        mu /= m;
        const int conv = res_d <= p->tol * gnorm && res_p <= p->tol * hnorm && mu <= p->tol;
        /* reduced-accuracy acceptance when the iteration cannot continue (the reference accepts ECOS's
         * OPTIMAL_INACCURATE too, mpc.py:196) */
        loose = (res_d <= tol_loose * gnorm && res_p <= tol_loose * hnorm && mu <= tol_loose);
        /* stagnation exit: the stationarity residual of badly conditioned instances stalls at its rounding floor while mu keeps
         * collapsing; after 4 consecutive reduced-accuracy iterates stop before the factorisation degrades them */
        loose_run = loose ? loose_run + 1 : 0;
        const int stop = conv || loose_run >= 4 || it == p->max_iter;
        /* active-set polish: at every exit, and as soon as the iterate is close enough for the rows with s < lam to be the active set */
        if (ORC_POLISH && (stop || pol_pred)
            && polish(n, m, H, g, G, h, u, s, lam, M, du, w, dsa)) {
            status = ORC_OK; mu = 0.0;
            for (int i = 0; i < m; i++) mu += s[i] * lam[i];
            mu /= m;
            break;
        }
        if (conv || loose_run >= 4) { status = ORC_OK; break; }
        if (it == p->max_iter) { if (loose) status = ORC_OK; break; }
        /* M = H + G' D G */
        memcpy(M, H, sizeof(double) * n * n);
        for (int i = 0; i < m; i++) {
            double d = lam[i] / s[i];
            const double *Gi = G + (size_t)i * n;
            for (int a = 0; a < n; a++) {
                if (Gi[a] == 0.0) continue;
                double da = d * Gi[a];
                for (int b = 0; b <= a; b++) M[a * n + b] += da * Gi[b];
            }
        }
        if (chol(M, n)) { status = loose ? ORC_OK : ORC_NUMERIC; break; }
        /* predictor */
        for (int i = 0; i < m; i++) w[i] = -lam[i] + (lam[i] / s[i]) * rp[i];   /* (-rc + lam*rp)/s with rc = s*lam */
        for (int k = 0; k < n; k++) { double a = -rd[k]; for (int i = 0; i < m; i++) a -= G[i * n + k] * w[i]; du[k] = a; }
        chol_solve(M, n, du);
        double alpha = 1.0;
        for (int i = 0; i < m; i++) {
            double gd = 0; for (int k = 0; k < n; k++) gd += G[i * n + k] * du[k];
            dsa[i] = -rp[i] - gd;
            dla[i] = -lam[i] - (lam[i] / s[i]) * dsa[i];
            if (dsa[i] < 0 && -s[i] / dsa[i] < alpha) alpha = -s[i] / dsa[i];
            if (dla[i] < 0 && -lam[i] / dla[i] < alpha) alpha = -lam[i] / dla[i];
        }
        double mu_aff = 0;
        for (int i = 0; i < m; i++) mu_aff += (s[i] + alpha * dsa[i]) * (lam[i] + alpha * dla[i]);
        mu_aff /= m;
        double sigma = mu_aff / mu; sigma = sigma * sigma * sigma;
        /* centring target, never below a tenth of the tolerance: once mu has converged, driving it further down only worsens the
         * conditioning of M (lam/s grows without bound) while the stationarity residual sits at its rounding floor */
        double smu = sigma * mu; if (smu < 0.1 * p->tol) smu = 0.1 * p->tol;
        /* corrector */
        for (int i = 0; i < m; i++) {
            rc[i] = s[i] * lam[i] + alpha * (dsa[i] * dla[i]) - smu;   /* second-order term damped by the affine step length */
            w[i] = (-rc[i] + lam[i] * rp[i]) / s[i];
        }
        for (int k = 0; k < n; k++) { double a = -rd[k]; for (int i = 0; i < m; i++) a -= G[i * n + k] * w[i]; du[k] = a; }
        chol_solve(M, n, du);
        double amax_p = 1e300, amax_d = 1e300;
        for (int i = 0; i < m; i++) {
            double gd = 0; for (int k = 0; k < n; k++) gd += G[i * n + k] * du[k];
            ds[i] = -rp[i] - gd;
            dl[i] = -(rc[i] + lam[i] * ds[i]) / s[i];
            if (ds[i] < 0 && -s[i] / ds[i] < amax_p) amax_p = -s[i] / ds[i];
            if (dl[i] < 0 && -lam[i] / dl[i] < amax_d) amax_d = -lam[i] / dl[i];
        }
        /* step to the boundary: 0.999 of the way (0.995 in round 1: on the closed-loop workload the larger fraction saves 0.8 of 6.1
         * iterations on average, tail unchanged; same constant in both HIP solvers).
         * SEPARATE step lengths for the primal side (u, s) and the multipliers (round 2): the creeping instances are blocked by a
         * slack in one iteration and by a multiplier in the next, and a common step length pays for both every time.  On the
         * constrained closed-loop problems: mean 8.4 -> 7.9 iterations, 99th percentile 15 -> 14, maximum 20 -> 16; the golden cold
         * starts 19 / 15 / 22 -> 15 / 14 / 17 at T = 10 / 13 / 20; a batch lasts as long as its slowest problem. */
        double alpha_p = ORC_STEP_FRACTION * amax_p, alpha_d = ORC_STEP_FRACTION * amax_d;
        if (alpha_p > 1.0) alpha_p = 1.0;
        if (alpha_d > 1.0) alpha_d = 1.0;
        /* centrality safeguard (wide neighbourhood): shorten the step until min_i s_i*lam_i >= 1e-3 * mu at the new point;
         * plain Mehrotra otherwise cycles on poorly centred iterates (mu oscillates, residuals -> 0) */
        double psum = 0.0;
        for (int tr = 0; tr < 6; tr++) {
            double pmin = 1e300;
            psum = 0.0;
            for (int i = 0; i < m; i++) {
                double pr = (s[i] + alpha_p * ds[i]) * (lam[i] + alpha_d * dl[i]);
                if (pr < pmin) pmin = pr;
                psum += pr;
            }
            if (pmin >= 1e-3 * (psum / m)) break;
            alpha_p *= 0.7; alpha_d *= 0.7;
        }
        /* is the next iterate close enough to polish?  mu of the new point is psum / m (exactly, while the step was not shortened after
         * the last evaluation); the primal residual shrinks by 1 - alpha_p, the dual one by about the smaller of the two steps */
        {
            double keep = 1.0 - (alpha_p < alpha_d ? alpha_p : alpha_d);
            pol_pred = psum / m <= ORC_POLISH_MU && (1.0 - alpha_p) * res_p <= ORC_POLISH_RP * hnorm && keep * res_d <= ORC_POLISH_RD * gnorm;
        }
        for (int k = 0; k < n; k++) u[k] += alpha_p * du[k];
        for (int i = 0; i < m; i++) { s[i] += alpha_p * ds[i]; lam[i] += alpha_d * dl[i]; }
    }
finish:
    *iters = it;
    if (kkt4) {
        /* certificate on the final iterate: stationarity, primal violation, complementarity, max(lam<0) */
        double st = 0, pv = 0, cp = 0;
        for (int k = 0; k < n; k++) {
            double a = g[k];
            for (int j = 0; j < n; j++) a += H[k * n + j] * u[j];
            for (int i = 0; i < m; i++) a += G[i * n + k] * lam[i];
            if (fabs(a) > st) st = fabs(a);
        }
        for (int i = 0; i < m; i++) {
            double a = -h[i];
            for (int k = 0; k < n; k++) a += G[i * n + k] * u[k];
            if (a > pv) pv = a;
            if (fabs(a * lam[i]) > cp) cp = fabs(a * lam[i]);
        }
        kkt4[0] = st; kkt4[1] = pv; kkt4[2] = cp; kkt4[3] = mu;
    }
    if (lam_out) for (int i = 0; i < m; i++) lam_out[i] = lam[i];
    free(M); free(du); free(rd); free(s); free(lam); free(rp); free(ds); free(dl); free(rc); free(dsa); free(dla); free(w);
    return status;
}

int32_t orc_qp_solve(const orc_mpc_params *p, const double *x0, const double *xref, const double *xbar,
                     const uint8_t *re, const double *u_warm, double *x_out, double *u_out, double *lam_out,
                     int32_t *iters, double *kkt4) {
    int32_t T = p->T, W = T + 1, n = 2 * T, mcap = 8 * T;
    double *H = malloc(sizeof(double) * n * n), *g = malloc(sizeof(double) * n);
    double *G = malloc(sizeof(double) * mcap * n), *h = malloc(sizeof(double) * mcap);
    double *S = malloc(sizeof(double) * W * 4 * n), *c = malloc(sizeof(double) * W * 4);
    double *u = calloc(n, sizeof(double));
    int32_t m = orc_qp_build(p, x0, xref, xbar, re, H, g, G, h, S, c);
    int32_t status;
    /* x[2,0] bounds are constant rows (mpc.py:187-188 include t=0) */
    if (x0[2] > p->max_speed + 1e-9 || x0[2] < p->min_speed - 1e-9) {
        status = ORC_INFEASIBLE; *iters = 0;
        if (kkt4) kkt4[0] = kkt4[1] = kkt4[2] = kkt4[3] = 0.0;
    } else {
        if (u_warm) for (int t = 0; t < T; t++) { u[2 * t] = u_warm[t]; u[2 * t + 1] = u_warm[T + t]; }
        status = orc_ipm_dense(p, n, m, H, g, G, h, u, lam_out, iters, kkt4);
    }
    for (int t = 0; t < T; t++) { u_out[t] = u[2 * t]; u_out[T + t] = u[2 * t + 1]; }
    for (int t = 0; t <= T; t++)
        for (int i = 0; i < 4; i++) {
            double a = c[t * 4 + i];
            for (int k = 0; k < n; k++) a += S[((size_t)t * 4 + i) * n + k] * u[k];
            x_out[i * W + t] = a;
        }
    free(H); free(g); free(G); free(h); free(S); free(c); free(u);
    return status;
}

/* ------------------------------------------------------------------ lib/maths.py:4-10 */
static double normalize_angle(double th) {
    const double tau = 2.0 * M_PI;
    th = fmod(th, tau);
    if (th < 0) th += tau;          /* python float % : result has the sign of the divisor */
    if (th >= tau) th = 0.0;        /* python: (-1e-17) % tau == tau after rounding -> stays tau; handled below */
    if (th >= M_PI) th -= tau;
    return th;
}

/* ------------------------------------------------------------------ motion_primitive_search.py:87-121
 * transform (linalg.py:4-54) + per-obstacle half-plane test (obstacles.py:157-176).  `collide[k]` is the
 * reference's `collides` flag; `nbr` is filled for every primitive (callers ignore it where collide = 1).
 * Products are kept un-fused in the two orders numpy/OpenBLAS was observed to use:
 *   N>=2 points: (x*m0 + y*m1) + t with the first product rounded, second fused;  N==1: fma(x, m0, y*m1) + t. */
void orc_expand(const orc_search_model *m, int32_t n_nodes, const double *nodes, const double *cs,
                double *nbr, uint8_t *collide) {
    int32_t P = m->n_prim;
    for (int32_t q = 0; q < n_nodes; q++) {
        double x = nodes[3 * q], y = nodes[3 * q + 1], th = nodes[3 * q + 2];
        double c = cs ? cs[2 * q] : cos(th), s = cs ? cs[2 * q + 1] : sin(th);
        int rot_only = (x == 0.0 && y == 0.0);   /* linalg.py:13-17: 2x2 matrix, no translation */
        double tx = rot_only ? 0.0 : x, ty = rot_only ? 0.0 : y;
        for (int32_t k = 0; k < P; k++) {
            int hit = 0;
            for (int32_t o = 0; o < m->n_obst && !hit; o++) {
                for (int32_t i = m->tmpl_off[k]; i < m->tmpl_off[k + 1] && !hit; i++) {
                    double px = m->tmpl_xy[2 * i], py = m->tmpl_xy[2 * i + 1];
                    double wx = fma(py, -s, px * c) + tx;
                    double wy = fma(py, c, px * s) + ty;
                    int inside = 1;
                    for (int32_t r = m->hp_off[o]; r < m->hp_off[o + 1]; r++) {
                        const double *hp = m->hp + 3 * r;
                        double v = hp[0] * wx + hp[1] * wy + hp[2];
                        if (!(v <= 0.0)) { inside = 0; break; }
                    }
                    if (inside) hit = 1;
                }
            }
            collide[q * P + k] = (uint8_t)hit;
            const double *lp = m->last_pose + 3 * k;
            double *o3 = nbr + ((size_t)q * P + k) * 3;
            o3[0] = fma(lp[0], c, lp[1] * -s) + tx;
            o3[1] = fma(lp[0], s, lp[1] * c) + ty;
            o3[2] = normalize_angle(lp[2] + th);
        }
    }
}

/* ------------------------------------------------------------------ lib/trajectories.py:58-86 */
int32_t orc_resample_curve(const double *pts, int32_t n, int32_t stride, const double *dl_vec, double dl,
                           int32_t keep_last, int32_t *keep) {
    int32_t cnt = 0;
    double cum = 0.0;
    long prev = 0;
    for (int32_t i = 0; i < n; i++) {
        if (i > 0) {
            double dx = pts[i * stride] - pts[(i - 1) * stride], dy = pts[i * stride + 1] - pts[(i - 1) * stride + 1];
            cum += sqrt(dx * dx + dy * dy);
        }
        long q = (long)floor(cum / (dl_vec ? dl_vec[i] : dl));
        int k = (i == 0) || (q - prev >= 1) || (keep_last && i == n - 1);
        prev = q;
        if (k) keep[cnt++] = i;
    }
    return cnt;
}

/* ------------------------------------------------------------------ lib/moving_obstacles_prediction.py:21-47 */
void orc_predict_obstacle(const double *six, double dt, double L, int32_t steps, double *out) {
    double x = six[0], y = six[1], v = six[2], yaw = six[3], a = six[4], st = six[5];
    for (int32_t k = 0; k < steps; k++) {
        x += v * cos(yaw) * dt;
        y += v * sin(yaw) * dt;
        v += a * dt;
        yaw += (v / L) * tan(st) * dt;     /* uses the UPDATED v (order differs from the plant) */
        out[4 * k] = x; out[4 * k + 1] = y; out[4 * k + 2] = yaw; out[4 * k + 3] = k * dt;
    }
}

/* disc centre of pose (x,y,th) : trajectories.py:11-37 */
static void disc_xy(const double *pose, const double *cc, double *o) {
    double c = cos(pose[2]), s = sin(pose[2]);
    o[0] = (c * cc[0] - s * cc[1]) + pose[0];
    o[1] = (s * cc[0] + c * cc[1]) + pose[1];
}

/* ------------------------------------------------------------------ lib/collision_avoidance.py:66-104
 * Row order of the reference's flattened pair table (derived from _get_rowwise_diffs :32-46 and
 * _offset_trajectories_by_frames :49-63): frame f (major), agent disc ca, obstacle o, offset index, obstacle disc co.
 * Shifted obstacle pose at frame f, offset d: traj_o[clamp(f - d, 0, last)]; all trajectories are edge-padded
 * to F = max(na, nsteps) frames. */
int32_t orc_check_collision_moving_cars(const double *centers, int32_t nc, double radius,
                                        const double *ta, int32_t na, const double *path, int32_t np_,
                                        const double *tobs, int32_t nobs, int32_t nsteps, int32_t w, double *hit_xy) {
    if (nobs == 0) return -1;
    double md = 2.0 * radius;
    int32_t F = na > nsteps ? na : nsteps;
    for (int32_t f = 0; f < F; f++) {
        const double *pa = ta + 3 * (f < na ? f : na - 1);
        for (int32_t ca = 0; ca < nc; ca++) {
            double a[2];
            disc_xy(pa, centers + 2 * ca, a);
            for (int32_t o = 0; o < nobs; o++)
                for (int32_t d = -w; d <= w; d++) {
                    int32_t ff = f < nsteps ? f : nsteps - 1;   /* padded obstacle frame */
                    int32_t gidx;
                    /* offset applied on the un-padded 'nsteps'-long trajectory, then padded to F */
                    if (d < 0) { gidx = ff - d; if (gidx > nsteps - 1) gidx = nsteps - 1; }
                    else if (d > 0) { gidx = ff - d; if (gidx < 0) gidx = 0; }
                    else gidx = ff;
                    const double *po = tobs + ((size_t)o * nsteps + gidx) * 4;
                    double pose[3] = {po[0], po[1], po[2]};
                    for (int32_t co = 0; co < nc; co++) {
                        double b[2];
                        disc_xy(pose, centers + 2 * co, b);
                        double dx = a[0] - b[0], dy = a[1] - b[1];
                        if (sqrt(dx * dx + dy * dy) <= md) {
                            /* earliest pose of the detailed path whose disc (front block first) is within md */
                            int32_t first = 0;
                            for (int32_t cc = 0; cc < nc; cc++) {
                                int found = 0;
                                for (int32_t i = 0; i < np_; i++) {
                                    double q[2];
                                    disc_xy(path + 3 * i, centers + 2 * cc, q);
                                    double ex = b[0] - q[0], ey = b[1] - q[1];
                                    if (sqrt(ex * ex + ey * ey) <= md) { first = i; found = 1; break; }
                                }
                                if (found) break;
                            }
                            hit_xy[0] = path[3 * first]; hit_xy[1] = path[3 * first + 1];
                            return first;
                        }
                    }
                }
        }
    }
    return -1;
}

/* ------------------------------------------------------------------ lib/collision_avoidance.py:107-119 */
int32_t orc_cutoff_idx(const double *pts, int32_t n, double x, double y, double radius) {
    for (int32_t i = 0; i < n; i++) {
        double dx = pts[3 * i] - x, dy = pts[3 * i + 1] - y;
        if (sqrt(dx * dx + dy * dy) <= radius) return i;
    }
    return -1;
}
