/*
 * oracle_jerk.c -- CPU restatement of the five-state MPC variant, reference main/lib/mpc_jerk.py.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * What differs from lib/mpc.py (cited per line below): a fifth state that integrates the acceleration input and feeds the
 * speed (mpc_jerk.py:67, 73, 78), a penalty on its change between consecutive stages (line 190), only x[:4, 0] pinned (line
 * 193, so the fifth state's initial value z is one more unknown of the problem), weights 10 / 1 on the cross-track / along-track
 * error (lines 167, 171), Rd = (0.3, 1) and MAX_DECEL = -5 (lines 22, 39).  Reference window, linearisation point
 * (the 4-state plant, lines 112-126) and every inequality are those of lib/mpc.py.
 *
 * Condensed unknowns: w = [a_0, d_0, ..., a_{T-1}, d_{T-1}, z], n = 2T + 1.  The interior-point iteration is the shared
 * orc_ipm_dense (same rules as the 4-state problem), started from the warm inputs and z = 0.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* mpc_jerk.py:62-86 (delta = dref = 0, line 100) */
static void jerk_model(double v, double phi, double dt, double L, double *A /*5x5*/, double *B /*5x2*/, double *C /*5*/) {
    memset(A, 0, sizeof(double) * 25); memset(B, 0, sizeof(double) * 10); memset(C, 0, sizeof(double) * 5);
    for (int i = 0; i < 5; i++) A[i * 5 + i] = 1.0;
    A[0 * 5 + 2] = dt * cos(phi); A[0 * 5 + 3] = -dt * v * sin(phi);
    A[1 * 5 + 2] = dt * sin(phi); A[1 * 5 + 3] = dt * v * cos(phi);
    A[2 * 5 + 4] = dt;
    B[2 * 2 + 0] = dt; B[3 * 2 + 1] = dt * v / L; B[4 * 2 + 0] = dt;
    C[0] = dt * v * sin(phi) * phi; C[1] = -dt * v * cos(phi) * phi;
}

/* mpc_jerk.py:143-199.  xref, xbar: 4 x (T+1) (the reference's fifth rows are zero and carry zero weight: Qf[4] = 0,
 * the linear model does not depend on the fifth operating point).  S: (T+1) x 5 x n, c: (T+1) x 5.  Returns m. */
int32_t orc_qp_build_jerk(const orc_mpc_params *p, const double *x0, const double *xref, const double *xbar,
                          const uint8_t *re, double *H, double *g, double *G, double *h, double *S, double *c) {
    const int32_t T = p->T, W = T + 1, n = 2 * T + 1, NX = 5;
    const int32_t m = 4 * T + 2 * (T - 1) + 2 * T;
    memset(H, 0, sizeof(double) * n * n);
    memset(g, 0, sizeof(double) * n);
    memset(G, 0, sizeof(double) * m * n);
    memset(S, 0, sizeof(double) * W * NX * n);
    for (int i = 0; i < 4; i++) c[i] = x0[i];
    c[4] = 0.0;
    S[4 * n + 2 * T] = 1.0;                         /* x[4, 0] = z */
    for (int32_t t = 0; t < T; t++) {
        double A[25], B[10], C[5];
        jerk_model(xbar[2 * W + t], xbar[3 * W + t], p->dt, p->L, A, B, C);
        const double *St = S + (size_t)t * NX * n;
        double *Sn = S + (size_t)(t + 1) * NX * n;
        for (int i = 0; i < NX; i++) {
            for (int k = 0; k < n; k++) {
                double acc = 0.0;
                for (int j = 0; j < NX; j++) acc += A[i * NX + j] * St[j * n + k];
                Sn[i * n + k] = acc;
            }
            Sn[i * n + 2 * t + 0] += B[i * 2 + 0];
            Sn[i * n + 2 * t + 1] += B[i * 2 + 1];
            double acc = C[i];
            for (int j = 0; j < NX; j++) acc += A[i * NX + j] * c[t * NX + j];
            c[(t + 1) * NX + i] = acc;
        }
    }
    /* state costs, t = 1..T (lines 163-177); no weight on the fifth state in either branch */
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
        const double *St = S + (size_t)t * NX * n;
        double e[4], We[4];
        for (int i = 0; i < 4; i++) e[i] = c[t * NX + i] - xref[i * W + t];
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
    /* input costs (lines 183-186), rate costs (line 189) */
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
    /* jerk term (line 190): w * (x4_{t+1} - x4_t)^2, t = 0..T-2, through the condensed rows of the fifth state */
    for (int32_t t = 0; t + 1 < T; t++) {
        const double *S0 = S + ((size_t)t * NX + 4) * n, *S1 = S + ((size_t)(t + 1) * NX + 4) * n;
        const double e = c[(t + 1) * NX + 4] - c[t * NX + 4];
        for (int a = 0; a < n; a++) {
            const double da = S1[a] - S0[a];
            if (da == 0.0) continue;
            for (int b = 0; b < n; b++) H[a * n + b] += 2.0 * p->jerk_weight * da * (S1[b] - S0[b]);
            g[a] += 2.0 * p->jerk_weight * da * e;
        }
    }
    /* constraints (lines 191, 194-198), same row order as orc_qp_build */
    int32_t r = 0;
    for (int32_t t = 0; t < T; t++) {
        G[(size_t)r * n + 2 * t] = 1.0; h[r++] = p->max_accel;
        G[(size_t)r * n + 2 * t] = -1.0; h[r++] = -p->max_decel;
        G[(size_t)r * n + 2 * t + 1] = 1.0; h[r++] = p->max_steer;
        G[(size_t)r * n + 2 * t + 1] = -1.0; h[r++] = p->max_steer;
    }
    for (int32_t t = 0; t + 1 < T; t++) {
        G[(size_t)r * n + 2 * (t + 1) + 1] = 1.0; G[(size_t)r * n + 2 * t + 1] = -1.0; h[r++] = p->max_dsteer * p->dt;
        G[(size_t)r * n + 2 * (t + 1) + 1] = -1.0; G[(size_t)r * n + 2 * t + 1] = 1.0; h[r++] = p->max_dsteer * p->dt;
    }
    for (int32_t t = 1; t <= T; t++) {
        const double *Sv = S + ((size_t)t * NX + 2) * n;
        for (int k = 0; k < n; k++) G[(size_t)r * n + k] = Sv[k];
        h[r++] = p->max_speed - c[t * NX + 2];
        for (int k = 0; k < n; k++) G[(size_t)r * n + k] = -Sv[k];
        h[r++] = -p->min_speed + c[t * NX + 2];
    }
    return r;
}

/* mpc_jerk.py:143-215; x_out is 5 x (T+1) (the reference returns rows 0..3, line 201-206; row 4 is kept for the KKT checks) */
int32_t orc_qp_solve_jerk(const orc_mpc_params *p, const double *x0, const double *xref, const double *xbar,
                          const uint8_t *re, const double *u_warm, double *x_out, double *u_out, double *lam_out,
                          int32_t *iters, double *kkt4) {
    const int32_t T = p->T, W = T + 1, n = 2 * T + 1, mcap = 8 * T, NX = 5;
    double *H = malloc(sizeof(double) * n * n), *g = malloc(sizeof(double) * n);
    double *G = malloc(sizeof(double) * mcap * n), *h = malloc(sizeof(double) * mcap);
    double *S = malloc(sizeof(double) * W * NX * n), *c = malloc(sizeof(double) * W * NX);
    double *u = calloc(n, sizeof(double));
    int32_t m = orc_qp_build_jerk(p, x0, xref, xbar, re, H, g, G, h, S, c);
    int32_t status;
    if (x0[2] > p->max_speed + 1e-9 || x0[2] < p->min_speed - 1e-9) {
        status = ORC_INFEASIBLE; *iters = 0;
        if (kkt4) kkt4[0] = kkt4[1] = kkt4[2] = kkt4[3] = 0.0;
    } else {
        if (u_warm) for (int t = 0; t < T; t++) { u[2 * t] = u_warm[t]; u[2 * t + 1] = u_warm[T + t]; }
        status = orc_ipm_dense(p, n, m, H, g, G, h, u, lam_out, iters, kkt4);
    }
    for (int t = 0; t < T; t++) { u_out[t] = u[2 * t]; u_out[T + t] = u[2 * t + 1]; }
    for (int t = 0; t <= T; t++)
        for (int i = 0; i < NX; i++) {
            double a = c[t * NX + i];
            for (int k = 0; k < n; k++) a += S[((size_t)t * NX + i) * n + k] * u[k];
            x_out[i * W + t] = a;
        }
    free(H); free(g); free(G); free(h); free(S); free(c); free(u);
    return status;
}
