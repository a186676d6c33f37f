/*
 * oracle_batch.c -- the per-agent closed-loop step of the oracle as ONE C call, and a pthread driver over many agents.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  orc_agent_step is the body of main/scenarios/mpc_intersection.py:95-159 for one
 * ego, composed from the restated functions of oracle.c exactly as oracle_py.agent_step composes them in Python
 * (tests/test_oracle_golden.py checks the two against each other); orc_agent_steps_mt runs it for every (instance, agent) pair
 * of a captured batch state on n_threads host threads -- bench.py's cpu_baseline at all host cores, with no Python in the loop.
 */
#include "oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* out6 = (traj_idx, cut, target_ind, hit index or -1, QP status, QP iterations); returns 0, or -1 for the reference's
 * Exception("something wrong") (trajectories.py:120) */
int32_t orc_agent_step(const orc_mpc_params *p, const double *full /*n,3*/, int32_t n, double dl, const double *state4,
                       const double *obs6 /*K,6*/, int32_t K, int32_t traj_idx, int32_t prev_cut, int32_t target_ind,
                       const double *u_warm /*2,T or NULL*/, const double *centers /*2,2*/, double radius, int32_t cutoff_margin,
                       int32_t pred_steps, int32_t frame_window, double max_accel,
                       int32_t *out6, double *x_out /*4,T+1*/, double *u_out /*2,T*/) {
    const int32_t T = p->T, W = T + 1;
    const double x = state4[0], y = state4[1], v = state4[2];
    double *cx = malloc(sizeof(double) * n), *cy = malloc(sizeof(double) * n), *cyaw = malloc(sizeof(double) * n);
    double *rdl = malloc(sizeof(double) * n), *tres = malloc(sizeof(double) * 3 * n);
    int32_t *keep = malloc(sizeof(int32_t) * n);
    double *trajs = malloc(sizeof(double) * (size_t)(K > 0 ? K : 1) * pred_steps * 4);
    double xref[4 * (ORC_T_MAX + 1)], xbar[4 * (ORC_T_MAX + 1)], uw[2 * ORC_T_MAX], kkt[4];
    uint8_t re[ORC_T_MAX + 1];
    int32_t rc = 0, hit = -1, cut = n, iters = 0, status = ORC_MAXITER;
    for (int32_t i = 0; i < n; i++) { cx[i] = full[3 * i]; cy[i] = full[3 * i + 1]; cyaw[i] = full[3 * i + 2]; }
    /* :103-105 advance traj_agent_idx unless tmp_trajectory collapsed onto it */
    int advance = 1;
    if (prev_cut > 0) {
        const double *a = full + 3 * (size_t)traj_idx, *b = full + 3 * (size_t)(prev_cut - 1);
        advance = (a[0] != b[0]) || (a[1] != b[1]) || (a[2] != b[2]);
    }
    if (advance) {
        traj_idx = orc_nearest_index_in_direction(x, y, cx, cy, n, traj_idx, 1);
        if (traj_idx < 0) { rc = -1; goto done; }
    }
    {
        const double *traj = full + 3 * (size_t)traj_idx;
        const int32_t nt = n - traj_idx;
        int32_t na;
        /* :110-116 ego prediction */
        if (v < p->max_speed) {
            double c = 0.0;
            for (int32_t i = 0; i < nt; i++) { c += max_accel; rdl[i] = p->dt * fmin(c + v, p->max_speed); }
            na = orc_resample_curve(traj, nt, 3, rdl, 0.0, 1, keep);
        } else {
            na = orc_resample_curve(traj, nt, 3, NULL, p->dt * p->max_speed, 1, keep);
        }
        for (int32_t i = 0; i < na; i++) memcpy(tres + 3 * (size_t)i, traj + 3 * (size_t)keep[i], 3 * sizeof(double));
        /* :119-122 predictions of the others, :125-136 conflict + cut */
        for (int32_t k = 0; k < K; k++) orc_predict_obstacle(obs6 + 6 * (size_t)k, p->dt, p->L, pred_steps, trajs + (size_t)k * pred_steps * 4);
        double hxy[2] = {0.0, 0.0};
        if (K > 0) hit = orc_check_collision_moving_cars(centers, 2, radius, tres, na, traj, nt, trajs, K, pred_steps, frame_window, hxy);
        if (hit >= 0) {
            cut = orc_cutoff_idx(full, n, hxy[0], hxy[1], 0.001) - cutoff_margin;
            if (cut < traj_idx + 1) cut = traj_idx + 1;
        }
    }
    /* mpc.py:211-239 on tmp_trajectory = full[:cut] */
    target_ind = orc_calc_ref_trajectory(p, state4, cx, cy, cyaw, NULL, cut, dl, target_ind, xref, re);
    if (target_ind < 0) { rc = -1; goto done; }
    if (u_warm) memcpy(uw, u_warm, sizeof(double) * 2 * T); else memset(uw, 0, sizeof(double) * 2 * T);
    orc_predict_motion(p, state4, uw, uw + T, xbar);
    if (p->model == 1) {            /* lib/mpc_jerk.py: five state rows come back, the reference keeps the first four (mpc_jerk.py:201-206) */
        double x5[5 * (ORC_T_MAX + 1)];
        status = orc_qp_solve_jerk(p, state4, xref, xbar, re, uw, x5, u_out, NULL, &iters, kkt);
        memcpy(x_out, x5, sizeof(double) * 4 * W);
    } else {
        status = orc_qp_solve(p, state4, xref, xbar, re, uw, x_out, u_out, NULL, &iters, kkt);
    }
done:
    out6[0] = traj_idx; out6[1] = cut; out6[2] = target_ind; out6[3] = hit; out6[4] = status; out6[5] = iters;
    free(cx); free(cy); free(cyaw); free(rdl); free(tres); free(keep); free(trajs);
    return rc;
}

typedef struct {
    const orc_mpc_params *p;
    int32_t P, A, next;                 /* next: shared work counter */
    const double *path; const int32_t *path_off, *path_len; double dl;
    const double *state, *applied, *u_warm;
    const int32_t *traj_idx, *prev_cut, *target_ind;
    const double *centers; double radius; int32_t cutoff_margin, pred_steps, frame_window; double max_accel;
    int32_t *out6; double *x_out, *u_out;
} batch_job;

static void *batch_worker(void *arg) {
    batch_job *j = arg;
    const int32_t T = j->p->T, A = j->A;
    double obs6[6 * 64];
    for (;;) {
        const int32_t q = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if (q >= j->P) break;
        /* every other agent of the instance is a moving obstacle: (x, y, v, yaw, a, steer) as MovingObstacle*.get() returns them */
        const int32_t b = q / A;
        int32_t K = 0;
        for (int32_t o = b * A; o < (b + 1) * A && K < 64; o++) {
            if (o == q) continue;
            const double *s = j->state + 4 * (size_t)o;
            double *d = obs6 + 6 * K++;
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3]; d[4] = j->applied[2 * o + 1]; d[5] = j->applied[2 * o];
        }
        (void)orc_agent_step(j->p, j->path + 3 * (size_t)j->path_off[q], j->path_len[q], j->dl, j->state + 4 * (size_t)q, obs6, K,
                             j->traj_idx[q], j->prev_cut[q], j->target_ind[q], j->u_warm + (size_t)q * 2 * T, j->centers, j->radius,
                             j->cutoff_margin, j->pred_steps, j->frame_window, j->max_accel,
                             j->out6 + 6 * (size_t)q, j->x_out + (size_t)q * 4 * (T + 1), j->u_out + (size_t)q * 2 * T);
    }
    return NULL;
}

/* P = n_instances * A agents (agent q belongs to instance q / A); applied = (steer, accel) of the last step per agent */
int32_t orc_agent_steps_mt(int32_t n_threads, const orc_mpc_params *p, int32_t P, int32_t A,
                           const double *path, const int32_t *path_off, const int32_t *path_len, double dl,
                           const double *state /*P,4*/, const double *applied /*P,2*/, const double *u_warm /*P,2,T*/,
                           const int32_t *traj_idx, const int32_t *prev_cut, const int32_t *target_ind,
                           const double *centers, double radius, int32_t cutoff_margin, int32_t pred_steps, int32_t frame_window,
                           double max_accel, int32_t *out6 /*P,6*/, double *x_out /*P,4,T+1*/, double *u_out /*P,2,T*/) {
    if (n_threads < 1 || A < 1 || A > 65 || P < 0) return -1;
    batch_job j = {p, P, A, 0, path, path_off, path_len, dl, state, applied, u_warm, traj_idx, prev_cut, target_ind,
                   centers, radius, cutoff_margin, pred_steps, frame_window, max_accel, out6, x_out, u_out};
    pthread_t *th = malloc(sizeof(pthread_t) * n_threads);
    int started = 0;
    for (int i = 1; i < n_threads; i++) if (pthread_create(&th[started], NULL, batch_worker, &j) == 0) started++;
    batch_worker(&j);
    for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    free(th);
    return started + 1;         /* threads that actually ran */
}
