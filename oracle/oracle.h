/*
 * oracle.h -- CPU restatement (plain C, float64) of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker.  The product path (mpc_for_av_at_intersection_amd/ + libmpcx.so) never
 * links, imports or calls it and fails loudly when the HIP library is missing.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/main).  Parity status: the numpy-only functions are pinned by
 * the npz fixtures under tests/golden (generated from the reference by tests/golden/make_golden.py);
 * the QP solve (lib/mpc.py:148-194, cvxpy -> ECOS, neither installable here) is
 * "parity unpinned" by reference outputs and is certified per instance by KKT
 * residuals of the un-condensed problem plus a scipy cross-check (tests/test_oracle_qp.py).
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_T_MAX 32

/* lib/mpc.py:13-36 + config/mpc_config.json + lib/simulation.py:23-25 */
typedef struct {
    int32_t T;          /* horizon */
    int32_t max_iter;   /* interior-point iteration cap */
    double dt;          /* 0.2 */
    double L;           /* wheelbase, car_dimensions.py:82-107 */
    double w_perp, w_para;
    double R[2], Rd[2], Q_v_yaw[2];
    double Qf[4];       /* already multiplied by T (mpc.py:25) */
    double R_end[2];    /* diag(10,10), mpc.py:178 */
    double max_speed, min_speed, max_accel, max_decel, max_steer, max_dsteer; /* dsteer in rad/s */
    double tol;         /* KKT tolerance of the interior-point loop */
    int32_t model;      /* 0: lib/mpc.py (4 states)   1: lib/mpc_jerk.py (5 states) */
    int32_t reserved;
    double jerk_weight; /* jerk_penalty_weight, mpc_jerk.py:30 (model 1 only) */
} orc_mpc_params;

/* status codes shared with the product library */
enum { ORC_OK = 0, ORC_MAXITER = 1, ORC_INFEASIBLE = 2, ORC_NUMERIC = 3 };

void orc_linear_model(double v, double phi, double delta, double dt, double L,
                      double *A16, double *B8, double *C4);                      /* mpc.py:58-79 */
void orc_xy_cost_mtx(double angle, double *M4);                                  /* mpc.py:129-135 */
void orc_smooth_yaw(double *yaw, int32_t n);                                     /* mpc.py:43-55 */
int32_t orc_nearest_index_in_direction(double x, double y, const double *cx, const double *cy, int32_t n,
                                       int32_t start, int32_t forward);          /* trajectories.py:100-126; -1 = "something wrong" */
int32_t orc_calc_ref_trajectory_ov(const orc_mpc_params *p, const double *state4, const double *cx, const double *cy,
                                   const double *cyaw, const double *cv, const double *ov_prev /* T+1 or NULL */,
                                   int32_t n, double dl, int32_t start_idx, double *xref, uint8_t *reaches_end);
int32_t orc_calc_ref_trajectory(const orc_mpc_params *p, const double *state4 /*x,y,v,yaw*/,
                                const double *cx, const double *cy, const double *cyaw, const double *cv /*or NULL*/, int32_t n, double dl,
                                int32_t start_idx, double *xref /*4,(T+1)*/, uint8_t *reaches_end /*T+1*/); /* mpc.py:86-109; returns new start idx or -1 */
void orc_predict_motion(const orc_mpc_params *p, const double *x0 /*x,y,v,yaw*/, const double *oa, const double *od,
                        double *xbar /*4,(T+1)*/);                               /* mpc.py:112-126, simulation.py:35-47, bicycle/main.py:28-41 */
void orc_plant_step(const orc_mpc_params *p, double *state4 /*x,y,v,yaw in-out*/, double a, double delta); /* simulation.py:35-47 */

/* mpc.py:138-208: dense condensed QP  min 1/2 u'Hu + g'u  s.t. Gu <= h, u interleaved [a0,d0,a1,d1,...] */
int32_t orc_qp_build(const orc_mpc_params *p, const double *x0, const double *xref, const double *xbar,
                     const uint8_t *reaches_end, double *H /*n,n*/, double *g /*n*/, double *G /*m,n*/, double *h /*m*/,
                     double *S /*(T+1),4,n*/, double *c /*(T+1),4*/);            /* returns m */
int32_t orc_qp_solve(const orc_mpc_params *p, const double *x0, const double *xref, const double *xbar,
                     const uint8_t *reaches_end, const double *u_warm /*2,T or NULL*/,
                     double *x_out /*4,(T+1)*/, double *u_out /*2,T*/, double *lam_out /*m or NULL*/,
                     int32_t *iters, double *kkt4 /*stat, prim, comp, gap*/);     /* returns status */

/* the interior-point iteration itself on a dense problem (n unknowns, m rows), from the start w (in-out) */
int32_t orc_ipm_dense(const orc_mpc_params *p, int32_t n, int32_t m, const double *H, const double *g, const double *G,
                      const double *h, double *w, double *lam_out, int32_t *iters, double *kkt4);
/* mpc_jerk.py:143-215 (oracle_jerk.c): unknowns [a0,d0,...,a_{T-1},d_{T-1}, z = x[4,0]], n = 2T+1; S (T+1),5,n ; c (T+1),5 */
int32_t orc_qp_build_jerk(const orc_mpc_params *p, const double *x0, const double *xref, const double *xbar,
                          const uint8_t *reaches_end, double *H, double *g, double *G, double *h, double *S, double *c);
int32_t orc_qp_solve_jerk(const orc_mpc_params *p, const double *x0, const double *xref, const double *xbar,
                          const uint8_t *reaches_end, const double *u_warm, double *x_out /*5,(T+1)*/, double *u_out /*2,T*/,
                          double *lam_out, int32_t *iters, double *kkt4);

/* motion_primitive_search.py:87-121 + obstacles.py:157-176 + linalg.py:4-54 + maths.py:4-10 */
typedef struct {
    int32_t n_prim, n_obst;
    const int32_t *tmpl_off;   /* n_prim+1 offsets into tmpl_xy (points) */
    const double *tmpl_xy;     /* collision template points (x,y) in the primitive frame */
    const double *last_pose;   /* n_prim x 3 : last point of each primitive */
    const double *edge_cost;   /* n_prim : total_length */
    const int32_t *hp_off;     /* n_obst+1 offsets into hp rows */
    const double *hp;          /* rows (a,b,c) */
} orc_search_model;
void orc_expand(const orc_search_model *m, int32_t n_nodes, const double *nodes /*n,3*/, const double *cs /*n,2 cos,sin or NULL*/,
                double *nbr /*n,P,3*/, uint8_t *collide /*n,P*/);

/* trajectories.py:58-86 ; dl_vec may be NULL (then scalar dl) ; returns number kept, indices in keep */
int32_t orc_resample_curve(const double *pts, int32_t n, int32_t stride, const double *dl_vec, double dl, int32_t keep_last, int32_t *keep);
/* moving_obstacles_prediction.py:21-47 */
void orc_predict_obstacle(const double *six /*x,y,v,yaw,a,steer*/, double dt, double L, int32_t steps, double *out /*steps,4: x,y,yaw,t*/);
/* collision_avoidance.py:66-104 ; returns idx on the detailed path or -1 (None) */
int32_t orc_check_collision_moving_cars(const double *centers /*nc,2*/, int32_t nc, double radius,
                                        const double *traj_agent /*na,3*/, int32_t na,
                                        const double *path /*np,3*/, int32_t np_,
                                        const double *traj_obs /*nobs, nsteps, 4*/, int32_t nobs, int32_t nsteps,
                                        int32_t frame_window, double *hit_xy);
/* collision_avoidance.py:107-119 ; -1 when nothing matches */
int32_t orc_cutoff_idx(const double *pts /*n,3*/, int32_t n, double x, double y, double radius);

/* oracle_batch.c: the whole per-agent step (scenarios/mpc_intersection.py:95-159) as one call, and a pthread driver over a batch */
int32_t orc_agent_step(const orc_mpc_params *p, const double *full /*n,3*/, int32_t n, double dl, const double *state4,
                       const double *obs6 /*K,6*/, int32_t K, int32_t traj_idx, int32_t prev_cut, int32_t target_ind,
                       const double *u_warm /*2,T or NULL*/, const double *centers /*2,2*/, double radius, int32_t cutoff_margin,
                       int32_t pred_steps, int32_t frame_window, double max_accel,
                       int32_t *out6 /*traj_idx, cut, target_ind, hit, status, iters*/, double *x_out /*4,T+1*/, double *u_out /*2,T*/);
int32_t orc_agent_steps_mt(int32_t n_threads, const orc_mpc_params *p, int32_t P, int32_t A,
                           const double *path, const int32_t *path_off, const int32_t *path_len, double dl,
                           const double *state /*P,4*/, const double *applied /*P,2*/, const double *u_warm /*P,2,T*/,
                           const int32_t *traj_idx, const int32_t *prev_cut, const int32_t *target_ind,
                           const double *centers, double radius, int32_t cutoff_margin, int32_t pred_steps, int32_t frame_window,
                           double max_accel, int32_t *out6 /*P,6*/, double *x_out /*P,4,T+1*/, double *u_out /*P,2,T*/);

#ifdef __cplusplus
}
#endif
#endif
