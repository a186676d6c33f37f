/*
 * mpcx.h -- C ABI of libmpcx.so, the MI355X (gfx950) implementation of the per-timestep hot path of
 * SaeedRahmani/MPC_for_AV_at_Intersection.  The reference has no FFI of its own (pure Python); each entry
 * point below names the reference function(s) it replaces (paths relative to /root/reference/main) -- the
 * Python objects in mpc_for_av_at_intersection_amd/ present the reference's call surface on top of this ABI,
 * and INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every function returns int32 status: 0 = OK, <0 = error
 *     (MPCX_E_*; text via mpcx_last_error()).  Solver non-convergence is per-instance DATA (status[] array),
 *     never an error code -- mirroring lib/mpc.py:196-206.
 *   - every array argument is a DEVICE pointer (HIP global memory, e.g. torch tensor.data_ptr()), float64 /
 *     int32 / uint8, C-contiguous, batch as the slowest index.  Caller owns every buffer.
 *   - launches go to the HIP stream given at mpcx_create (NULL = default stream); calls are asynchronous
 *     with respect to the host exactly like a kernel launch, the caller synchronises the stream.
 *   - one mpcx_ctx per host thread / stream; a ctx is not re-entrant.
 */
#ifndef MPCX_H
#define MPCX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCX_T_MAX 32            /* horizon capacity (2T lanes of one wavefront) */
#define MPCX_MAX_PRIM 16         /* motion primitives per search model */
#define MPCX_MAX_OBS 16          /* moving obstacles seen by one ego */
#define MPCX_PRED_STEPS_MAX 64   /* prediction horizon frames */
#define MPCX_MAX_REMAINING 1024  /* path points ahead of an agent mpcx_interaction_batch handles by default ... */
#define MPCX_MAX_PATH_LEN 4096   /* ... and at most, when mpcx_interaction_params.max_path_len asks for more */
#define MPCX_EGO_FRAMES_MAX 128  /* resampled ego poses: mpcx_moving_collision_batch; mpcx_interaction_batch handles capacity/4 - 32 */

enum {
    MPCX_OK = 0,
    MPCX_E_INVALID = -1,   /* bad argument (null pointer, size out of range, unsupported horizon) */
    MPCX_E_LAUNCH = -2,    /* HIP launch / runtime failure */
    MPCX_E_NODEVICE = -3   /* no usable GPU */
};

/* per-instance solver status (status[] outputs) */
enum { MPCX_QP_OPTIMAL = 0, MPCX_QP_MAXITER = 1, MPCX_QP_INFEASIBLE = 2, MPCX_QP_NUMERIC = 3 };

typedef struct mpcx_ctx mpcx_ctx;

/* lib/mpc.py:13-36 (config/mpc_config.json) + lib/simulation.py:23-25 + car_dimensions.py:82-107 */
typedef struct {
    int32_t T;          /* horizon, 1..MPCX_T_MAX */
    int32_t max_iter;   /* interior-point iteration cap */
    double dt, L;
    double w_perp, w_para;
    double R[2], Rd[2], Q_v_yaw[2];
    double Qf[4];       /* ALREADY multiplied by T (mpc.py:25) */
    double R_end[2];    /* diag(10,10) (mpc.py:178) */
    double max_speed, min_speed, max_accel, max_decel, max_steer, max_dsteer /* rad/s */;
    double tol;         /* KKT tolerance (relative) */
    int32_t model;      /* MPCX_MODEL_BICYCLE4 (lib/mpc.py) or MPCX_MODEL_JERK5 (lib/mpc_jerk.py:143-208: a fifth state integrates
                         * the acceleration input, its initial value is free; solved by the stage-structured solver whatever the batch) */
    int32_t reserved;
    double jerk_weight; /* jerk_penalty_weight, mpc_jerk.py:30 (MPCX_MODEL_JERK5 only) */
} mpcx_mpc_params;
enum { MPCX_MODEL_BICYCLE4 = 0, MPCX_MODEL_JERK5 = 1 };

mpcx_ctx *mpcx_create(int32_t device, void *hip_stream);
void mpcx_destroy(mpcx_ctx *ctx);
const char *mpcx_last_error(mpcx_ctx *ctx);
const char *mpcx_version(void);
int32_t mpcx_set_mpc_params(mpcx_ctx *ctx, const mpcx_mpc_params *p);

/* ---- lib/mpc.py:138-208 `_linear_mpc_control` (incl. :58-79 `_get_linear_model_matrix`, :129-135): the QP.
 * Solves B independent problems.  Outputs laid out as the reference's x.value / u.value:
 * x_out[b,0..3,:] = (x, y, v, yaw), u_out[b,0,:] = accel, u_out[b,1,:] = steer.  u_warm may be NULL (zeros).
 * status[b] != 0  <=>  the reference's "Cannot solve mpc" branch (outputs then hold the last iterate).
 * kkt[b,:] = (stationarity inf-norm, primal residual inf-norm, mean complementarity, unused). */
int32_t mpcx_qp_solve_batch(mpcx_ctx *ctx, int32_t B,
                            const double *x0 /*B,4: x,y,v,yaw*/, const double *xref /*B,4,T+1*/,
                            const double *xbar /*B,4,T+1*/, const uint8_t *reaches_end /*B,T+1*/,
                            const double *u_warm /*B,2,T or NULL*/,
                            double *x_out /*B,4,T+1*/, double *u_out /*B,2,T*/,
                            int32_t *status /*B*/, int32_t *iters /*B*/, double *kkt /*B,4*/);

/* ---- lib/mpc.py:86-109 `_calc_ref_trajectory` (+ trajectories.py:100-126) and :112-126 `_predict_motion`
 * (+ simulation.py:35-47, bicycle/main.py:28-41).  Paths are ragged: instance b tracks points
 * [path_off[b], path_off[b]+path_len[b]) of path_xyyaw (rows x,y,yaw; yaw already smooth_yaw'ed);
 * path_len[b] is the CURRENT (possibly cut) length, as after MPC.set_trajectory_fromarray.
 * target_ind is in-out (mpc.py:226).  target_ind[b] = -1 on the reference's Exception("something wrong").
 * path_v (NULL for lib/mpc.py) is the speed profile `cv` of lib/mpc_with_speed.py:85-108: xref[2,:] = cv[idx]. */
int32_t mpcx_mpc_prepare_batch(mpcx_ctx *ctx, int32_t B, const double *state /*B,4: x,y,v,yaw*/,
                               const double *u_warm /*B,2,T or NULL*/,
                               const double *path_xyyaw /*npts,3*/, const double *path_v /*npts or NULL*/,
                               const int32_t *path_off /*B*/,
                               const int32_t *path_len /*B*/, double dl, int32_t *target_ind /*B in-out*/,
                               double *xref /*B,4,T+1*/, uint8_t *reaches_end /*B,T+1*/, double *xbar /*B,4,T+1*/);
/* the same for the second and later of MAX_ITER linearisation passes (lib/mpc.py:226-237 `_iterative_linear_mpc_control`): `ov`, the
 * speeds of the previous pass's solution (row 2 of its x, T+1 values per instance, instance b at ov + b * ov_stride), spaces the
 * reference window (mpc.py:95-98 with ov given) and u_warm = the previous pass's inputs makes the rollout (mpc.py:231).  ov = NULL is
 * mpcx_mpc_prepare_batch. */
int32_t mpcx_mpc_prepare_batch_ov(mpcx_ctx *ctx, int32_t B, const double *state, const double *u_warm,
                                  const double *path_xyyaw, const double *path_v, const int32_t *path_off, const int32_t *path_len,
                                  double dl, int32_t *target_ind, const double *ov /*or NULL*/, int64_t ov_stride,
                                  double *xref, uint8_t *reaches_end, double *xbar);
/* lib/mpc.py:226 `for _ in range(MAX_ITER)` inside mpcx_closed_loop_run: passes >= 1 (default 1 = the stock mpc_config.json) */
int32_t mpcx_set_linearisation_passes(mpcx_ctx *ctx, int32_t passes);

/* ---- lib/motion_primitive_search.py:87-121 `neighbor_function` (+ obstacles.py:157-176 `check_collision`,
 * linalg.py:4-54, maths.py:4-10).  Model tables are copied to the device once by mpcx_search_model_create
 * (HOST pointers there).  primitive id = index in the arrays given (callers use sorted names).
 * A* nodes are compared by exact float equality and ordered by exact f-values (a_star.py:34-49): a search that must
 * replay the reference's pop order passes nodes_cs computed by the host's numpy (what linalg.py:4-22 calls), which
 * makes successor coordinates bit-identical to the reference; bulk expansion passes NULL. */
typedef struct mpcx_search_model mpcx_search_model;
mpcx_search_model *mpcx_search_model_create(mpcx_ctx *ctx, int32_t n_prim,
                                            const int32_t *tmpl_off /*n_prim+1*/, const double *tmpl_xy /*npts,2*/,
                                            const double *last_pose /*n_prim,3*/, const double *edge_cost /*n_prim*/,
                                            int32_t n_obst, const int32_t *hp_off /*n_obst+1*/, const double *hp /*rows,3*/);
void mpcx_search_model_destroy(mpcx_search_model *m);
int32_t mpcx_expand_batch(mpcx_ctx *ctx, const mpcx_search_model *m, int32_t n_nodes, const double *nodes /*n,3*/,
                          const double *nodes_cs /*n,2: cos,sin of nodes[:,2], or NULL = device sincos*/,
                          double *nbr /*n,P,3*/, double *cost /*n,P*/, uint8_t *collide /*n,P*/);

/* ---- lib/a_star.py:31-78 `AStar.run` + lib/motion_primitive_search.py:64-75,87-121 (is_goal, distance_to_goal, neighbor_function),
 * lib/motion_primitive_search_modified.py:80-89, lib/motion_primitive_search_multi_lane.py:56-108,155-181,226-237,
 * lib/motion_primitive_search_roundabout.py:131-157,212 and lib/motion_primitive_search_single_lane.py:145-162,218 for MANY independent
 * searches, open list and closed set resident on the device: one wavefront per search, no host work between expansions.  The pop order
 * is the reference's (tuples (g + h, g, node, predecessor) compared field by field, node identity = float equality), which needs the
 * reference's bits in every number: cos / sin of a node's heading come from a table the host fills with numpy (cs_theta ascending,
 * cs_val (cos, sin)): a heading missing from it ends the search with MPCX_ASTAR_MISS and the heading in buffers.miss.  Heuristic and
 * edge values are evaluated with un-fused IEEE operations; where the reference goes through libm pow (Python's x ** 2: one ulp from
 * x * x for ~0.08 % of arguments) or BLAS (np.linalg.norm) the device value can differ by an ulp, so the kernel LOGS every value it
 * used (successor log below), the host re-evaluates them with the reference's own expressions and hands the few that differ back in a
 * PER-SEARCH override table: rows (x, y, theta, kind) sorted as tuples inside each search's slice [ov_off, ov_off + ov_cnt) of
 * ov_key / ov_val; kind = -1: ov_val is h(node); kind = k >= 0: ov_val is the edge value of primitive k leaving `node`.  Overrides are
 * never consulted for MPCX_ASTAR_BASE (its arithmetic is exact on the device).  models / searches are HOST arrays; every pointer inside
 * mpcx_astar_buffers is a DEVICE pointer to caller-owned memory: heap n x heap_cap x 10 doubles, table n x table_cap x 8 doubles
 * FILLED WITH NaN (table_cap a power of two), log n x log_cap x 8 (node, g, h, predecessor per expansion: a_star.py:52), push_log
 * n x push_cap x 8 -- the successor log: (node, h, edge value, index of the expansion it came from, primitive id, g) per PUSH for
 * BASE / MODIFIED and per FREE successor for the other variants (h = NaN if the successor was not pushed) --, path n x path_cap x 3 and
 * path_prim n x path_cap (goal first, primitive that led to each node, -1 at the start), cost / miss / status / n_exp / n_push /
 * path_len n each.  A path longer than path_cap ends in MPCX_ASTAR_PATH_CAPACITY (nothing is truncated silently). */
enum { MPCX_ASTAR_BASE = 0, MPCX_ASTAR_MODIFIED = 1, MPCX_ASTAR_MULTI_LANE = 2, MPCX_ASTAR_ROUNDABOUT = 3, MPCX_ASTAR_SINGLE_LANE = 4 };
enum { MPCX_ASTAR_FOUND = 0, MPCX_ASTAR_EXHAUSTED = 1 /* "No solution found." */, MPCX_ASTAR_CAPACITY = 2, MPCX_ASTAR_MISS = 3,
       MPCX_ASTAR_PATH_CAPACITY = 4 };
typedef struct {
    double start[3];
    double goal_box[4];         /* BoxObstacle.xy1, xy2 of scenario.goal_area */
    double goal_point[3];
    double allowed_dtheta;      /* scenario.allowed_goal_theta_difference */
    double wh[5];               /* MULTI_LANE: wh_dist, wh_theta, wh_steering, wh_obstacle, wh_center (_multi_lane.py:24) */
    double wc[4];               /* MULTI_LANE: wc_dist, wc_steering, wc_obstacle, wc_center (_multi_lane.py:26) */
    const double *hp_norm;      /* DEVICE, one per half-plane row of the model: (a**2 + b**2)**0.5 as the host's Python evaluates it
                                 * (_multi_lane.py:95); needed by ROUNDABOUT / SINGLE_LANE and by MULTI_LANE with wh_obstacle != 0 */
    int32_t variant;            /* MPCX_ASTAR_BASE ... MPCX_ASTAR_SINGLE_LANE */
    int32_t max_expansions;
    int32_t ov_off, ov_cnt;     /* this search's slice of the override table */
} mpcx_astar_search;
typedef struct {
    int32_t heap_cap, table_cap, log_cap, push_cap, path_cap;
    double *heap, *table, *log, *push_log, *path, *cost, *miss;
    int32_t *status, *n_exp, *n_push, *path_len, *path_prim;
} mpcx_astar_buffers;
int32_t mpcx_astar_batch(mpcx_ctx *ctx, int32_t n_search, const mpcx_search_model *const *models, const mpcx_astar_search *searches,
                         int32_t n_cs, const double *cs_theta, const double *cs_val,
                         int32_t n_ov, const double *ov_key /*n_ov,4*/, const double *ov_val /*n_ov*/, const mpcx_astar_buffers *buffers);

/* ---- the same expansion for SEVERAL searches in one launch (many independent planners running concurrently): segment s = nodes
 * seg_off[s] .. seg_off[s+1]-1 of the node table (HOST array, n_seg+1 entries), expanded against models[s] (HOST array of
 * handles; all with the same number of primitives).  Outputs are laid out exactly as n_seg separate mpcx_expand_batch calls
 * on the segments would lay them out. */
int32_t mpcx_expand_multi_batch(mpcx_ctx *ctx, int32_t n_seg, const mpcx_search_model *const *models, const int32_t *seg_off,
                                const double *nodes /*n,3*/, const double *nodes_cs /*n,2 or NULL*/,
                                double *nbr /*n,P,3*/, double *cost /*n,P*/, uint8_t *collide /*n,P*/);

/* ---- lib/collision_avoidance.py:66-119 `check_collision_moving_cars` + `get_cutoff_curve_by_position_idx`,
 * lib/moving_obstacles_prediction.py:21-47, trajectories.py:58-86 `resample_curve`, and the caller sequence
 * scenarios/mpc_intersection.py:103-136.  P independent problems (one ego each).  Moving obstacles live in a
 * pool obs6[NOBS,6] of 6-tuples (x, y, v, yaw, a, steer) -- what MovingObstacle*.get() returns,
 * mpc_intersection.py:119-122; each is predicted once.  Problem p sees obstacles
 * obs_off[p] .. obs_off[p]+obs_cnt[p]-1 of the pool except index obs_skip[p] (-1 = none): an N-agent instance
 * puts its N agents in the pool and every agent skips itself.
 * path_cs holds cos/sin of the path yaw column (the host computes them once per path with the same libm the
 * reference uses, so disc centres match trajectories.py:11-37 bit for bit).
 * Outputs: traj_idx (in-out, the scenario's traj_agent_idx), hit_idx (-1 = None, else index on the remaining
 * path; -2 = limits exceeded: more path points ahead / resampled poses than the call's capacity (max_path_len) or more than
 * MPCX_MAX_OBS obstacles -- the agent's path is then left uncut, callers must treat it as an error (batch.check()); -3 = the reference's Exception("something wrong")), hit_xy, cut_len (length of the tmp_trajectory
 * handed to MPC.set_trajectory_fromarray). */
typedef struct {
    int32_t pred_steps;      /* len(arange(0, TIME_HORIZON, DT)) = 35 */
    int32_t frame_window;    /* 20 */
    int32_t cutoff_margin;   /* EXTRA_CUTOFF_MARGIN = 4*ceil(radius/dl) */
    int32_t max_path_len;    /* mpcx_interaction_batch: longest path of the call in points (sizes the kernel's LDS: state it exactly,
                              * resident wavefronts hide the kernel's latency); 0 = MPCX_MAX_REMAINING; at least 512, at most
                              * MPCX_MAX_PATH_LEN.  Resampled ego poses handled: capacity / 4 - 32 */
    double dt, L, radius;
    double circle_centers[4]; /* (x,y) of the 2 discs, car_dimensions.py:61-79 */
    double max_accel, max_speed;
    /* optional (NULL = off): per point of the path table, the arc length from the first point of ITS path (any running sum whose
     * differences inside one path are arc lengths will do), and a bound on how far such a difference can be from the reference's own
     * np.cumsum over the same steps (trajectories.py:72-79; a few n * 2^-53 * length: the host knows n and the length).  Paths are
     * constants of a run while every step of every agent re-derives step lengths and their running sum from the points: with the table
     * mpcx_interaction_batch takes floor(c_i / dl_i) from differences of its entries wherever c_i / dl_i is farther from an integer than
     * that bound (+ rounding) can move it, and falls back to the sequential sum over the points for an agent with a closer call --
     * identical outputs either way. */
    const double *path_cum;
    double path_cum_err;
    /* optional (NULL = off): per point k of the path table, the index (relative to the first point of ITS path) of the first point j <= k of
     * that path with sqrt(dx*dx + dy*dy) <= 0.001 from point k -- get_cutoff_curve_by_position_idx (collision_avoidance.py:107-119) asked
     * for the position of path point k, which is the only way mpc_intersection.py:125-131 ever asks it (collision_xy IS a path point).  A
     * property of the path alone (k itself unless the path has duplicate points), evaluated once on the host with the reference's own
     * expression; with it mpcx_interaction_batch looks the cut index up instead of scanning the path up to the conflict. */
    const int32_t *path_first_within;
    /* optional (plan_cnt = NULL: off; needs path_cum): the EGO PREDICTION of mpc_intersection.py:107-116 per path point.  Once the predicted
     * speed v + MAX_ACCEL (i + 1) has reached MAX_SPEED -- after four points from standstill with the stock constants -- resample_curve's dl
     * is the constant DT * MAX_SPEED, and as long as the few points before that stay in bucket 0 (checked per agent and step) the poses it
     * keeps from trajectory_full[t:] depend on t alone: row t of the tables holds their number (plan_cnt[t]; 0 = not tabulated), the disc
     * centres of the kept poses (plan_disc[t][plan_cap][4]: trajectories.py:11-37) and the boxes of the eight runs of frames the conflict
     * search culls with (plan_box[t][8][4] = xlo, xhi, ylo, yhi, inflated), all computed on the host with the reference's own numpy
     * expressions for plan_dl = DT * MAX_SPEED, plan_steps = pred_steps and the car's discs / radius.  path_disc[npts][4]: the disc centres
     * of every path point (the earliest-pose scan of collision_avoidance.py:88-104 reads them instead of rebuilding them).  An agent
     * whose step does not meet the condition takes the resampling pass as before: identical outputs either way. */
    const int32_t *plan_cnt;
    const double *plan_disc, *plan_box, *path_disc;
    int32_t plan_cap, plan_steps;
    double plan_dl, plan_radius;
} mpcx_interaction_params;
int32_t mpcx_interaction_batch(mpcx_ctx *ctx, const mpcx_interaction_params *ip, int32_t P,
                               const double *state /*P,4*/,
                               const double *path_xyyaw /*npts,3*/, const double *path_cs /*npts,2: cos(yaw),sin(yaw)*/,
                               const int32_t *path_off /*P*/, const int32_t *path_len /*P (full length)*/,
                               const int32_t *prev_cut_len /*P or NULL*/,
                               int32_t n_obs_pool, const double *obs6 /*NOBS,6*/, const int32_t *obs_off /*P*/,
                               const int32_t *obs_cnt /*P*/, const int32_t *obs_skip /*P or NULL*/,
                               int32_t *traj_idx /*P in-out*/, int32_t *hit_idx /*P*/, double *hit_xy /*P,2*/,
                               int32_t *cut_len /*P*/);

/* ---- lib/collision_avoidance.py:66-104 `check_collision_moving_cars` with the reference's own argument meaning:
 * problem p has an already-resampled ego trajectory ego_xyyaw[ego_off[p] .. +ego_len[p]) (traj_agent), a detailed
 * path (path_agent_detailed) and obs_cnt[p] already-predicted obstacle trajectories of exactly ip->pred_steps
 * poses each, stored as pool entries obs_off[p].. (traj_obstacles; only x, y, yaw are used by the reference).
 * *_cs = cos/sin of the yaw column, computed by the host.  hit_idx[p] = -1 for None, else the index on the
 * detailed path (the third element of the reference's return tuple), hit_xy = (x, y). -2 = limits exceeded. */
int32_t mpcx_moving_collision_batch(mpcx_ctx *ctx, const mpcx_interaction_params *ip, int32_t P,
                                    const double *ego_xyyaw /*ne,3*/, const double *ego_cs /*ne,2*/,
                                    const int32_t *ego_off /*P*/, const int32_t *ego_len /*P*/,
                                    const double *path_xyyaw /*np,3*/, const double *path_cs /*np,2*/,
                                    const int32_t *path_off /*P*/, const int32_t *path_len /*P*/,
                                    int32_t n_obs_pool, const double *obs_xyyaw /*NOBS,steps,3*/, const double *obs_cs /*NOBS,steps,2*/,
                                    const int32_t *obs_off /*P*/, const int32_t *obs_cnt /*P*/,
                                    int32_t *hit_idx /*P*/, double *hit_xy /*P,2*/);

/* ---- lib/linalg.py:4-54 `create_2d_transform_mtx` + `transform_2d_pts` for n_items (pose, point-set) pairs:
 * item i maps points pts[pts_off[i] .. +pts_cnt[i]) (rows x, y, theta) to world space at nodes[i]; used for
 * motion_primitive_at / collision_checking_points_at / path_to_full_trajectory
 * (motion_primitive_search.py:77-85,123-135).  out rows beyond pts_cnt[i] are zero. */
int32_t mpcx_transform_batch(mpcx_ctx *ctx, int32_t n_items, int32_t max_pts, const double *nodes /*n,3*/,
                             const int32_t *pts_off /*n*/, const int32_t *pts_cnt /*n*/, const double *pts /*npts,3*/,
                             double *out /*n,max_pts,3*/);

/* ---- lib/collision_avoidance.py:107-119 `get_cutoff_curve_by_position_idx`: first index of each point list within
 * `radius` of xy[p]; -1 where the reference would return the input array ("no cutoff"). */
int32_t mpcx_cutoff_index_batch(mpcx_ctx *ctx, int32_t P, const double *pts /*npts,3*/, const int32_t *off /*P*/,
                                const int32_t *len /*P*/, const double *xy /*P,2*/, double radius, int32_t *out /*P*/);

/* ---- lib/moving_obstacles_prediction.py:21-47 `state_prediction`: poses (x, y, yaw) of `steps` Euler steps. */
int32_t mpcx_predict_obstacles_batch(mpcx_ctx *ctx, int32_t n, int32_t steps, double dt, double L,
                                     const double *obs6 /*n,6*/, double *out_xyyaw /*n,steps,3*/);

/* ---- self-test of the wave-level DPP helpers the kernels rely on (scans, shifts, reductions, reciprocal):
 * in64 = 64 doubles, out322 = results, layout documented at selftest_kernel in csrc/mpcx_misc.hip. */
int32_t mpcx_selftest_wave_ops(mpcx_ctx *ctx, const double *in64, double *out322);
/* f64 MFMA lane maps (v_mfma_f64_16x16x4) used by the Hessian build: D(16x16) = A(16x4, row-major) * B(4x16, row-major) */
int32_t mpcx_selftest_mfma(mpcx_ctx *ctx, const double *A64, const double *B64, double *D256);

/* ---- plant: lib/simulation.py:35-47 `Simulation.step` on B states with the first control of each solution;
 * failed instances (status != 0) get (previous steer, MAX_DECEL) as MPC.step does (mpc.py:294-297) and their row of
 * u is zeroed, which is the warm-start reset of mpc.py:222-224 for the next step. */
int32_t mpcx_plant_step_batch(mpcx_ctx *ctx, int32_t B, double *state /*B,4 in-out*/, double *u /*B,2,T in-out*/,
                              const int32_t *status /*B or NULL*/, double *applied /*B,2 in-out: (steer, accel)*/);

/* ---- the closed loop itself: scenarios/mpc_intersection.py:95-159 for P agents, n_steps times, with no host work
 * between steps.  One step = [pool row q <- (x, y, v, yaw, accel, steer) of agent q, what MovingObstacle*.get()
 * returns] -> mpcx_interaction_batch (prev_cut_len = the cut_len of the previous step; zero = none yet) ->
 * mpcx_mpc_prepare_batch (path_len = cut_len, warm start = u_sol) -> mpcx_qp_solve_batch (warm start = u_sol, in
 * place) -> mpcx_plant_step_batch.  Every agent is a moving obstacle for the agents whose obs_off/obs_cnt window
 * covers it, so the pool has exactly P rows.  All pointers are device pointers owned by the caller; the buffers
 * are the same ones the per-stage entry points take and hold the same values afterwards.
 * use_graph != 0 captures one step into a hipGraph on the context's stream (which must then not be the null
 * stream) and replays it n_steps times; the instantiated graph is cached in the context per descriptor. */
typedef struct {
    int32_t P;
    int32_t exchange;   /* 0: the pool obs6 is this rank's own P agents; MPCX_SHARD_AGENTS: agent-sharded multi-GPU layout, see below */
    double dl;
    double *state /*P,4*/, *applied /*P,2: (steer, accel)*/, *obs6 /*P,6 scratch*/;
    const double *path_xyyaw, *path_cs, *path_v /*or NULL*/;
    const int32_t *path_off /*P*/, *path_len /*P*/, *obs_off /*P*/, *obs_cnt /*P*/, *obs_skip /*P or NULL*/;
    int32_t *traj_idx /*P*/, *target_ind /*P*/, *hit_idx /*P*/, *cut_len /*P, zero-initialised*/;
    double *hit_xy /*P,2*/, *xref /*P,4,T+1*/, *xbar /*P,4,T+1*/;
    uint8_t *reaches_end /*P,T+1*/;
    double *x_sol /*P,4,T+1*/, *u_sol /*P,2,T zero-initialised*/;
    int32_t *status /*P*/, *iters /*P*/;
    double *kkt /*P,4*/;
    /* exchange == MPCX_SHARD_AGENTS only: this rank owns agents_local agents of each of n_inst instances (P = n_inst *
     * agents_local, agent order (instance, local agent)); the pool obs6 then has world * P rows laid out
     * [n_inst][world * agents_local][6] and is filled every step by mpcx_allgather_states from obs_local. */
    int32_t n_inst, agents_local;
    double *obs_local /*P,6 scratch*/;
} mpcx_closed_loop;
int32_t mpcx_closed_loop_run(mpcx_ctx *ctx, const mpcx_interaction_params *ip, const mpcx_closed_loop *cl,
                             int32_t n_steps, int32_t use_graph);
/* run statistics accumulated on the device by every step of mpcx_closed_loop_run since the last reset (what the reference's scripts
 * print per run: solver failures; plus iteration counts): out4 = (agent-steps, interior-point iterations, failed solves, max iterations).
 * Synchronises the context's stream. */
int32_t mpcx_closed_loop_stats(mpcx_ctx *ctx, int64_t *out4 /*host*/, int32_t reset);

/* ---- multi-GPU exchange (the reference is single-process and has no counterpart).  One process per
 * GPU, one communicator per context: rank 0 calls mpcx_comm_unique_id, the caller distributes the MPCX_COMM_ID_BYTES bytes
 * to every rank by whatever means it has (torch.distributed broadcast in this package), every rank calls mpcx_comm_init.
 * mpcx_allgather_states is ONE RCCL all-gather over xGMI of 6-double agent states (x, y, v, yaw, accel, steer), enqueued on
 * the context's stream:
 *   MPCX_SHARD_INSTANCES  rank r holds n_inst instances x agents_local (= all) agents; all = the rank blocks one after the other
 *                         ([world * n_inst][agents_local][6]).  The instance-sharded data path itself needs no exchange
 *                         (all agents of an instance are rank-local); this form serves logging / result collection.
 *   MPCX_SHARD_AGENTS     rank r holds agents r*agents_local .. of EVERY one of n_inst instances; all = [n_inst][world *
 *                         agents_local][6], the obstacle pool of mpcx_interaction_batch on every rank (obs_off[p] = instance *
 *                         world * agents_local, obs_cnt[p] = world * agents_local, obs_skip[p] = the agent's own row).
 * Without a communicator (single rank) the call is a device copy.  local and all are DEVICE pointers. */
#define MPCX_COMM_ID_BYTES 128
#define MPCX_SHARD_INSTANCES 1
#define MPCX_SHARD_AGENTS 2
int32_t mpcx_comm_unique_id(void *id /*MPCX_COMM_ID_BYTES, host*/);
int32_t mpcx_comm_init(mpcx_ctx *ctx, int32_t world, int32_t rank, const void *id /*MPCX_COMM_ID_BYTES, host*/);
int32_t mpcx_comm_destroy(mpcx_ctx *ctx);
int32_t mpcx_allgather_states(mpcx_ctx *ctx, int32_t layout, int32_t n_inst, int32_t agents_local,
                              const double *local /*n_inst,agents_local,6*/, double *all /*see layout*/);

/* ---- scheduling hint for the next mpcx_qp_solve_batch calls (results never depend on it).  Interior-point iteration counts
 * are 5 for most problems with a tail to ~17, and one late-drawn hard problem ends the launch alone, so the work queue is
 * sorted longest-expected-first (counting sort, 64 bins) by
 *     key[b] = prev_iters[b]  (+ 6 if ref_now[b] != ref_prev[b]),
 * prev_iters = iterations problem b took in the previous MPC step (correlation with this step ~0.5), ref_now / ref_prev = any
 * pair of int32 arrays whose inequality marks a discontinuous change of the problem's reference since then -- the closed loop
 * passes the cut lengths of the current and the previous step (problems whose path cut moved take 8.0 iterations on average,
 * the others 5.3).  List-scheduling on recorded counts: ideal 28.7 rounds, FIFO 42-44, this order 31-32.  DEVICE pointers,
 * read at the start of each solve (prev_iters may alias the `iters` output); each may be NULL; all NULL = FIFO.
 * mpcx_closed_loop_run applies the hint by itself. */
int32_t mpcx_qp_set_order_hint(mpcx_ctx *ctx, const int32_t *prev_iters /*B or NULL*/, const int32_t *ref_now /*B or NULL*/,
                               const int32_t *ref_prev /*B or NULL*/);

/* ---- which kernel solves the QP:
 *   0 = automatic: the stage-structured solver for batches of >= 11264 problems or T > 20 (throughput: 8 problems per
 *       wavefront, O(T) work per iteration), the condensed solver below that (latency: 0.1-0.3 ms per launch against a
 *       0.4-0.8 ms floor);
 *   1 = condensed (csrc/mpcx_qp.hip, one wavefront per problem, any T <= MPCX_T_MAX, not competitive beyond T = 20);
 *   2 = stage-structured (csrc/mpcx_qp_quad.hip, eight lanes per problem, any T <= MPCX_T_MAX).
 * Same problem, same iteration, same exit rules: the choice changes speed, not results (agreement <= 1e-9 is tested).  The
 * environment variable MPCX_QP_KERNEL=wave|stage sets the default of new contexts. */
int32_t mpcx_set_qp_solver(mpcx_ctx *ctx, int32_t which);

/* ---- measurement hook: while enabled, every mpcx_qp_solve_batch launch (direct or through mpcx_closed_loop_run
 * without a graph) is bracketed by a pair of HIP events on the context's stream.  mpcx_profile_qp_read waits for
 * the recorded launches, returns their summed duration and count, and clears the record.  mpcx_closed_loop_run refuses use_graph
 * while the hook is on (launches inside a replayed graph cannot be bracketed). */
int32_t mpcx_profile_qp(mpcx_ctx *ctx, int32_t enable);
int32_t mpcx_profile_qp_read(mpcx_ctx *ctx, double *total_ms, int32_t *launches);

/* ---- per-instance tuning: the quantities main/lib/mpc_sensitivity.py:150-163 re-reads from
 * config/mpc_config_sensitivity.json before every solve, as one row per problem, so that a whole sensitivity sweep
 * (scenarios/mpc_sensitivity_analysis.py: one closed loop per parameter set) runs as ONE batch.  While rows are set,
 * problem b of every mpcx_qp_solve_batch / mpcx_plant_step_batch / mpcx_closed_loop_run call takes these fields
 * from rows[b] instead of the context-wide parameter set -- the batch size must equal n_rows; everything else (T, dt, L, speed and
 * steering limits, R_end, tolerances) still comes from mpcx_set_mpc_params.  rows is a DEVICE pointer that must
 * stay valid while set; (NULL, 0) clears. */
typedef struct {
    double w_perp, w_para;
    double R[2], Rd[2], Q_v_yaw[2];
    double Qf[4];        /* ALREADY multiplied by T */
    double max_accel, max_decel, max_dsteer /* rad/s */;
    double reserved;
} mpcx_qp_tuning;        /* 16 doubles */
int32_t mpcx_set_instance_tuning(mpcx_ctx *ctx, const mpcx_qp_tuning *rows, int32_t n_rows);

#ifdef __cplusplus
}
#endif
#endif
