// mpcx_common.h -- shared device helpers for the libmpcx.so kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mpcx.h"
#include <vector>

struct mpcx_ctx {
    int device;
    hipStream_t stream;
    mpcx_mpc_params mpc;
    bool have_mpc;
    int32_t *ticket;     // device words: [0] work-queue head of the persistent QP kernel, [4] the list of problems the condensed solver gives up on, [5] the head of the launch that works that list off; 8 allocated
    int n_cu;            // compute units of the device
    double *pred;        // scratch: predicted obstacle disc centres [NOBS][steps][2 discs][2]
    size_t pred_cap;     // capacity of pred in doubles
    hipGraphExec_t loop_exec;   // cached one-step graph of mpcx_closed_loop_run (nullptr = none)
    unsigned char loop_key[768]; // descriptor + parameters the cached graph was captured for
    const mpcx_qp_tuning *tune; // per-instance tuning rows (device) or nullptr
    int32_t tune_rows;
    const int32_t *order_hint;  // iteration counts of a previous solve (device) or nullptr (mpcx_qp_set_order_hint)
    const int32_t *order_now, *order_prev;   // optional pair: entries that differ mark a discontinuous change of the reference
    int32_t *prev_cut;          // scratch of mpcx_closed_loop_run: cut lengths of the previous step
    size_t prev_cut_cap;
    int32_t *order;             // scratch: work-queue order built from the hint | per-block key histograms | list of given-up problems | 2 counters
    size_t order_cap;
    bool order_ready = false;   // mpcx_closed_loop_run: the order of this step's first solve is in `order` already and the ticket is zero
    // closed loop: the counting sort of the work queue rides in the kernels of the step instead of two launches and two fills of its own.
    // The conflict search files every agent under its queue key (bins[key]++ -> slot), the window selection turns (key, slot) into the
    // agent's place in `order`, the plant kernel zeroes the bins and the ticket for the next step.
    int32_t *bins = nullptr;    // [MPCX_ORDER_COPIES][MPCX_ORDER_BINS] counters (agent p counts in copy p % COPIES: 32 k atomics on 64 words are slow) | [P] (key << 24 | slot)
    size_t bins_cap = 0;
    bool bins_clean = false;    // host's knowledge: the counters and the ticket are zero (the last closed-loop step ran through and nothing has drawn tickets since)
    const int32_t *bin_hint = nullptr;    // set around the conflict search: iteration counts of the previous step (queue key)
    bool bin_scatter = false;             // set around the window selection: write `order`
    bool bin_reset = false;               // set around the plant step: zero bins and ticket
    const double *pack_state = nullptr, *pack_applied = nullptr;   // mpcx_closed_loop_run, local pool: the prediction kernel packs the pool rows itself
    int32_t *inter_prev_save = nullptr;   // mpcx_closed_loop_run: where the conflict search leaves the cut lengths it read (the queue order's `moved` test)
    // mpcx_closed_loop_run: the conflict search and the window selection both run calc_nearest_index_in_direction for the same agent, state
    // and path, mostly from the same start index.  The conflict search leaves (its start index, the largest of its three nearest indices
    // or -1) per agent here (behind prev_cut), and the window selection takes the conflict search's answer where that is provably its own.
    int32_t *inter_near = nullptr;        // set around the conflict search (2 ints per agent) ...
    const int32_t *window_near = nullptr, *window_tidx = nullptr;   // ... and around the window selection (+ the conflict search's updated traj_idx)
    hipStream_t side = nullptr; // side stream of mpcx_mpc_prepare_batch: the warm-start rollout runs beside the window selection (fork / join by events)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool rollout_forked = false; // mpcx_closed_loop_run has the rollout of this step in flight on the side stream (mpcx_rollout_fork)
    double *cs = nullptr;       // scratch of mpcx_expand_batch: (cos, sin) of the nodes' headings
    size_t cs_cap = 0;
    void *multi;                // scratch of mpcx_expand_multi_batch (segment descriptors + block tables)
    size_t multi_cap;
    int qp_solver;              // 0 = automatic, 1 = condensed (one wavefront per QP), 2 = stage-structured (mpcx_set_qp_solver)
    int lin_passes = 1;         // linearisation passes per step of mpcx_closed_loop_run (lib/mpc.py MAX_ITER; mpcx_set_linearisation_passes)
    bool prof_qp;               // bracket qp_kernel launches with events (mpcx_profile_qp)
    std::vector<hipEvent_t> prof_ev;   // start/stop pairs recorded so far
    std::vector<hipEvent_t> prof_free; // recycled events
    unsigned long long *stats = nullptr;    // run statistics of mpcx_closed_loop_run: per wavefront of the plant kernel (agent-steps, iterations, failures, max iterations)
    size_t stats_slots = 0;                 // wavefront slots allocated
    const int32_t *stats_iters = nullptr;   // set by mpcx_closed_loop_run around its plant step
    void *comm = nullptr;       // ncclComm_t (mpcx_comm_init) or nullptr = single rank
    int comm_world = 1, comm_rank = 0;
    double *xchg = nullptr;     // all-gather landing buffer of the agent-sharded layout
    size_t xchg_cap = 0;
    char err[256];
};

int32_t mpcx_fail(mpcx_ctx *ctx, int32_t code, const char *fmt, ...);
int32_t mpcx_check_launch(mpcx_ctx *ctx, const char *what);
int32_t mpcx_ensure_pred(mpcx_ctx *ctx, size_t need_doubles);   // prediction scratch (mpcx_interaction.hip)
int32_t mpcx_rollout_fork(mpcx_ctx *ctx, int32_t B, const double *state, const double *u_warm, double *xbar);   // mpcx_prepare.hip
int32_t mpcx_ensure_ticket(mpcx_ctx *ctx);                      // work-queue word (mpcx_qp.hip)
// Work-queue key: expected length of a solve.  hint = the previous step's iteration count; a problem whose path cut moved since
// the previous step starts far from its warm start and is counted as MPCX_JUMP_BONUS iterations (mpcx_qp.hip has the measurements).
#ifndef MPCX_JUMP_BONUS
#define MPCX_JUMP_BONUS 11
#endif
#define MPCX_ORDER_BINS 64
#define MPCX_TICKET_WORDS 8      /* ctx->ticket: the queue head and the other per-launch counters, zeroed together */
#define MPCX_ORDER_COPIES 16
namespace mpcx {
__device__ __forceinline__ int order_key_of(int hint, bool moved) {
    int k = hint < 0 ? 0 : hint;
    if (moved) k += MPCX_JUMP_BONUS;
    return k < MPCX_ORDER_BINS ? k : MPCX_ORDER_BINS - 1;
}
}
int32_t mpcx_ensure_order(mpcx_ctx *ctx, size_t B);             // work-queue order scratch (mpcx_qp.hip)
int32_t mpcx_qp_build_order(mpcx_ctx *ctx, int32_t B, hipStream_t st);   // counting sort of the work queue on stream st; also zeroes the ticket (mpcx_qp.hip)

// fraction of the step to the boundary the interior-point iteration takes (both solvers must agree, and the tests' CPU checker
// uses the same value).  0.995 in round 1; 0.999 saves 0.8 of 6.1 iterations on the closed-loop workload (numpy replica of the iteration over
// 1280 harvested QPs: mean 6.09 -> 5.27, 99th percentile 13 -> 13, max 15 -> 15; 0.9999 is worse again)
#ifndef MPCX_STEP_FRACTION
#define MPCX_STEP_FRACTION 0.999
#endif
#ifndef MPCX_SLACK_FLOOR
#define MPCX_SLACK_FLOOR 0.5    /* starting point of the iteration: s = max(slack, floor), lam = MPCX_LAM0 (same in mpcx_qp_stage.h) */
#endif
#ifndef MPCX_LAM0
#define MPCX_LAM0 3.0              /* 1 until round 2; with separate step lengths 3 takes the hardest problems of a launch from 23 to 18 iterations (2 / 5: 18 / 17, slower on average) */
#endif
#ifndef MPCX_TRIAL_STEP
#define MPCX_TRIAL_STEP 1      /* unconstrained trial step before the interior-point iteration: mpcx_qp_stage.h, mpcx_qp.hip (the tests' CPU checker follows the same rule) */
#endif

/* active-set polish at the end of the interior-point iteration: the rule and the constants are those of mpcx_qp_stage.h (the host
   build of that header has no other source; the tests' CPU checker carries the same values) */
#ifndef MPCX_POLISH
#define MPCX_POLISH 1
#endif
#ifndef MPCX_POLISH_MU
#define MPCX_POLISH_MU 1e-5
#endif
#ifndef MPCX_POLISH_RP
#define MPCX_POLISH_RP 1e-6
#endif
#ifndef MPCX_POLISH_RD
#define MPCX_POLISH_RD 1e-3
#endif
#ifndef MPCX_POLISH_RHO
#define MPCX_POLISH_RHO 1e8
#define MPCX_POLISH_TRIES 3
#define MPCX_POLISH_EPS_L 1e-9
#define MPCX_POLISH_EPS_G 1e-9
#endif

namespace mpcx {

constexpr int WAVE = 64;

struct QpArgs {
    mpcx_mpc_params p;
    int B;
    int32_t *ticket;      // work queue head (zeroed before the launch): wavefronts draw QP indices until B is exhausted
    int has_warm;         // u_warm != NULL (tested on the host: a device-side null test of a kernel-argument pointer trips a
                          // gfx950 instruction-selection bug in some register-allocation outcomes)
    const double *x0, *xref, *xbar, *u_warm;
    const uint8_t *re;
    double *x_out, *u_out, *kkt;
    int32_t *status, *iters;
    const mpcx_qp_tuning *tune;   // per-problem rows or nullptr
    int has_tune;                 // tune != NULL, tested on the host like has_warm
    const int32_t *order;         // ticket -> problem index (hard problems first) or nullptr = identity
    int has_order;
    // second chance for problems the condensed solver gives up on (MAXITER / NUMERIC): with defer_fail it leaves their outputs
    // untouched and appends their indices to fail_list; the stage solver then runs over that list, whose length it reads from
    // *queue_len (has_queue_len) instead of B
    int defer_fail;
    int32_t *fail_list, *fail_count;
    const int32_t *queue_len;
    int has_queue_len;
};

void launch_qp_stage(const QpArgs &a, hipStream_t st, int n_cu);   // mpcx_qp_quad.hip
int qp_stage_grid(int B, int n_cu);                                 // wavefronts launch_qp_stage starts for B problems (mpcx_qp_quad.hip)


__device__ __forceinline__ double rdlane(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
// 1/d and 1/sqrt(d): hardware seed + two Newton steps (full double accuracy, not correctly rounded)
__device__ __forceinline__ double frcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double frsq(double d) {
    double y = __builtin_amdgcn_rsq(d);
    double h = 0.5 * d;
    y = fma(y, fma(-h * y, y, 0.5), y);
    y = fma(y, fma(-h * y, y, 0.5), y);
    return y;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmax(v, __shfl_xor(v, d, WAVE));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmin(v, __shfl_xor(v, d, WAVE));
    return v;
}
// inclusive prefix sum over lanes (lane i gets sum of lanes 0..i)
__device__ __forceinline__ double scan_up(double v, int lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        double t = __shfl_up(v, d, WAVE);
        if (lane >= d) v += t;
    }
    return v;
}
// inclusive suffix sum over lanes (lane i gets sum of lanes i..63)
__device__ __forceinline__ double scan_down(double v, int lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        double t = __shfl_down(v, d, WAVE);
        if (lane + d < WAVE) v += t;
    }
    return v;
}

// ---------------------------------------------------------------- DPP cross-lane helpers (no LDS round trip)
// gfx9 DPP controls: row_shl:n 0x100+n, row_shr:n 0x110+n, wave_shl:1 0x130, wave_shr:1 0x138, row_bcast:15 0x142,
// row_bcast:31 0x143.  A "row" is 16 consecutive lanes.  Lanes whose source is out of range keep `old`.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_mov(double old, double src) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, ROW_MASK, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
// value of lane+1 / lane-1 (wave-wide shift by one; the edge lane gets `fill`)
__device__ __forceinline__ double lane_next(double v, double fill = 0.0) { return dpp_mov<0x130>(fill, v); }
__device__ __forceinline__ double lane_prev(double v, double fill = 0.0) { return dpp_mov<0x138>(fill, v); }

// inclusive prefix sum over lanes 0..31 (rows 0 and 1); lanes >= 32 return garbage-free but meaningless values
__device__ __forceinline__ double scan_up32(double v) {
    v += dpp_mov<0x111>(0.0, v);
    v += dpp_mov<0x112>(0.0, v);
    v += dpp_mov<0x114>(0.0, v);
    v += dpp_mov<0x118>(0.0, v);
    v += dpp_mov<0x142, 0xA>(0.0, v);      // row_bcast:15 into rows 1 and 3: add the previous row's total
    return v;
}
// inclusive suffix sum over lanes 0..31: lane i gets sum of lanes i..31
__device__ __forceinline__ double scan_down32(double v, int lane) {
    v += dpp_mov<0x101>(0.0, v);
    v += dpp_mov<0x102>(0.0, v);
    v += dpp_mov<0x104>(0.0, v);
    v += dpp_mov<0x108>(0.0, v);
    const double r1 = rdlane(v, 16);       // total of row 1
    return v + ((lane < 16) ? r1 : 0.0);
}
// wave-wide reductions whose result is wave-uniform (lanes 0..63)
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_mov<0x111>(0.0, v);
    v += dpp_mov<0x112>(0.0, v);
    v += dpp_mov<0x114>(0.0, v);
    v += dpp_mov<0x118>(0.0, v);           // lane 15 of each row holds the row total
    return (rdlane(v, 15) + rdlane(v, 31)) + (rdlane(v, 47) + rdlane(v, 63));
}
__device__ __forceinline__ double wave_max_dpp(double v) {
    v = fmax(v, dpp_mov<0x111>(v, v));
    v = fmax(v, dpp_mov<0x112>(v, v));
    v = fmax(v, dpp_mov<0x114>(v, v));
    v = fmax(v, dpp_mov<0x118>(v, v));
    return fmax(fmax(rdlane(v, 15), rdlane(v, 31)), fmax(rdlane(v, 47), rdlane(v, 63)));
}
// 1/d with ONE Newton step on the hardware seed (~1e-15 relative; used where the consumer is itself iterative)
__device__ __forceinline__ double frcp1(double d) {
    double r = __builtin_amdgcn_rcp(d);
    return fma(fma(-d, r, 1.0), r, r);
}

// lexicographic (distance, index) minimum over the wave
__device__ __forceinline__ void wave_argmin(double &d, int &i) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        double od = __shfl_xor(d, s, WAVE);
        int oi = __shfl_xor(i, s, WAVE);
        bool take = (od < d) || (od == d && oi < i);
        d = take ? od : d;
        i = take ? oi : i;
    }
}

__device__ __forceinline__ double pt_dist(const double *path, int idx, double x, double y) {
    double dx = __dadd_rn(path[3 * idx], -x), dy = __dadd_rn(path[3 * idx + 1], -y);
    return __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
}

// trajectories.py:100-126 on path[start .. n) ; returns absolute index or -1 ("something wrong").
// The three smallest (distance, index) pairs in ascending order, ties by lower index (= numpy's argpartition + argsort on these
// data): ONE pass over the points keeps each lane's three smallest (the next 64 points are in flight meanwhile; the square root is
// taken only where the squared distance could enter the lane's three: sqrt is monotone), then the wave-wide minimum is popped
// three times.  Round 1 made three passes with a square root per point each.
__device__ inline int nearest_index_in_direction(const double *path, int n, int start, double x, double y, int lane) {
    const int len = n - start;
    if (len <= 1) return start;
    if (len == 2) return start + 1;
    double b0d = INFINITY, b1d = INFINITY, b2d = INFINITY, b2s = INFINITY, b1s = INFINITY, b0s = INFINITY;
    int b0i = 0x7fffffff, b1i = 0x7fffffff, b2i = 0x7fffffff;
    // the points arrive in batches of DEPTH x 64 with the next batch's loads all in flight (one batch deep the scan waited for an
    // L2 round trip per 64 points: there is almost no arithmetic to hide it behind)
    constexpr int DEPTH = 4;
    double bx[DEPTH], by[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; k++) {
        const int i = k * WAVE + lane;
        const double *q = path + 3 * (size_t)(start + (i < len ? i : 0));        // clamped address, selected afterwards
        const double vx = q[0], vy = q[1];
        bx[k] = vx; by[k] = vy;
    }
    for (int i0 = 0; i0 < len; i0 += DEPTH * WAVE) {
        double nbx[DEPTH], nby[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) {
            const int i = i0 + (DEPTH + k) * WAVE + lane;
            const double *q = path + 3 * (size_t)(start + (i < len ? i : 0));
            nbx[k] = q[0]; nby[k] = q[1];
        }
#pragma unroll
        for (int k = 0; k < DEPTH; k++) {
            const int i = i0 + k * WAVE + lane;
            if (i < len) {
                const double dx = __dadd_rn(bx[k], -x), dy = __dadd_rn(by[k], -y);
                const double d2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
                if (d2 < b2s || b2i == 0x7fffffff) {
                    const double d = __dsqrt_rn(d2);
                    if (d < b2d || (d == b2d && i < b2i)) {
                        if (d < b1d || (d == b1d && i < b1i)) {
                            b2d = b1d; b2i = b1i; b2s = b1s;
                            if (d < b0d || (d == b0d && i < b0i)) { b1d = b0d; b1i = b0i; b1s = b0s; b0d = d; b0i = i; b0s = d2; }
                            else { b1d = d; b1i = i; b1s = d2; }
                        } else { b2d = d; b2i = i; b2s = d2; }
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < DEPTH; k++) { bx[k] = nbx[k]; by[k] = nby[k]; }
    }
    int bi[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {     // pop the wave-wide minimum three times
        double d = b0d; int ix = b0i;
        wave_argmin(d, ix);
        bi[r] = ix;
        if (b0i == ix) { b0d = b1d; b0i = b1i; b1d = b2d; b1i = b2i; b2d = INFINITY; b2i = 0x7fffffff; }
    }
    if (abs(bi[1] - bi[2]) == 2) return bi[0] + start;
    if (abs(bi[0] - bi[1]) == 1) return max(bi[0], bi[1]) + start;
    return -1;
}

}  // namespace mpcx
