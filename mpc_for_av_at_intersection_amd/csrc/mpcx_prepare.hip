// mpcx_prepare.hip -- reference window, warm-start rollout and plant update, batched.
//
// Replaces (paths relative to /root/reference/main):
//   lib/mpc.py:86-109   _calc_ref_trajectory      -> ref_window_kernel  (one wavefront per instance)
//   lib/trajectories.py:100-126 calc_nearest_index_in_direction (inside the above)
//   lib/mpc.py:112-126  _predict_motion            -> rollout_kernel     (one thread per instance)
//   lib/simulation.py:35-47 Simulation.step + bicycle/main.py:28-41      -> plant_kernel / rollout
// Arithmetic follows the reference's operation order; products are kept un-fused (-ffp-contract is
// irrelevant here because every product/sum below goes through __dmul_rn/__dadd_rn where order matters).
#include "mpcx_common.h"

namespace mpcx {

struct RefArgs {
    mpcx_mpc_params p;
    int B;
    const double *state, *path, *path_v;
    const int32_t *path_off, *path_len;
    double dl;
    int32_t *target_ind;
    double *xref;
    uint8_t *re;
    const double *ov;       // speeds of the previous linearisation pass (mpc.py:226-237, MAX_ITER > 1): row b at ov + b * ov_stride, or nullptr
    long ov_stride;
    const int32_t *bin_cnt, *keyslot;   // closed loop: the conflict search filed agent b as (key, slot) = keyslot[b]; its place in the QP work queue
    int32_t *order;                     // (keys descending) = number of agents with a larger key + slot.  nullptr: no queue order is built
    // closed loop: the conflict search ran the same nearest-index scan for this agent (same state, same path) from near[3b]; near[3b+1] /
    // [3b+2] = the largest / smallest of its three nearest indices or -1, near_tidx[b] = its answer.  nullptr: always scan
    const int32_t *near, *near_tidx;
};

__device__ __forceinline__ void ref_window_block(const RefArgs &a, int b) {
    const int lane = threadIdx.x & 63;
    const int T = a.p.T, W = T + 1;
#ifndef MPCX_WINDOW_ABLATE_ORDER      /* dev aid (timing only): what the queue scatter costs */
    if (a.order) {
        // lane k holds bin k.  Agents with a larger key come first (suffix sums over the bins by shuffles), then the agents of the same key
        // that counted in a lower copy of the bins, then the slot the conflict search drew
        static_assert(MPCX_ORDER_BINS == 64, "one bin per lane");
        const int mine = b % MPCX_ORDER_COPIES;
        int c = 0, lower = 0;
#pragma unroll
        for (int q = 0; q < MPCX_ORDER_COPIES; q++) {
            const int v = a.bin_cnt[q * MPCX_ORDER_BINS + lane];
            c += v;
            lower += q < mine ? v : 0;
        }
        int sfx = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_down(sfx, d); sfx += lane + d < 64 ? o : 0; }
        const int ks = a.keyslot[b];
        const int first = __shfl(sfx - c + lower, ks >> 24);
        if (lane == 0) a.order[first + (ks & 0xFFFFFF)] = b;
    }
#endif
    const double *path = a.path + 3 * (size_t)a.path_off[b];
    const double *pv = a.path_v ? a.path_v + (size_t)a.path_off[b] : nullptr;   // mpc_with_speed.py:103-104
    const int n = a.path_len[b];
    const double x = a.state[4 * b], y = a.state[4 * b + 1], v = a.state[4 * b + 2];
    int start = a.target_ind[b];
#ifdef MPCX_WINDOW_ABLATE_SCAN          /* dev aid (timing only): what the nearest-index scan costs */
    int s = start;
#else
    // The three nearest points of path[start .. n) are those of the conflict search's scan over path[hs .. len) whenever this range lies
    // inside that one (start >= hs) and holds all three (the three smallest of a set are the three smallest of every subset that contains
    // them; ties go to the lower index in both; the answer is a function of their absolute indices): then the answer is the conflict
    // search's and the scan is skipped.  Else (an earlier start, a cut in front of one of the three, an ego that did not advance): scan.
    int s;
    const int hs = a.near ? a.near[3 * b] : -1, hm = a.near ? a.near[3 * b + 1] : -1, hl = a.near ? a.near[3 * b + 2] : -1;
    if (hm >= 0 && start >= hs && hl >= start && hm < n) s = a.near_tidx[b];
    else s = (start < 0 || n <= 0) ? -1 : nearest_index_in_direction(path, n, start, x, y, lane);
#endif
    if (lane == 0) a.target_ind[b] = s;
    double *xr = a.xref + (size_t)b * 4 * W;
    uint8_t *re = a.re + (size_t)b * W;
    if (s < 0) {  // reference raised: leave a defined (zero) window, caller sees target_ind = -1
        if (lane <= T) { xr[lane] = 0; xr[W + lane] = 0; xr[2 * W + lane] = 0; xr[3 * W + lane] = 0; re[lane] = 0; }
        return;
    }
    // mpc.py:95-100: ov = max(v, 10/3.6); travel = cumsum(|ov|*dt); idx = min(rint(travel/dl) + s, n-1)
    // from the second of MAX_ITER linearisation passes on, ov = the previous pass's speeds (mpc.py:226-237)
    const double ov = v > 10.0 / 3.6 ? v : 10.0 / 3.6;
    const double step = __dmul_rn(fabs(ov), a.p.dt);
    const double *ovp = a.ov ? a.ov + (size_t)b * a.ov_stride : nullptr;
    if (lane <= T) {
        double travel = ovp ? __dmul_rn(fabs(ovp[0]), a.p.dt) : step;                       // np.cumsum: sequential adds
        for (int k = 1; k <= lane; k++) travel = __dadd_rn(travel, ovp ? __dmul_rn(fabs(ovp[k]), a.p.dt) : step);
        long long idx = (long long)rint(__ddiv_rn(travel, a.dl)) + s;
        if (idx > n - 1) idx = n - 1;
        xr[0 * W + lane] = path[3 * idx];
        xr[1 * W + lane] = path[3 * idx + 1];
        xr[2 * W + lane] = pv ? pv[idx] : 0.0;
        xr[3 * W + lane] = path[3 * idx + 2];
        re[lane] = (idx == n - 1);
    }
}

// simulation.py:35-47 + bicycle/main.py:28-41
__device__ __forceinline__ void plant_step(const mpcx_mpc_params &p, double &x, double &y, double &v, double &th,
                                           double a, double delta) {
    delta = fmax(fmin(delta, p.max_steer), -p.max_steer);
    double s, c;
    sincos(th, &s, &c);
    const double xd = __dmul_rn(v, c), yd = __dmul_rn(v, s), td = __dmul_rn(__ddiv_rn(v, p.L), tan(delta));
    x = __dadd_rn(x, __dmul_rn(xd, p.dt));
    y = __dadd_rn(y, __dmul_rn(yd, p.dt));
    th = __dadd_rn(th, __dmul_rn(td, p.dt));
    v = __dadd_rn(v, __dmul_rn(a, p.dt));
    v = fmax(fmin(v, p.max_speed), p.min_speed);
}

struct RollArgs {
    mpcx_mpc_params p;
    int B;
    const double *state, *u_warm;
    double *xbar;
};

// Rollout of 64 instances by one wavefront (lane = instance: the reference's operation order, one dependent chain of T x (sincos, tan,
// divide) per instance), results staged in LDS and written as ONE contiguous run: the 64 instances' xbar rows are adjacent in memory
// (64 x 4 x (T+1) doubles).  Until round 3 every lane stored its doubles straight to memory, 8 bytes at a 4 (T+1) x 8-byte lane stride,
// each its own write transaction: 165 MB of write traffic for 45 MB of output.
// (round 4: the staging buffer is dynamic LDS sized for the horizon in use -- 43.5 KB at T = 20 instead of 68 KB for T = 32: this kernel runs
// beside the conflict search on the side stream, and its two workgroups per CU left that kernel's 6.5-KB workgroups 24 KB of a CU's LDS)
__global__ __launch_bounds__(64) void rollout_kernel(RollArgs a) {
    extern __shared__ double s_roll[];                   // [64][4 W + 1]; +1: rows of 4 W doubles would sit 8 lanes to a bank group
    const int T = a.p.T, W = T + 1;
    const int RS = 4 * W + 1;
    auto s_x = [&](int lane) -> double * { return s_roll + (size_t)lane * RS; };
    const int b0 = (int)blockIdx.x * 64;
    const int n = a.B - b0 < 64 ? a.B - b0 : 64;
    if (threadIdx.x < (unsigned)n) {
        const int b = b0 + (int)threadIdx.x;
        double x = a.state[4 * b], y = a.state[4 * b + 1], v = a.state[4 * b + 2], th = a.state[4 * b + 3];
        double *xb = s_x(threadIdx.x);
        xb[0] = x; xb[W] = y; xb[2 * W] = v; xb[3 * W] = th;
        const double *oa = a.u_warm ? a.u_warm + (size_t)b * 2 * T : nullptr;
        for (int t = 1; t <= T; t++) {
            const double ai = oa ? oa[t - 1] : 0.0, di = oa ? oa[T + t - 1] : 0.0;   // mpc.py:222-224: zeros when no warm start
            plant_step(a.p, x, y, v, th, ai, di);
            xb[t] = x; xb[W + t] = y; xb[2 * W + t] = v; xb[3 * W + t] = th;
        }
    }
    __syncthreads();
    double *out = a.xbar + (size_t)b0 * 4 * W;
    for (int i = threadIdx.x; i < n * 4 * W; i += blockDim.x) out[i] = s_x(i / (4 * W))[i % (4 * W)];
}

// The two halves of mpc.py:211-239's preparation are independent -- the rollout is a chain of T dependent sincos / tan evaluations per
// instance (44 us), the window selection a scan of the path (31 us) -- and run BESIDE each other: the rollout on the context's side
// stream (fork / join by events, capturable into the closed loop's hipGraph), the window selection on the context's stream.  (Until
// round 3 both were workgroups of one launch; the rollout's 68-KB staging buffer would have been allocated for every workgroup of it.)
// Window workgroups of four wavefronts = four instances: fewer, larger workgroups to dispatch (57 -> 54 us).
constexpr int PREP_WAVES = 4;
__global__ __launch_bounds__(64 * PREP_WAVES) void ref_window_kernel(RefArgs ra) {
    const int b = (int)blockIdx.x * PREP_WAVES + (threadIdx.x >> 6);
    if (b < ra.B) ref_window_block(ra, b);
}

struct PlantArgs {
    mpcx_mpc_params p;
    int B;
    double *state;
    double *u;
    const int32_t *status;
    double *applied;
    const mpcx_qp_tuning *tune;   // per-instance MAX_DECEL or nullptr
    const int32_t *iters;         // closed loop only: the solver's iteration counts ...
    unsigned long long *stats;    // ... and the run statistics they are added to: agent-steps, iterations, failed solves, max iterations
    int has_stats;
    int32_t *zero_bins, *zero_ticket;   // closed loop: the queue bins and the ticket are zeroed for the next step (nullptr otherwise)
};

// MPC.step's tail (mpc.py:294-297) + Simulation.step
__global__ __launch_bounds__(256) void plant_kernel(PlantArgs a) {
    if (a.zero_bins && blockIdx.x == 0) {
        for (int i = threadIdx.x; i < MPCX_ORDER_COPIES * MPCX_ORDER_BINS; i += blockDim.x) a.zero_bins[i] = 0;
        if (threadIdx.x < MPCX_TICKET_WORDS) a.zero_ticket[threadIdx.x] = 0;
    }
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (a.has_stats) {          // run statistics (mpcx_closed_loop_stats)
        const bool in = b < a.B;
        const int it = in ? a.iters[b] : 0;
        const int bad = (in && a.status[b] != MPCX_QP_OPTIMAL) ? 1 : 0;
        int s_it = it, s_bad = bad, s_n = in ? 1 : 0, s_mx = it;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            s_it += __shfl_xor(s_it, d, WAVE); s_bad += __shfl_xor(s_bad, d, WAVE); s_n += __shfl_xor(s_n, d, WAVE);
            const int o = __shfl_xor(s_mx, d, WAVE); s_mx = o > s_mx ? o : s_mx;
        }
        if ((threadIdx.x & 63) == 0) {
            // one slot of four counters per wavefront, touched by that wavefront only (steps are ordered by the stream): no atomics --
            // 2048 atomics on four hot words made this 5-us kernel a 22-us one
            unsigned long long *w = a.stats + 4 * (size_t)(b >> 6);
            w[0] += (unsigned long long)s_n; w[1] += (unsigned long long)s_it; w[2] += (unsigned long long)s_bad;
            if ((unsigned long long)s_mx > w[3]) w[3] = (unsigned long long)s_mx;
        }
    }
    if (b >= a.B) return;
    const int T = a.p.T;
    double di = a.applied[2 * b], ai;
    const bool ok = !a.status || a.status[b] == MPCX_QP_OPTIMAL;
    if (ok) { di = a.u[(size_t)b * 2 * T + T]; ai = a.u[(size_t)b * 2 * T]; }
    else {
        ai = a.tune ? a.tune[b].max_decel : a.p.max_decel;
        for (int t = 0; t < 2 * T; t++) a.u[(size_t)b * 2 * T + t] = 0.0;   // warm start reset, mpc.py:222-224
    }
    a.applied[2 * b] = di; a.applied[2 * b + 1] = ai;
    double x = a.state[4 * b], y = a.state[4 * b + 1], v = a.state[4 * b + 2], th = a.state[4 * b + 3];
    plant_step(a.p, x, y, v, th, ai, di);
    a.state[4 * b] = x; a.state[4 * b + 1] = y; a.state[4 * b + 2] = v; a.state[4 * b + 3] = th;
}

}  // namespace mpcx

extern "C" int32_t mpcx_mpc_prepare_batch(mpcx_ctx *ctx, int32_t B, const double *state, const double *u_warm,
                                          const double *path_xyyaw, const double *path_v, const int32_t *path_off,
                                          const int32_t *path_len, double dl, int32_t *target_ind, double *xref, uint8_t *reaches_end, double *xbar) {
    return mpcx_mpc_prepare_batch_ov(ctx, B, state, u_warm, path_xyyaw, path_v, path_off, path_len, dl, target_ind, nullptr, 0, xref, reaches_end, xbar);
}

extern "C" int32_t mpcx_mpc_prepare_batch_ov(mpcx_ctx *ctx, int32_t B, const double *state, const double *u_warm,
                                             const double *path_xyyaw, const double *path_v, const int32_t *path_off,
                                             const int32_t *path_len, double dl, int32_t *target_ind, const double *ov, int64_t ov_stride,
                                             double *xref, uint8_t *reaches_end, double *xbar) {
    if (!ctx) return MPCX_E_INVALID;
    if (!ctx->have_mpc) return mpcx_fail(ctx, MPCX_E_INVALID, "mpcx_set_mpc_params has not been called");
    if (B == 0) return MPCX_OK;       // empty batch: nothing to do (zero-size tensors have null data pointers)
    if (B < 0 || !state || !path_xyyaw || !path_off || !path_len || !target_ind || !xref || !reaches_end || !xbar || !(dl > 0))
        return mpcx_fail(ctx, MPCX_E_INVALID, "mpc_prepare_batch: null pointer, negative batch or dl <= 0");
    if (B == 0) return MPCX_OK;
    if (ov && ov_stride < (int64_t)ctx->mpc.T + 1)
        return mpcx_fail(ctx, MPCX_E_INVALID, "mpc_prepare_batch_ov: ov_stride %lld is smaller than T + 1", (long long)ov_stride);
    const bool scatter = ctx->bin_scatter;
    ctx->bin_scatter = false;
    mpcx::RefArgs ra{ctx->mpc, B, state, path_xyyaw, path_v, path_off, path_len, dl, target_ind, xref, reaches_end, ov, (long)ov_stride,
                     scatter ? ctx->bins : nullptr, scatter ? ctx->bins + MPCX_ORDER_COPIES * MPCX_ORDER_BINS : nullptr, scatter ? ctx->order : nullptr,
                     ctx->window_near, ctx->window_near ? ctx->window_tidx : nullptr};
    // the rollout may already be in flight: mpcx_closed_loop_run forks it at the start of the step, beside the conflict search
    const bool forked = ctx->rollout_forked;
    ctx->rollout_forked = false;
    if (!forked) {
        int32_t rc = mpcx_rollout_fork(ctx, B, state, u_warm, xbar);
        if (rc != MPCX_OK) return rc;
    }
    // a rollout forked long ago (mpcx_closed_loop_run: at the start of the step) is joined IN FRONT of the window selection: the queue
    // works the barrier off while the conflict search is still running, and nothing stands between the window kernel and the solve;
    // a rollout forked just now runs beside the window selection and is joined behind it
    if (forked && hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0) != hipSuccess)
        return mpcx_fail(ctx, MPCX_E_LAUNCH, "mpc_prepare_batch: cannot join the side stream");
    hipLaunchKernelGGL(mpcx::ref_window_kernel, dim3((B + mpcx::PREP_WAVES - 1) / mpcx::PREP_WAVES), dim3(64 * mpcx::PREP_WAVES), 0, ctx->stream, ra);
    if (!forked && hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0) != hipSuccess)
        return mpcx_fail(ctx, MPCX_E_LAUNCH, "mpc_prepare_batch: cannot join the side stream");
    return mpcx_check_launch(ctx, "prepare kernels");
}

// the warm-start rollout (mpc.py:112-126 `_predict_motion`) on the context's side stream, ordered behind everything enqueued on the
// context's stream so far; the next mpcx_mpc_prepare_batch[_ov] joins it instead of launching its own
int32_t mpcx_rollout_fork(mpcx_ctx *ctx, int32_t B, const double *state, const double *u_warm, double *xbar) {
    mpcx::RollArgs ro{ctx->mpc, B, state, u_warm, xbar};
    if (hipEventRecord(ctx->ev_fork, ctx->stream) != hipSuccess || hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0) != hipSuccess)
        return mpcx_fail(ctx, MPCX_E_LAUNCH, "mpc_prepare_batch: cannot fork the side stream");
    hipLaunchKernelGGL(mpcx::rollout_kernel, dim3((B + 63) / 64), dim3(64), 64 * (4 * (size_t)(ctx->mpc.T + 1) + 1) * sizeof(double), ctx->side, ro);
    // the join event right behind the rollout: by the time the context's stream waits for it (in front of the solve) the marker has long
    // been processed -- recorded there, the wait paid for the side queue's marker AND its own barrier
    if (hipEventRecord(ctx->ev_join, ctx->side) != hipSuccess)
        return mpcx_fail(ctx, MPCX_E_LAUNCH, "mpc_prepare_batch: cannot record the join event");
    return MPCX_OK;
}

extern "C" int32_t mpcx_plant_step_batch(mpcx_ctx *ctx, int32_t B, double *state, double *u,
                                         const int32_t *status, double *applied) {
    if (!ctx) return MPCX_E_INVALID;
    if (!ctx->have_mpc) return mpcx_fail(ctx, MPCX_E_INVALID, "mpcx_set_mpc_params has not been called");
    if (B == 0) return MPCX_OK;       // empty batch: nothing to do (zero-size tensors have null data pointers)
    if (B < 0 || !state || !u || !applied) return mpcx_fail(ctx, MPCX_E_INVALID, "plant_step_batch: null pointer");
    if (B == 0) return MPCX_OK;
    if (ctx->tune && ctx->tune_rows != B)
        return mpcx_fail(ctx, MPCX_E_INVALID, "plant_step_batch: %d tuning rows are set but the batch has %d agents", ctx->tune_rows, B);
    // inside mpcx_closed_loop_run the step also feeds the run statistics (ctx->stats_iters names the step's iteration counts)
    const bool st = ctx->stats && ctx->stats_iters && status;
    const bool rz = ctx->bin_reset && ctx->bins && ctx->ticket;
    ctx->bin_reset = false;
    mpcx::PlantArgs pa{ctx->mpc, B, state, u, status, applied, ctx->tune, ctx->stats_iters, ctx->stats, st ? 1 : 0,
                       rz ? ctx->bins : nullptr, rz ? ctx->ticket : nullptr};
    hipLaunchKernelGGL(mpcx::plant_kernel, dim3((B + 63) / 64), dim3(64), 0, ctx->stream, pa);
    return mpcx_check_launch(ctx, "plant_kernel");
}

extern "C" int32_t mpcx_closed_loop_stats(mpcx_ctx *ctx, int64_t *out4, int32_t reset) {
    if (!ctx || !out4) return MPCX_E_INVALID;
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    if (!ctx->stats || !ctx->stats_slots) return MPCX_OK;
    std::vector<unsigned long long> h(4 * ctx->stats_slots);
    if (hipMemcpyAsync(h.data(), ctx->stats, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return mpcx_fail(ctx, MPCX_E_LAUNCH, "closed_loop_stats: copy failed");
    for (size_t w = 0; w < ctx->stats_slots; w++) {
        out4[0] += (int64_t)h[4 * w]; out4[1] += (int64_t)h[4 * w + 1]; out4[2] += (int64_t)h[4 * w + 2];
        if ((int64_t)h[4 * w + 3] > out4[3]) out4[3] = (int64_t)h[4 * w + 3];
    }
    if (reset && hipMemsetAsync(ctx->stats, 0, h.size() * sizeof(unsigned long long), ctx->stream) != hipSuccess)
        return mpcx_fail(ctx, MPCX_E_LAUNCH, "closed_loop_stats: reset failed");
    return MPCX_OK;
}
