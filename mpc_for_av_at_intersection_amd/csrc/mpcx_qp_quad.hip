// mpcx_qp_quad.hip -- device build of the stage-structured QP solver (mpcx_qp_stage.h): EIGHT lanes per problem, eight
// problems per wavefront (a four-lane geometry is kept for experiments).  Lane q of a group owns stages q*SPL ..
// q*SPL+SPL-1; the Riccati / costate / rollout sweeps hand their carry to the neighbour lane with row_shl:1 / row_shr:1 DPP
// moves (full rate, no LDS), group reductions are three DPP butterflies, slacks / multipliers / the state block of the gains
// live in LDS as [row][lane] (conflict-free, 36 KB per workgroup at SPL = 3 = four workgroups per CU), everything else in
// registers.  Wavefronts are persistent: groups draw problems from a global ticket (QueueSrc).
#include "mpcx_common.h"
#include <cstdlib>
#include "mpcx_qp_stage.h"

namespace mpcx {

typedef __attribute__((address_space(3))) double lds_double;

// group geometry: LQ_ = 4 (quad_perm selectors) or 8 (row shifts + half-row mirror), both inside one DPP row of 16 lanes
template <int LQ_, int SPL_, bool JERK_ = false>
struct GroupCx {
    static constexpr int LQ = LQ_, SPL = SPL_;
    static constexpr bool JERK = JERK_;       // the five-state problem of lib/mpc_jerk.py (mpcx_mpc_params.model)
    static_assert(LQ == 4 || LQ == 8, "groups of 4 or 8 lanes");
    int q, lane;
    lds_double *sh;                       // s [SPL*8][64], lam [SPL*8][64], gains [SPL*8][64], spare [2*SPL][64]
    // quad_perm [1,2,3,3] / [0,0,1,2] = value of lane q+1 / q-1; row_shl:1 / row_shr:1 do the same across a whole row.
    // The value arriving at a group's edge lane comes from the neighbouring group and is never used.
    __device__ __forceinline__ double nxt(double v) const { return LQ == 4 ? dpp_mov<0xF9>(v, v) : dpp_mov<0x101>(v, v); }
    __device__ __forceinline__ double prv(double v) const { return LQ == 4 ? dpp_mov<0x90>(v, v) : dpp_mov<0x111>(v, v); }
    // butterflies: quad_perm [1,0,3,2] (xor 1), [2,3,0,1] (xor 2), row_half_mirror (lane i <-> 7-i of each half row)
    __device__ __forceinline__ double gsum(double v) const {
        v += dpp_mov<0xB1>(v, v); v += dpp_mov<0x4E>(v, v);
        if (LQ == 8) v += dpp_mov<0x141>(v, v);
        return v;
    }
    __device__ __forceinline__ double gmax(double v) const {
        v = fmax(v, dpp_mov<0xB1>(v, v)); v = fmax(v, dpp_mov<0x4E>(v, v));
        if (LQ == 8) v = fmax(v, dpp_mov<0x141>(v, v));
        return v;
    }
    __device__ __forceinline__ double gmin(double v) const {
        v = fmin(v, dpp_mov<0xB1>(v, v)); v = fmin(v, dpp_mov<0x4E>(v, v));
        if (LQ == 8) v = fmin(v, dpp_mov<0x141>(v, v));
        return v;
    }
    __device__ __forceinline__ bool gany(bool b) const {
        const unsigned long long m = __ballot(b);
        return ((m >> (lane & ~(LQ - 1))) & ((1ull << LQ) - 1ull)) != 0ull;
    }
    __device__ __forceinline__ bool any(bool b) const { return __ballot(b) != 0ull; }
    __device__ __forceinline__ int count(bool b) const { return __popcll(__ballot(b)); }       // lanes, not groups
    __device__ __forceinline__ double rcp(double v) const { return frcp(v); }
    __device__ __forceinline__ double rcp_fast(double v) const { return frcp1(v); }     // seed + one Newton step
    // ratio tests only: the hardware seed; the 0.001 margin of the step fraction is four orders of magnitude wider than its error
    __device__ __forceinline__ double rcp_seed(double v) const { return __builtin_amdgcn_rcp(v); }
    // phase boundary: LDS values are re-read afterwards instead of being carried in registers across the phase
    __device__ __forceinline__ void fence() const { asm volatile("" ::: "memory"); }
#ifdef MPCX_STAGE_PROFILE
    // dev build: shader-clock time per phase, summed per wavefront (lane 0 adds to prof[phase] at the end)
    unsigned long long t_last = 0, t_acc[10] = {};
    __device__ __forceinline__ void stamp(int k) { const unsigned long long t = __builtin_amdgcn_s_memtime(); if (t_last) t_acc[k] += t - t_last; t_last = t; }
    unsigned char *occ = nullptr;        // dev build: per wavefront 64 rounds x (groups at work, of which in a trial / polish round)
    unsigned char *life = nullptr;       // dev build: per problem (round it was drawn in, round it was handed in: + 100 = handed over)
    __device__ __forceinline__ void lifetime(int b, int which, int round) { if (q == 0 && life) life[2 * b + which] = (unsigned char)round; }
    __device__ __forceinline__ void occupancy(int round, int groups, int special) { if (lane == 0 && occ && round < 64) { occ[2 * round] = (unsigned char)groups; occ[2 * round + 1] = (unsigned char)special; } }
#else
    __device__ __forceinline__ void stamp(int) const {}
#endif
#ifdef MPCX_STAGE_TRACE
    double *trace_buf = nullptr;          // dev build: 8 doubles per iteration of the problem in group 0 (needs room behind kkt)
    __device__ __forceinline__ void trace(int it, double a, double b, double c, double d, double e, double f) {
        if (lane == 0 && trace_buf && it < 64) { double *t = trace_buf + 8 * it; t[0] = a; t[1] = b; t[2] = c; t[3] = d; t[4] = e; t[5] = f; t[6] = 1.0; }
    }
#endif
    __device__ __forceinline__ double ld_s(int k) const { return sh[(0 * SPL * 8 + k) * 64 + lane]; }
    __device__ __forceinline__ double ld_l(int k) const { return sh[(1 * SPL * 8 + k) * 64 + lane]; }
    __device__ __forceinline__ double ld_k(int k) const { return sh[(2 * SPL * 8 + k) * 64 + lane]; }
    __device__ __forceinline__ void st_s(int k, double v) { sh[(0 * SPL * 8 + k) * 64 + lane] = v; }
    __device__ __forceinline__ void st_l(int k, double v) { sh[(1 * SPL * 8 + k) * 64 + lane] = v; }
    __device__ __forceinline__ void st_k(int k, double v) { sh[(2 * SPL * 8 + k) * 64 + lane] = v; }
    __device__ __forceinline__ double ld_w(int k) const { return sh[(3 * SPL * 8 + k) * 64 + lane]; }     // per-problem constants (2 * SPL per lane)
    __device__ __forceinline__ void st_w(int k, double v) { sh[(3 * SPL * 8 + k) * 64 + lane] = v; }
};

// work queue: group leaders draw problem indices from a global ticket until the batch is exhausted
template <int LQ, int SPL, bool TUNED>
struct QueueSrc {
    const QpArgs &a;
    __device__ __forceinline__ mpcx_mpc_params params() const { return a.p; }
    __device__ __forceinline__ mpcx_stage::Problem at(int b) const {
        const int T = a.p.T, W = T + 1;
        return mpcx_stage::Problem{a.x0 + (size_t)b * 4, a.xref + (size_t)b * 4 * W, a.xbar + (size_t)b * 4 * W,
                                   a.has_warm ? a.u_warm + (size_t)b * 2 * T : nullptr, a.re + (size_t)b * W,
                                   a.x_out + (size_t)b * 4 * W, a.u_out + (size_t)b * 2 * T, a.kkt + (size_t)b * 4,
                                   a.status + b, a.iters + b};
    }
    // every round either advances some group's iteration counter or consumes a ticket
#ifndef MPCX_REFILL_GROUPS
#define MPCX_REFILL_GROUPS 2
#endif
    __device__ __forceinline__ int refill_min() const { return MPCX_REFILL_GROUPS * LQ; }   // in lanes
    __device__ __forceinline__ long max_rounds() const { return ((long)a.B + 2) * (long)(a.p.max_iter + 6) * (MPCX_POLISH_TRIES + 1); }
    template <class Cx>
    __device__ __forceinline__ bool fetch(Cx &cx, mpcx_mpc_params &P, int &pbi) const {
        int t = 0;
        if (cx.q == 0) t = atomicAdd(a.ticket, 1);
        t = (int)cx.gsum((double)t);                  // the other lanes contribute 0: everybody gets the leader's ticket
        const bool have = t < (a.has_queue_len ? *a.queue_len : a.B);
        const int b = have ? (a.has_order ? a.order[t] : t) : 0;
        pbi = b;
        if (TUNED) {
            const mpcx_qp_tuning &tu = a.tune[b];
            P.w_perp = tu.w_perp; P.w_para = tu.w_para;
            P.R[0] = tu.R[0]; P.R[1] = tu.R[1]; P.Rd[0] = tu.Rd[0]; P.Rd[1] = tu.Rd[1];
            P.Q_v_yaw[0] = tu.Q_v_yaw[0]; P.Q_v_yaw[1] = tu.Q_v_yaw[1];
            P.Qf[0] = tu.Qf[0]; P.Qf[1] = tu.Qf[1]; P.Qf[2] = tu.Qf[2]; P.Qf[3] = tu.Qf[3];
            P.max_accel = tu.max_accel; P.max_decel = tu.max_decel; P.max_dsteer = tu.max_dsteer;
        }
        return have;
    }
};

template <int LQ, int SPL, bool TUNED, bool JERK>
__global__ __launch_bounds__(64, 1) void qp_quad_kernel(QpArgs a) {
    // the second-chance launch behind the condensed solver finds its list empty on (almost) every step: leave before the set-up of
    // eight empty lane groups and the first residual pass (that was 16 us per step at 4096 problems per GPU, the 8-GPU regime)
    if (a.has_queue_len && *a.queue_len == 0) return;
    __shared__ double sh[(3 * SPL * 8 + 2 * SPL) * 64];
    const int lane = threadIdx.x;
    GroupCx<LQ, SPL, JERK> cx{lane & (LQ - 1), lane, (lds_double *)sh};
    QueueSrc<LQ, SPL, TUNED> src{a};
#ifdef MPCX_STAGE_TRACE
    if (blockIdx.x == 0) cx.trace_buf = a.kkt + 4 * (size_t)a.B;
#endif
#ifdef MPCX_STAGE_PROFILE
    cx.occ = (unsigned char *)(a.kkt + 4 * (size_t)a.B + 16) + 128 * (size_t)blockIdx.x;      // dev build only: needs 16 + 16 * grid spare doubles behind kkt
    cx.life = (unsigned char *)(a.kkt + 4 * (size_t)a.B + 16 + 16 * 1024);                   // ... and B / 4 more behind those (grid <= 1024)
#endif
    mpcx_stage::solve_queue(cx, src);
#ifdef MPCX_STAGE_PROFILE
    if (lane == 0) for (int k = 0; k < 10; k++) atomicAdd((unsigned long long *)(a.kkt + 4 * (size_t)a.B) + k, cx.t_acc[k]);   // dev build only: needs 10 spare slots behind kkt
#endif
}

int qp_stage_grid(int B, int n_cu) {
    const int need = (B + 7) / 8;       // eight lane groups per wavefront
    // one wavefront per SIMD; MPCX_QP_GRID_DIV=d (dev aid: several shards in flight on separate streams, each on 1/d of the chip)
    static const int grid_div = [] { const char *e = getenv("MPCX_QP_GRID_DIV"); const int d = e ? atoi(e) : 1; return d >= 1 && d <= 16 ? d : 1; }();
    const int resident = n_cu * 4 / grid_div;
    return need < resident ? need : resident;
}

template <int LQ, int SPL, bool JERK>
void launch_qp_group(const QpArgs &a, hipStream_t st, int n_cu) {
    static_assert(LQ == 8, "qp_stage_grid counts eight lane groups per wavefront");
    const int grid = qp_stage_grid(a.B, n_cu);
    if (a.has_tune) hipLaunchKernelGGL((qp_quad_kernel<LQ, SPL, true, JERK>), dim3(grid), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((qp_quad_kernel<LQ, SPL, false, JERK>), dim3(grid), dim3(64), 0, st, a);
}

void launch_qp_stage(const QpArgs &a, hipStream_t st, int n_cu) {
    const int T = a.p.T;
    if (a.p.model == MPCX_MODEL_JERK5) {
        if (T <= 16) launch_qp_group<8, 2, true>(a, st, n_cu);
        else if (T <= 24) launch_qp_group<8, 3, true>(a, st, n_cu);
        else launch_qp_group<8, 4, true>(a, st, n_cu);
        return;
    }
    if (T <= 16) launch_qp_group<8, 2, false>(a, st, n_cu);
    else if (T <= 24) launch_qp_group<8, 3, false>(a, st, n_cu);
    else launch_qp_group<8, 4, false>(a, st, n_cu);
}

}  // namespace mpcx
