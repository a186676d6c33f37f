// mpcx_qp.hip -- batched linear-time-varying MPC quadratic program, one wavefront per instance.
//
// Replaces lib/mpc.py:138-208 `_linear_mpc_control` (cvxpy problem build + ECOS solve) together with
// lib/mpc.py:58-79 `_get_linear_model_matrix` and :129-135 `_get_xy_cost_mtx_for_orientation`
// (paths relative to /root/reference/main).  See DESIGN.md "QP kernel" for the derivation.
//
// Formulation.  delta_bar = 0 always (mpc.py:93), so v_t and yaw_t are exact cumulative sums of the inputs
// and only x,y couple accel and steer.  The states are eliminated (condensed QP in u = [a_0..a_{T-1},
// d_0..d_{T-1}], n = 2T <= 64 unknowns): lane i of the wavefront owns unknown i, row i of the Hessian and
// the four inequality rows attached to that unknown (a-lane k: a_k <= amax, -a_k <= -amin, +-v_{k+1} speed
// rows; d-lane k: +-d_k box, +-(d_{k+1}-d_k) rate rows).  Sensitivities have closed forms in prefix sums of
// the linearisation coefficients, the constraint matrix is never formed: G'DG is "diag + dt^2*min(S_i,S_k)"
// on the accel block (S = suffix sums of the speed-row weights) and tridiagonal on the steer block.
// Solver: Mehrotra predictor-corrector primal-dual interior point; per iteration one register-resident
// right-looking Cholesky (row i in lane i's VGPRs, column broadcast through a double-buffered LDS line,
// lane j keeps its (unscaled) Schur row as the transposed factor row, so both triangular sweeps run from
// registers) and two solves whose pivots travel by v_readlane.
#include "mpcx_common.h"
#include <cstdlib>
#include <cstring>

#ifndef MPCX_CHUNK
#define MPCX_CHUNK 40
#endif
#ifndef MPCX_SPLIT
#define MPCX_SPLIT 1
#endif
#ifndef MPCX_PRE
#define MPCX_PRE 8
#endif
#ifndef MPCX_BCH
#define MPCX_BCH 5
#endif
#ifndef MPCX_OCC
#define MPCX_OCC 1
#endif
#ifdef MPCX_STAGE_BARRIER
#define STAGE_BARRIER __builtin_amdgcn_sched_barrier(0)
#else
#define STAGE_BARRIER
#endif

// automatic choice: the stage-structured solver wins on throughput (8 problems per wavefront, O(T) work) once the batch fills
// the chip, the condensed solver on latency (one problem per wavefront: 0.08-0.26 ms per launch up to ~1000 problems against a
// 0.37-0.78 ms floor); measured crossover on the closed-loop benchmark workload (warm starts, mean 6.2 iterations with a tail to ~20: a small batch is bound by its
// SLOWEST problem, 41 us per iteration here against 9-18 us in the condensed kernel): ~11500 problems at T = 20.  With the trial pass
// (three quarters of the problems need one pass only) and the hardest-first queue the condensed kernel's work fell more than the
// stage solver's tail; launch times condensed / stage: 4096 problems 0.33 / 0.64 ms, 8192 0.51 / 0.68, 10240 0.61 / 0.68,
// 12288 0.71 / 0.68, 14336 0.83 / 0.70, 16384 0.92 / 0.71.  Beyond T = 20 the condensed kernel spills and is never competitive.
#ifndef MPCX_STAGE_MIN_BATCH
#define MPCX_STAGE_MIN_BATCH 11264
#endif

namespace mpcx {

typedef double double4_t __attribute__((ext_vector_type(4)));


template <int NT>
struct QpShared {
    static constexpr int N = 2 * NT;
    double H[N * N + WAVE]; // column-major: H[k*N + i] = H(i,k); lane i reads/writes its own row conflict-free;
                            // the last WAVE slots are a write sink for lanes that have no band entry to patch
    double pre[6][NT + 1];  // exclusive prefix sums over t: Px, Py (accel paths), Qx, Qy (steer paths), Cx, Cy (affine)
    double beta[32];        // dt * vbar_k / L
    double wt[33][6];       // per-t cost weights: wxx, wxy, wyy, wv, wyaw (padded to 48 B so rows stay 16-B aligned)
    double we[NT + 1][4];   // W_t * (free response - reference)
    double wf[33][4][4];    // per-t factor rows F_t (F_t'F_t = W_t): [t][component][qx, qy, qv, qyaw]
    double cb[2][WAVE];     // factorisation column broadcast (double buffered)
    double ub[WAVE];        // current iterate broadcast
    double sb[WAVE];        // speed-row suffix sums broadcast
};

// All hot loops below are written branch-free (selects / 0-1 masks): a single straight-line block keeps the
// register allocator out of scratch.  The block is ONE wavefront, so LDS traffic between lanes is ordered by
// the in-order LDS queue; `lds_sync()` only stops the compiler from reordering across it.
// compiler-only ordering point between LDS accesses of different lanes (no scheduling barrier)
__device__ __forceinline__ void lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_sched_barrier(0);      // keep the (huge, fully unrolled) blocks from being reordered wholesale
}

// Optimisation barrier: the value must exist in a VGPR here.  Without it LLVM sinks the ~10^3 accumulator
// updates of the unrolled loops below their last use and spills every operand to scratch.
__device__ __forceinline__ void pin(double &v) { asm volatile("" : "+v"(v)); }

// Solve (L~ D L~') x = b.  Lane i holds Rlo[k] = L~(i,k) for k<i (0 for k>=i) and Rup[k] = the UNSCALED Schur-complement
// entry M_i(i,k) = L~(k,i)*d_i for k>i (0 for k<=i) -- the row the lane held when its own column was eliminated -- and
// dinv = 1/d_i.  With the zero fills neither sweep can disturb a finished lane, so no per-step capture is needed; scaling
// the transposed part in place would cost a select per element or lose digits, the backward sweep applies dinv instead.
// Pivots travel by v_readlane.
template <int N>
__device__ __forceinline__ double ldl_solve(const double (&Rlo)[N], const double (&Rup)[N], double dinv, double b, int lane) {
    double acc = b;
#pragma unroll
    for (int j = 0; j < N; j++) acc = fma(-Rlo[j], rdlane(acc, j), acc);
    const double w = acc * dinv;
    double sum = 0.0;
#pragma unroll
    for (int j = N - 1; j >= 0; j--) sum = fma(Rup[j], rdlane(fma(-dinv, sum, w), j), sum);
    return fma(-dinv, sum, w);
}

// Single-array variant (MPCX_SPLIT=0): lane i keeps L~(i,k) for k<i and the unscaled Schur entries for k>i in ONE array (80
// VGPRs less); a finished lane's accumulators are then disturbed by later steps, so each sweep captures its result.
template <int N>
__device__ __forceinline__ double ldl_solve1(const double (&R)[N], double dinv, double b, int lane) {
    double acc = b, z = 0.0;
#pragma unroll
    for (int j = 0; j < N; j++) {
        const double zj = rdlane(acc, j);
        z = (lane == j) ? acc : z;
        acc = fma(-R[j], zj, acc);
    }
    const double w = z * dinv;
    double sum = 0.0, x = 0.0;
#pragma unroll
    for (int j = N - 1; j >= 0; j--) {
        const double tmp = fma(-dinv, sum, w);
        const double xj = rdlane(tmp, j);
        x = (lane == j) ? tmp : x;
        sum = fma(R[j], xj, sum);
    }
    return x;
}

template <int NT>
__global__ __launch_bounds__(64, MPCX_OCC) void qp_kernel(QpArgs a) {
    constexpr int N = 2 * NT;
    constexpr int CHUNK = MPCX_CHUNK;      // LDS values staged per batch in the unrolled loops
    constexpr int PRE = (N - 1 < MPCX_PRE) ? N - 1 : MPCX_PRE;   // column entries prefetched one column ahead in the factorisation
    constexpr int BCH = MPCX_BCH;          // same, stage pairs in the Hessian build (8 doubles each)
    __shared__ QpShared<NT> sh;
  for (int guard = 0; guard <= a.B; guard++) {      // every wavefront leaves after at most B+1 tickets (bounded by construction)
    // the lane index is made opaque once per problem: everything derived from it (table indices, row masks, tile maps: ~40 values) is
    // then recomputed per problem -- a few dozen integer instructions -- instead of being hoisted out of the work loop, where it
    // stayed live across the whole solve and was spilled (164 B/lane of scratch in the T = 20 build, written by every wavefront)
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    int b = 0;
    if (lane == 0) b = atomicAdd(a.ticket, 1);
    b = __builtin_amdgcn_readfirstlane(b);
    if (b >= a.B) break;
    if (a.has_order) b = a.order[b];      // hardest first (mpcx_qp_set_order_hint): a wavefront that draws a long problem draws nothing
                                          // else while the short ones are shared out among the others
    lds_sync();                       // the previous problem's LDS reads are done before this one overwrites the tables
    mpcx_mpc_params P = a.p;
    if (a.has_tune) {                 // wave-uniform row (b comes from readfirstlane): scalar loads
        const mpcx_qp_tuning &tu = a.tune[b];
        P.w_perp = tu.w_perp; P.w_para = tu.w_para;
        P.R[0] = tu.R[0]; P.R[1] = tu.R[1]; P.Rd[0] = tu.Rd[0]; P.Rd[1] = tu.Rd[1];
        P.Q_v_yaw[0] = tu.Q_v_yaw[0]; P.Q_v_yaw[1] = tu.Q_v_yaw[1];
        P.Qf[0] = tu.Qf[0]; P.Qf[1] = tu.Qf[1]; P.Qf[2] = tu.Qf[2]; P.Qf[3] = tu.Qf[3];
        P.max_accel = tu.max_accel; P.max_decel = tu.max_decel; P.max_dsteer = tu.max_dsteer;
    }
    const int T = P.T, W = T + 1;
    const double dt = P.dt;

    const double *xref = a.xref + (size_t)b * 4 * W;
    const double *xbar = a.xbar + (size_t)b * 4 * W;
    const uint8_t *re = a.re + (size_t)b * W;
    const double x0 = a.x0[4 * b + 0], y0 = a.x0[4 * b + 1], v0 = a.x0[4 * b + 2], yaw0 = a.x0[4 * b + 3];

    const int kind = lane >= NT;            // 0: accel unknown, 1: steer unknown
    const int k = lane - kind * NT;         // stage of this unknown
    const bool inrow = lane < N;
    const bool real = inrow && (k < T);
    const int kc = real ? k : 0;            // clamped stage for table reads
    const int li = inrow ? lane : N - 1;    // clamped row for H reads

    // ---------------------------------------------------------------- per-stage linearisation + weights
    // lanes [0,32): dynamics at xbar[:,t]; lanes [32,64): cost rotation at xref[3,t']   (one sincos for both)
    {
        const bool dyn = lane < 32;
        const int t = dyn ? lane : (lane - 32 + 1);      // dynamics stage 0..31 | cost stage 1..32
        const bool on = dyn ? (t < T) : (t <= T);
        const int tc = on ? t : 0;
        const double vb = (dyn && on) ? xbar[2 * W + tc] : 0.0;
        const double ang = on ? (dyn ? xbar[3 * W + tc] : xref[3 * W + tc]) : 0.0;
        const bool ended = (!dyn && on) ? (re[tc] != 0) : false;
        double s, c;
        sincos(ang, &s, &c);
        // mpc.py:58-79 with delta = 0:  A[0,2]=dt c, A[0,3]=-dt v s, A[1,2]=dt s, A[1,3]=dt v c,
        //                               C[0]=dt v s phi, C[1]=-dt v c phi, B[3,1]=dt v / L
        const double m = (dyn && on) ? 1.0 : 0.0;
        double vals[6] = {m * (dt * c), m * (dt * s), m * (-dt * vb * s), m * (dt * vb * c),
                          m * (dt * vb * s * ang), m * (-dt * vb * c * ang)};
        if (dyn) sh.beta[t] = m * (dt * vb / P.L);
#pragma unroll
        for (int r = 0; r < 6; r++) {
            double v = scan_up32(vals[r]);               // DPP scan over lanes 0..31 (lanes >= 32 carry zeros)
            if (dyn && t < NT) sh.pre[r][t + 1] = v;
            if (lane == 0) sh.pre[r][0] = 0.0;
        }
        if (!dyn) {
            // mpc.py:160-170; cos(psi+pi/2) = -sin(psi), sin(psi+pi/2) = cos(psi)
            double wxx = (s * s) * P.w_perp + (c * c) * P.w_para;
            double wxy = (-s * c) * P.w_perp + (c * s) * P.w_para;
            double wyy = (c * c) * P.w_perp + (s * s) * P.w_para;
            double wv = P.Q_v_yaw[0], wp = P.Q_v_yaw[1];
            if (ended) { wxx = P.Qf[0]; wxy = 0.0; wyy = P.Qf[1]; wv = P.Qf[2]; wp = P.Qf[3]; }
            const double mm = on ? 1.0 : 0.0;
            sh.wt[t][0] = mm * wxx; sh.wt[t][1] = mm * wxy; sh.wt[t][2] = mm * wyy; sh.wt[t][3] = mm * wv; sh.wt[t][4] = mm * wp;
            // factor rows: along-track, cross-track, speed, yaw (terminal stages: axis-aligned sqrt(Qf))
            const double ra = ended ? sqrt(P.Qf[0]) : sqrt(P.w_para), rc = ended ? sqrt(P.Qf[1]) : sqrt(P.w_perp);
            const double rv = mm * sqrt(wv), rp = mm * sqrt(wp);
            const double cc = ended ? 1.0 : c, ss = ended ? 0.0 : s;
            sh.wf[t][0][0] = mm * ra * cc;  sh.wf[t][0][1] = mm * ra * ss;  sh.wf[t][0][2] = 0.0; sh.wf[t][0][3] = 0.0;
            sh.wf[t][1][0] = -mm * rc * ss; sh.wf[t][1][1] = mm * rc * cc;  sh.wf[t][1][2] = 0.0; sh.wf[t][1][3] = 0.0;
            sh.wf[t][2][0] = 0.0; sh.wf[t][2][1] = 0.0; sh.wf[t][2][2] = rv;  sh.wf[t][2][3] = 0.0;
            sh.wf[t][3][0] = 0.0; sh.wf[t][3][1] = 0.0; sh.wf[t][3][2] = 0.0; sh.wf[t][3][3] = rp;
        }
        if (lane == 0) { sh.wt[0][0] = 0; sh.wt[0][1] = 0; sh.wt[0][2] = 0; sh.wt[0][3] = 0; sh.wt[0][4] = 0; }
    }
    lds_sync();
    // W_t (free response - reference): the free response has v = v0, yaw = yaw0 at every stage
    if (lane <= NT) {
        const int t = lane;
        const bool on = (t >= 1 && t <= T);
        const int tc = on ? t : 0;
        const double xf = x0 + v0 * sh.pre[0][tc] + yaw0 * sh.pre[2][tc] + sh.pre[4][tc];
        const double yf = y0 + v0 * sh.pre[1][tc] + yaw0 * sh.pre[3][tc] + sh.pre[5][tc];
        const double e0 = xf - xref[0 * W + tc], e1 = yf - xref[1 * W + tc];
        const double e2 = v0 - xref[2 * W + tc], e3 = yaw0 - xref[3 * W + tc];
        sh.we[t][0] = sh.wt[t][0] * e0 + sh.wt[t][1] * e1;     // weights are zero when !on
        sh.we[t][1] = sh.wt[t][1] * e0 + sh.wt[t][2] * e1;
        sh.we[t][2] = sh.wt[t][3] * e2;
        sh.we[t][3] = sh.wt[t][4] * e3;
    }
    lds_sync();

    // ---------------------------------------------------------------- gradient (lane = own unknown, plain VALU)
    double g = 0.0;
    {
        const double coef = real ? (kind ? sh.beta[kc] : dt) : 0.0;
        const int rx = kind ? 2 : 0, ry = kind ? 3 : 1;          // LDS row indices (no generic pointers)
        const double bx0 = sh.pre[rx][kc + 1], by0 = sh.pre[ry][kc + 1];
        const double cv = kind ? 0.0 : coef, cp = kind ? coef : 0.0;
#pragma unroll
        for (int t = 1; t <= NT; t++) {
            const double act = (real && (t >= k + 1)) ? 1.0 : 0.0;     // stages t > T carry zero weights
            const double ca = act * coef;
            const double gx = ca * (sh.pre[rx][t] - bx0), gy = ca * (sh.pre[ry][t] - by0), gv = act * cv, gp = act * cp;
            g += gx * sh.we[t][0] + gy * sh.we[t][1] + gv * sh.we[t][2] + gp * sh.we[t][3];
        }
        g *= 2.0;
    }
    // ---------------------------------------------------------------- condensed Hessian H = 2 * sum_t G_t' G_t  on the
    // matrix cores.  G_t (4 x n) = F_t * [sensitivities of every unknown at stage t], F_t'F_t = W_t (rotation into the path
    // frame scaled by sqrt of the along-/cross-track weights, or sqrt(Qf) on the terminal stages).  One v_mfma_f64_16x16x4
    // per 16x16 output tile and stage: lane l supplies component c = l>>4 of unknown 16*tile + (l&15) as both the A and the
    // B operand, the 4-deep contraction runs over the state components.  (FP64 MFMA and FP64 VALU peak at the same rate on
    // MI355X; the gain is that 1024 FMAs retire per issued instruction and the VALU stays free for the operands.)
    {
        constexpr int NTILE = (N + 15) / 16;
        const int cmp = lane >> 4;                                   // state component carried by this lane
        double4_t acc[NTILE][NTILE];
#pragma unroll
        for (int I = 0; I < NTILE; I++)
#pragma unroll
            for (int J = 0; J <= I; J++) acc[I][J] = double4_t{0.0, 0.0, 0.0, 0.0};
        double tc[NTILE], tb0x[NTILE], tb0y[NTILE];                  // per tile: coefficient and prefix offsets of "my" unknown
        int tk[NTILE], trx[NTILE];
        bool tkind[NTILE], treal[NTILE];
#pragma unroll
        for (int I = 0; I < NTILE; I++) {
            const int var = 16 * I + (lane & 15);
            tkind[I] = var >= NT;
            tk[I] = var - (tkind[I] ? NT : 0);
            treal[I] = (var < N) && (tk[I] < T);
            const int kcl = treal[I] ? tk[I] : 0;
            tc[I] = treal[I] ? (tkind[I] ? sh.beta[kcl] : dt) : 0.0;
            trx[I] = tkind[I] ? 2 : 0;
            tb0x[I] = sh.pre[trx[I]][kcl + 1];
            tb0y[I] = sh.pre[trx[I] + 1][kcl + 1];
        }
#pragma unroll
        for (int t = 1; t <= NT; t++) {
            const double qx = sh.wf[t][cmp][0], qy = sh.wf[t][cmp][1], qv = sh.wf[t][cmp][2], qp = sh.wf[t][cmp][3];
            double gt[NTILE];
#pragma unroll
            for (int I = 0; I < NTILE; I++) {
                const bool on = treal[I] && (t >= tk[I] + 1);
                const double dx = sh.pre[trx[I]][t] - tb0x[I], dy = sh.pre[trx[I] + 1][t] - tb0y[I];
                const double lin = qx * dx + qy * dy + (tkind[I] ? qp : qv);
                gt[I] = on ? tc[I] * lin : 0.0;
            }
#pragma unroll
            for (int I = 0; I < NTILE; I++)
#pragma unroll
                for (int J = 0; J <= I; J++) acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(gt[I], gt[J], acc[I][J], 0, 0, 0);
        }
        // C/D layout: register r of lane l is element (row = (l>>4) + 4r, col = l&15) of the tile.  H is symmetric: element
        // (row, col) is stored at both [row*N + col] and [col*N + row] of the column-major LDS copy.
#pragma unroll
        for (int I = 0; I < NTILE; I++)
#pragma unroll
            for (int J = 0; J <= I; J++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * I + cmp + 4 * r, col = 16 * J + (lane & 15);
                    const double v = 2.0 * acc[I][J][r];
                    if (row < N && col < N) {
                        sh.H[row * N + col] = v;
                        if (I != J) sh.H[col * N + row] = v;
                    }
                }
    }
    lds_sync();
    {
        // input cost (mpc.py:177-180) and input-rate cost (mpc.py:182-183); objective has no 1/2 => H = 2*(...)
        const double rr = re[kc] ? P.R_end[kind] : P.R[kind];
        const double rdc = P.Rd[kind];
        const int nb = (k >= 1) + (k + 1 < T);
        const double dg = real ? (2.0 * rr + 2.0 * rdc * nb) : 1.0;     // padding unknowns: identity row
        const double off_lo = (real && k >= 1) ? -2.0 * rdc : 0.0;
        const double off_hi = (real && k + 1 < T) ? -2.0 * rdc : 0.0;
        if (inrow) {
            sh.H[lane * N + lane] += dg;
            if (lane + 1 < N) sh.H[(lane + 1) * N + lane] += off_hi;
            if (lane >= 1) sh.H[(lane - 1) * N + lane] += off_lo;
        }
    }
    lds_sync();
    // pristine band entries H(i,i), H(i,i+1), H(i,i-1) of this lane's row: each iteration writes "band + G'DG band"
    // into LDS so that the row can be read back without per-element selects
    const bool has_hi = lane + 1 < N, has_lo = inrow && lane >= 1;
    const int i_d = inrow ? lane * N + lane : N * N + lane;           // LDS indices (no generic pointers)
    const int i_hi = has_hi ? (lane + 1) * N + lane : N * N + lane;
    const int i_lo = has_lo ? (lane - 1) * N + lane : N * N + lane;
    const double h_d = inrow ? sh.H[i_d] : 0.0, h_hi = has_hi ? sh.H[i_hi] : 0.0, h_lo = has_lo ? sh.H[i_lo] : 0.0;

    // ---------------------------------------------------------------- constraints owned by this lane
    // rows 0/1: +-u_i box ; rows 2/3: accel lane k -> +-v_{k+1} speed rows, steer lane k -> +-(d_{k+1}-d_k)
    const bool val01 = real;
    const bool val23 = real && (kind == 0 || k + 1 < T);
    const double h0 = kind ? P.max_steer : P.max_accel;
    const double h1 = kind ? P.max_steer : -P.max_decel;
    const double h2 = kind ? P.max_dsteer * dt : P.max_speed - v0;
    const double h3 = kind ? P.max_dsteer * dt : v0 - P.min_speed;
    const double m01 = val01 ? 1.0 : 0.0, m23 = val23 ? 1.0 : 0.0;
    const double minv = 1.0 / (double)(8 * T - 2);
    const double ma = (kind == 0 && real) ? 1.0 : 0.0;     // accel lane mask
    const double k0m = (k == 0) ? 0.0 : 1.0;

    double u = 0.0;
    if (real && a.has_warm) u = a.u_warm[(size_t)b * 2 * T + kind * T + k];

    // (G x) of the second row pair: accel lane: dt * sum_{j<=k} a_j ; steer lane: d_{k+1} - d_k   (DPP, no LDS)
    auto second_rows = [&](double x) -> double {
        const double pa = scan_up32(ma * x);
        const double nx = lane_next(x);
        return m23 * (kind == 0 ? dt * pa : (nx - x));
    };
    // (G' w) for this unknown from the lane's w0..w3 (already masked)
    auto gt_apply = [&](double w0, double w1, double w2, double w3) -> double {
        const double q = w2 - w3;
        const double sa = scan_down32(ma * q, lane);            // sum_{j>=k} q_j over accel lanes
        const double qp = k0m * lane_prev(q);                   // q_{k-1}
        return (w0 - w1) + (kind == 0 ? dt * sa : (qp - q));
    };

    double R[N];            // row `lane` of M, then of its factor (see ldl_solve)
    double s0, s1, s2, s3, l0 = MPCX_LAM0, l1 = MPCX_LAM0, l2 = MPCX_LAM0, l3 = MPCX_LAM0;
    {
        const double e2 = second_rows(u);
        s0 = fmax(h0 - u, MPCX_SLACK_FLOOR); s1 = fmax(h1 + u, MPCX_SLACK_FLOOR); s2 = fmax(h2 - e2, MPCX_SLACK_FLOOR); s3 = fmax(h3 + e2, MPCX_SLACK_FLOOR);
    }
    const double gnorm = fmax(1.0, wave_max_dpp(real ? fabs(g) : 0.0));
    const double hnorm = fmax(1.0, wave_max_dpp(fmax(m01 * fmax(fabs(h0), fabs(h1)), m23 * fmax(fabs(h2), fabs(h3)))));
    const double ign = 1.0 / gnorm, ihn = 1.0 / hnorm;

    int status = MPCX_QP_MAXITER;
    int it = 0, loose_run = 0;
    double mu = 0, rd = 0, rp0 = 0, rp1 = 0, rp2 = 0, rp3 = 0;
    const double dt2 = dt * dt;
    const double tol_loose = P.tol > 1e-7 ? P.tol : 1e-7;
    // x[2,0] rows of mpc.py:187-188 are constants: outside the speed interval the problem is infeasible
    const bool feasible0 = !(v0 > P.max_speed + 1e-9 || v0 < P.min_speed - 1e-9);
    if (!feasible0) status = MPCX_QP_INFEASIBLE;
    const int max_iter = feasible0 ? P.max_iter : -1;
    const double rowm = inrow ? 1.0 : 0.0;
    // trial step (see mpcx_qp_stage.h): the first pass runs with every multiplier taken as zero, so that M = H and the
    // predictor direction leads to the unconstrained minimiser; if that point violates no row it is the solution (0 iterations)
    bool trial = feasible0 && MPCX_TRIAL_STEP != 0, accepted = false;
    // active-set polish (see MPCX_POLISH in mpcx_qp_stage.h): a pass like the trial pass, with weight rho on the rows of the active set
    // (pm0..pm3: this lane's four rows) and lam_e + rho gap as their linear term; accepted if its end point is a KKT point
    bool polish = false, pol_pred = false, pm0 = false, pm1 = false, pm2 = false, pm3 = false;
    int ptries = 0, pend = 0, ptested = -1;
    for (it = 0; it <= max_iter; it++) {
        // -------- H u from the PRISTINE Hessian (restore the band entries the previous iteration patched)
        sh.H[i_d] = h_d; sh.H[i_hi] = h_hi; sh.H[i_lo] = h_lo;
        sh.ub[lane] = u;
        lds_sync();
        double hu = 0.0;
#pragma unroll
        for (int j = 0; j < N; j++) hu = fma(sh.H[j * N + li], sh.ub[j], hu);
        // -------- residuals
        double resn, resd_n, resp_n;
        {
            const double e2 = second_rows(u);
            const double t0 = trial ? 0.0 : l0, t1 = trial ? 0.0 : l1, t2 = trial ? 0.0 : l2, t3 = trial ? 0.0 : l3;
            rd = m01 * (hu + g + gt_apply(m01 * t0, m01 * t1, m23 * t2, m23 * t3));
            rp0 = m01 * (u + s0 - h0); rp1 = m01 * (-u + s1 - h1);
            rp2 = m23 * (e2 + s2 - h2); rp3 = m23 * (-e2 + s3 - h3);
            mu = wave_sum_dpp(m01 * (s0 * t0 + s1 * t1) + m23 * (s2 * t2 + s3 * t3)) * minv;
            resd_n = wave_max_dpp(fabs(rd) * ign);
            resp_n = wave_max_dpp(fmax(fmax(fabs(rp0), fabs(rp1)), fmax(fabs(rp2), fabs(rp3))) * ihn);
            resn = fmax(resd_n, resp_n);
        }
        if (accepted) { status = MPCX_QP_OPTIMAL; break; }      // the trial / polish point, with its residuals measured above for the report
        // reduced-accuracy acceptance when the iteration cannot continue (the reference accepts ECOS's OPTIMAL_INACCURATE, mpc.py:196)
        const bool loose = !trial && resn <= tol_loose && mu <= tol_loose;
        if (!trial && !polish) {
            if (MPCX_POLISH != 0 && pol_pred && ptested != it) {     // the step that led here predicted a point close enough to polish
                polish = true; ptries = 0; pend = 0;
            } else {                                 // the exit tests, once per iterate
                const bool conv = resn <= P.tol && mu <= P.tol;
                // stagnation exit: the stationarity residual of badly conditioned instances stalls at its rounding floor while mu keeps
                // collapsing; after 4 consecutive reduced-accuracy iterates stop before the factorisation degrades them
                loose_run = loose ? loose_run + 1 : 0;
                const bool stop_ok = conv || loose_run >= 4 || (it == max_iter && loose);
                const bool stop_fail = it == max_iter && !loose;
                if (MPCX_POLISH != 0 && (stop_ok || stop_fail) && ptested != it) { polish = true; ptries = 0; pend = stop_ok ? 1 : 2; }
                else if (stop_ok) { status = MPCX_QP_OPTIMAL; break; }
                else if (stop_fail) break;
            }
            if (polish) { pm0 = val01 && s0 < l0; pm1 = val01 && s1 < l1; pm2 = val23 && s2 < l2; pm3 = val23 && s3 < l3; }
        }

        // -------- row of M = H + G'DG: the band part (diagonal + steer tridiagonal) is written into LDS so that the row
        // reads back without per-element selects; the accel block adds dt^2 * min(S_i, S_j)
        double S;
        {
            const double is0 = frcp1(s0), is1 = frcp1(s1), is2 = frcp1(s2), is3 = frcp1(s3);
            const double t0 = trial ? 0.0 : l0, t1 = trial ? 0.0 : l1, t2 = trial ? 0.0 : l2, t3 = trial ? 0.0 : l3;
            const double r23 = polish ? ((pm2 ? MPCX_POLISH_RHO : 0.0) + (pm3 ? MPCX_POLISH_RHO : 0.0)) : m23 * (t2 * is2 + t3 * is3);
            S = scan_down32(ma * r23, lane);             // accel lanes: sum_{j>=k} (d2+d3)_j ; elsewhere 0
            const double r_own = kind ? r23 : 0.0;
            const double r_prev = k0m * lane_prev(r_own);
            sh.sb[lane] = S;
            const double r01 = polish ? ((pm0 ? MPCX_POLISH_RHO : 0.0) + (pm1 ? MPCX_POLISH_RHO : 0.0)) : m01 * (t0 * is0 + t1 * is1);
            sh.H[i_d] = h_d + (r01 + r_own + r_prev);
            sh.H[i_hi] = h_hi - r_own;
            sh.H[i_lo] = h_lo - r_prev;
        }
        lds_sync();
#pragma unroll
        for (int j = 0; j < N; j++) {
            double v = rowm * sh.H[j * N + li];
            if (j < NT) v = fma(dt2, fmin(S, sh.sb[j]), v);
            R[j] = v;
        }

        // -------- M = L~ D L~' (right-looking; see file header)
        // Software pipeline per column j (one straight-line region each):
        //  * the pivots form the only serial chain: d_{j+1} = M_j(j+1,j+1) - M_j(j+1,j)^2 / d_j uses lane j+1's OWN two
        //    entries (v_readlane), so 1/d_{j+1} is computed while column j's trailing update issues;
        //  * column j+1 is published to LDS right after its first update and its first PRE entries are read back
        //    immediately, so the next column starts without an exposed LDS round trip.
        double dinv = 1.0;
        bool bad = false;
        double pre[2][PRE];
#if MPCX_SPLIT
        double Rlo[N];
#define SOLVE(b_) ldl_solve<N>(Rlo, R, dinv, b_, lane)
#else
#define SOLVE(b_) ldl_solve1<N>(R, dinv, b_, lane)
#endif
        sh.cb[0][lane] = R[0];
        lds_order();
#pragma unroll
        for (int q = 0; q < PRE; q++) pre[0][q] = (1 + q < N) ? sh.cb[0][1 + q] : 0.0;
        double rinv;
        {
            const double d0 = rdlane(R[0], 0);
            bad = !(d0 > 0.0);
            rinv = frcp(bad ? 1.0 : d0);
        }
#pragma unroll
        for (int j = 0; j < N; j++) {
            constexpr int dummy = 0; (void)dummy;
            const int p = j & 1;
            const double tj = (lane > j) ? R[j] * rinv : 0.0;          // L~(lane, j)
            double rinv_n = 1.0;
            if (j + 1 < N) {
                const double a1 = rdlane(R[j], j + 1), d1 = rdlane(R[j + 1], j + 1);
                const double dn = fma(-(a1 * rinv), a1, d1);            // next pivot (wave-uniform)
                bad = bad || !(dn > 0.0);
                rinv_n = frcp((dn > 0.0) ? dn : 1.0);                  // full accuracy: errors here are amplified by cond(M)
                R[j + 1] = fma(-tj, pre[p][0], R[j + 1]);
                sh.cb[p ^ 1][lane] = R[j + 1];
                lds_order();
#pragma unroll
                for (int q = 0; q < PRE; q++) pre[p ^ 1][q] = (j + 2 + q < N) ? sh.cb[p ^ 1][j + 2 + q] : 0.0;
            }
            {
                // loads first, then FMAs, then the (volatile) pins: a pin between two loads would serialise them
                double cv[N];
#pragma unroll
                for (int kk = j + 2; kk < N; kk++) cv[kk] = (kk - (j + 1) < PRE) ? pre[p][kk - (j + 1)] : sh.cb[p][kk];
#pragma unroll
                for (int kk = j + 2; kk < N; kk++) R[kk] = fma(-tj, cv[kk], R[kk]);
#pragma unroll
                for (int kk = j + 2; kk < N; kk++) pin(R[kk]);
            }
#if MPCX_SPLIT
            Rlo[j] = tj;                                                // unit lower factor entry (0 for lanes <= j)
            R[j] = (lane < j) ? R[j] : 0.0;                             // lanes < j keep their unscaled Schur entry
#else
            R[j] = (lane < j) ? R[j] : tj;                              // lanes < j keep their unscaled Schur entry, lane j gets 0
#endif
            dinv = (lane == j) ? rinv : dinv;
            rinv = rinv_n;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (bad && trial) { trial = false; it--; continue; }    // H itself did not factorise: no trial, the iteration decides
        if (bad && polish) {                                    // counts as a rejected polish pass
            ptries++;
            if (ptries >= MPCX_POLISH_TRIES) {
                polish = false; ptested = it;
                if (pend == 1) { status = MPCX_QP_OPTIMAL; break; }
                if (pend == 2) break;
            }
            it--; continue;
        }
        if (bad) { status = loose ? MPCX_QP_OPTIMAL : MPCX_QP_NUMERIC; break; }

        // Everything derived from (u, s, lam) is recomputed here instead of being kept alive across the factorisation (the
        // pins make the compiler treat the inputs as new values): 80 VGPRs of the register file hold the factor.
        pin(u); pin(s0); pin(s1); pin(s2); pin(s3); pin(l0); pin(l1); pin(l2); pin(l3);
        const double is0 = frcp1(s0), is1 = frcp1(s1), is2 = frcp1(s2), is3 = frcp1(s3);
        const double t0 = trial ? 0.0 : l0, t1 = trial ? 0.0 : l1, t2 = trial ? 0.0 : l2, t3 = trial ? 0.0 : l3;
        const double d0 = m01 * t0 * is0, d1 = m01 * t1 * is1, d2 = m23 * t2 * is2, d3 = m23 * t3 * is3;
        const double e2u = second_rows(u);
        {
            const double e2 = e2u;
            rp0 = m01 * (u + s0 - h0); rp1 = m01 * (-u + s1 - h1);
            rp2 = m23 * (e2 + s2 - h2); rp3 = m23 * (-e2 + s3 - h3);
        }
        // -------- predictor (affine scaling) direction
        double w0 = -m01 * t0 + d0 * rp0, w1 = -m01 * t1 + d1 * rp1, w2 = -m23 * t2 + d2 * rp2, w3 = -m23 * t3 + d3 * rp3;
        // multiplier estimates of the augmented-Lagrangian solve: the iterate's
        const double e0 = l0, e1 = l1, e2l = l2, e3 = l3;
        if (polish) {       // rd above carries G' lam: take it out, put the active rows' lam_e + rho gap in (gap = rp - s)
            w0 = -m01 * l0 + (pm0 ? e0 + MPCX_POLISH_RHO * (rp0 - s0) : 0.0); w1 = -m01 * l1 + (pm1 ? e1 + MPCX_POLISH_RHO * (rp1 - s1) : 0.0);
            w2 = -m23 * l2 + (pm2 ? e2l + MPCX_POLISH_RHO * (rp2 - s2) : 0.0); w3 = -m23 * l3 + (pm3 ? e3 + MPCX_POLISH_RHO * (rp3 - s3) : 0.0);
        }
        double rhs = m01 * (-rd - gt_apply(w0, w1, w2, w3));
        double du = m01 * SOLVE(rhs);
        double f2 = second_rows(du);
        if (polish) {
            const double un = u + du, en = e2u + f2;
            const double g0 = un - h0, g1 = -un - h1, g2 = en - h2, g3 = -en - h3;          // row gaps at the end point
            const double n0 = e0 + MPCX_POLISH_RHO * g0, n1 = e1 + MPCX_POLISH_RHO * g1, n2 = e2l + MPCX_POLISH_RHO * g2, n3 = e3 + MPCX_POLISH_RHO * g3;
            const bool ng0 = pm0 && n0 < -MPCX_POLISH_EPS_L, ng1 = pm1 && n1 < -MPCX_POLISH_EPS_L, ng2 = pm2 && n2 < -MPCX_POLISH_EPS_L, ng3 = pm3 && n3 < -MPCX_POLISH_EPS_L;
            const bool vi0 = !pm0 && val01 && g0 > MPCX_POLISH_EPS_G, vi1 = !pm1 && val01 && g1 > MPCX_POLISH_EPS_G;
            const bool vi2 = !pm2 && val23 && g2 > MPCX_POLISH_EPS_G, vi3 = !pm3 && val23 && g3 > MPCX_POLISH_EPS_G;
            if (__ballot(ng0 || ng1 || ng2 || ng3 || vi0 || vi1 || vi2 || vi3) == 0ull) {
                u = un;
                s0 = fmax(-g0, 1e-30); s1 = fmax(-g1, 1e-30); s2 = fmax(-g2, 1e-30); s3 = fmax(-g3, 1e-30);
                l0 = pm0 ? fmax(n0, 0.0) : 0.0; l1 = pm1 ? fmax(n1, 0.0) : 0.0; l2 = pm2 ? fmax(n2, 0.0) : 0.0; l3 = pm3 ? fmax(n3, 0.0) : 0.0;
                accepted = true; polish = false;
            } else {
                pm0 = (pm0 || vi0) && !ng0; pm1 = (pm1 || vi1) && !ng1; pm2 = (pm2 || vi2) && !ng2; pm3 = (pm3 || vi3) && !ng3;
                ptries++;
                if (ptries >= MPCX_POLISH_TRIES) {
                    polish = false; ptested = it;       // pend == 0: the iterate now takes its exit tests and the iteration goes on
                    if (pend == 1) { status = MPCX_QP_OPTIMAL; break; }
                    if (pend == 2) break;
                }
            }
            it--;                                   // not an iteration
            continue;
        }
        if (trial) {
            // the four rows of this lane at u + du
            const double un = u + du, en = e2u + f2;
            const bool viol = (val01 && (!(un - h0 <= 0.0) || !(-un - h1 <= 0.0))) || (val23 && (!(en - h2 <= 0.0) || !(-en - h3 <= 0.0)));
            trial = false;
            it--;                                   // this pass was not an iteration
            if (__ballot(viol) == 0ull) {
                u = un;
                s0 = fmax(h0 - un, 1e-30); s1 = fmax(h1 + un, 1e-30); s2 = fmax(h2 - en, 1e-30); s3 = fmax(h3 + en, 1e-30);
                l0 = l1 = l2 = l3 = 0.0;
                accepted = true;
            }
            continue;
        }
        const double dsa0 = -rp0 - m01 * du, dsa1 = -rp1 + m01 * du, dsa2 = -rp2 - f2, dsa3 = -rp3 + f2;
        const double dla0 = m01 * (-l0 - d0 * dsa0), dla1 = m01 * (-l1 - d1 * dsa1);
        const double dla2 = m23 * (-l2 - d2 * dsa2), dla3 = m23 * (-l3 - d3 * dsa3);
        const double il0 = frcp1(l0), il1 = frcp1(l1), il2 = frcp1(l2), il3 = frcp1(l3);
        // largest step keeping s, lam >= 0: 1 / max(-ds/s, -dlam/lam)
        double rat = fmax(fmax(-dsa0 * is0, -dla0 * il0), fmax(-dsa1 * is1, -dla1 * il1));
        rat = fmax(rat, fmax(fmax(-dsa2 * is2, -dla2 * il2), fmax(-dsa3 * is3, -dla3 * il3)));
        rat = wave_max_dpp(rat);
        const double al = (rat > 1.0) ? frcp(rat) : 1.0;
        double mu_aff = m01 * ((s0 + al * dsa0) * (l0 + al * dla0) + (s1 + al * dsa1) * (l1 + al * dla1)) +
                        m23 * ((s2 + al * dsa2) * (l2 + al * dla2) + (s3 + al * dsa3) * (l3 + al * dla3));
        mu_aff = wave_sum_dpp(mu_aff) * minv;
        double sigma = mu_aff * frcp(mu);
        sigma = sigma * sigma * sigma;
        const double smu = fmax(sigma * mu, 0.1 * P.tol);       // centring target, floored: see mpcx_qp_stage.h

        // -------- corrector
        // second-order term damped by the affine step length (plain Mehrotra 2-cycles from boundary warm starts)
        const double rc0 = s0 * l0 + al * (dsa0 * dla0) - smu, rc1 = s1 * l1 + al * (dsa1 * dla1) - smu;
        const double rc2 = s2 * l2 + al * (dsa2 * dla2) - smu, rc3 = s3 * l3 + al * (dsa3 * dla3) - smu;
        w0 = m01 * (-rc0 + l0 * rp0) * is0; w1 = m01 * (-rc1 + l1 * rp1) * is1;
        w2 = m23 * (-rc2 + l2 * rp2) * is2; w3 = m23 * (-rc3 + l3 * rp3) * is3;
        rhs = m01 * (-rd - gt_apply(w0, w1, w2, w3));
        du = m01 * SOLVE(rhs);
        f2 = second_rows(du);
        const double ds0 = -rp0 - m01 * du, ds1 = -rp1 + m01 * du, ds2 = -rp2 - f2, ds3 = -rp3 + f2;
        const double dl0 = -m01 * (rc0 + l0 * ds0) * is0, dl1 = -m01 * (rc1 + l1 * ds1) * is1;
        const double dl2 = -m23 * (rc2 + l2 * ds2) * is2, dl3 = -m23 * (rc3 + l3 * ds3) * is3;
        // separate step lengths for the primal side (u, s) and the multipliers (see mpcx_qp_stage.h)
        rat = wave_max_dpp(fmax(fmax(-ds0 * is0, -ds1 * is1), fmax(-ds2 * is2, -ds3 * is3)));
        const double rat_d = wave_max_dpp(fmax(fmax(-dl0 * il0, -dl1 * il1), fmax(-dl2 * il2, -dl3 * il3)));
        double alpha = (MPCX_STEP_FRACTION < rat) ? MPCX_STEP_FRACTION * frcp(rat) : 1.0;           // min(1, fraction / rat)
        double alpha_d = (MPCX_STEP_FRACTION < rat_d) ? MPCX_STEP_FRACTION * frcp(rat_d) : 1.0;
        // centrality safeguard (wide neighbourhood): shorten the step until min_i s_i*lam_i >= 1e-3 * mu at the new point;
        // plain Mehrotra otherwise cycles on poorly centred iterates (mu oscillates, residuals -> 0)
        double mu_next = 0.0;
        for (int tr = 0; tr < 6; tr++) {
            const double q0 = (s0 + alpha * ds0) * (l0 + alpha_d * dl0), q1 = (s1 + alpha * ds1) * (l1 + alpha_d * dl1);
            const double q2 = (s2 + alpha * ds2) * (l2 + alpha_d * dl2), q3 = (s3 + alpha * ds3) * (l3 + alpha_d * dl3);
            const double big = 1e300;
            const double pmin = -wave_max_dpp(-fmin(fmin(val01 ? q0 : big, val01 ? q1 : big), fmin(val23 ? q2 : big, val23 ? q3 : big)));
            const double psum = wave_sum_dpp(m01 * (q0 + q1) + m23 * (q2 + q3));
            mu_next = psum * minv;
            if (pmin >= 1e-3 * (psum * minv)) break;
            alpha *= 0.7; alpha_d *= 0.7;
        }
        u += alpha * du;
        s0 += alpha * ds0; s1 += alpha * ds1; s2 += alpha * ds2; s3 += alpha * ds3;
        l0 += alpha_d * dl0; l1 += alpha_d * dl1; l2 += alpha_d * dl2; l3 += alpha_d * dl3;
        // is the new iterate close enough to polish?  (the rule of mpcx_qp_stage.h; resd_n / resp_n are relative to gnorm / hnorm)
        pol_pred = mu_next <= MPCX_POLISH_MU && (1.0 - alpha) * resp_n <= MPCX_POLISH_RP && (1.0 - fmin(alpha, alpha_d)) * resd_n <= MPCX_POLISH_RD;
    }
    // exit residuals (absolute, for the kkt[] report)
    const double res_d = wave_max_dpp(fabs(rd));
    const double res_p = wave_max_dpp(fmax(fmax(fabs(rp0), fabs(rp1)), fmax(fabs(rp2), fabs(rp3))));

    // ---------------------------------------------------------------- outputs: u and the linear prediction x
    // a problem this solver gives up on is handed to the stage solver untouched (its warm start may alias u_out)
    const bool keep_out = !(a.defer_fail && (status == MPCX_QP_MAXITER || status == MPCX_QP_NUMERIC));   // wave-uniform
    if (!keep_out && lane == 0) a.fail_list[atomicAdd(a.fail_count, 1)] = b;
    if (real && keep_out) a.u_out[(size_t)b * 2 * T + kind * T + k] = u;
    sh.ub[lane] = u;
    lds_sync();
    {
        const int s = lane;                       // stage s -> state at s+1
        const bool on = s < T;
        const int sc = on ? s : 0;
        const double onm = on ? 1.0 : 0.0;
        const double as = onm * sh.ub[sc], dsb = onm * sh.ub[NT + sc] * sh.beta[sc];
        const double va = scan_up32(as), fd = scan_up32(dsb);
        const double vs = dt * (va - as), ps = fd - dsb;       // deviation of (v, yaw) at stage s from (v0, yaw0)
        const double p0 = sh.pre[0][sc + 1] - sh.pre[0][sc], p1 = sh.pre[1][sc + 1] - sh.pre[1][sc];
        const double q0 = sh.pre[2][sc + 1] - sh.pre[2][sc], q1 = sh.pre[3][sc + 1] - sh.pre[3][sc];
        const double ix = onm * (p0 * vs + q0 * ps), iy = onm * (p1 * vs + q1 * ps);
        const double X = scan_up32(ix), Y = scan_up32(iy);
        double *xo = a.x_out + (size_t)b * 4 * W;
        if (on && keep_out) {
            const int t = s + 1;
            xo[0 * W + t] = x0 + v0 * sh.pre[0][t] + yaw0 * sh.pre[2][t] + sh.pre[4][t] + X;
            xo[1 * W + t] = y0 + v0 * sh.pre[1][t] + yaw0 * sh.pre[3][t] + sh.pre[5][t] + Y;
            xo[2 * W + t] = v0 + dt * va;
            xo[3 * W + t] = yaw0 + fd;
        }
        if (lane == 0 && keep_out) { xo[0] = x0; xo[W] = y0; xo[2 * W] = v0; xo[3 * W] = yaw0; }
    }
    if (lane == 0 && keep_out) {
        a.status[b] = status;
        a.iters[b] = it;
        a.kkt[4 * b + 0] = res_d; a.kkt[4 * b + 1] = res_p; a.kkt[4 * b + 2] = mu; a.kkt[4 * b + 3] = 0.0;
    }
  }   // work loop
}

template <int NT>
static void launch_qp(const QpArgs &a, hipStream_t st, int grid) {
    hipLaunchKernelGGL(qp_kernel<NT>, dim3(grid), dim3(64), 0, st, a);
}

}  // namespace mpcx

namespace mpcx {
// Work-queue order: longest expected job first.  key = previous iteration count (+ JUMP_BONUS when the problem's reference
// changed discontinuously since then, e.g. a different path cut); counting sort by descending key, 64 bins: scratch layout behind
// the order array is hist[64] | cursor[64].  With the trial pass the previous count of three quarters of the problems is 0, and
// EVERY problem that turns from unconstrained to hard (>= 10 iterations) between two steps has a moved reference
// (scripts/predict_probe.py: 100 % of them, 15 % of the moved ones); a launch lasts as long as its last problem, so those must
// start in round 0, ahead of the known 6-10-iteration ones: bonus 11 (6 until the trial pass: 0.864 -> 0.82 ms per launch;
// 3 / 8 / 9 / 10 / 12 / 14 / 20: 0.868 / 0.844 / 0.833 / 0.818 / 0.822 / 0.830 / 0.840).
constexpr int ORDER_BINS = MPCX_ORDER_BINS;
__device__ __forceinline__ int order_key(int i, const int32_t *hint, const int32_t *now, const int32_t *prev) {
    return order_key_of(hint ? hint[i] : 0, now && now[i] != prev[i]);
}
// Counting sort in two launches without zero-fills or global atomics: every block leaves its own key histogram (block_hist[b][64]);
// the scatter blocks add up the histograms in front of them.  Block 0 of the scatter also resets the work-queue ticket.
constexpr int ORDER_BLOCK = 256;
__global__ __launch_bounds__(ORDER_BLOCK) void qp_order_hist_kernel(int B, const int32_t *hint, const int32_t *now, const int32_t *prev, int32_t *block_hist) {
    __shared__ int32_t h[ORDER_BINS];
    if (threadIdx.x < ORDER_BINS) h[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) atomicAdd(&h[order_key(i, hint, now, prev)], 1);
    __syncthreads();
    if (threadIdx.x < ORDER_BINS) block_hist[blockIdx.x * ORDER_BINS + threadIdx.x] = h[threadIdx.x];
}
__global__ __launch_bounds__(ORDER_BLOCK) void qp_order_scatter_kernel(int B, const int32_t *hint, const int32_t *now, const int32_t *prev,
                                                                       const int32_t *block_hist, int32_t *order, int32_t *ticket) {
    __shared__ int32_t total[ORDER_BINS], base[ORDER_BINS], h[ORDER_BINS], part_all[ORDER_BLOCK], part_before[ORDER_BLOCK];
    if (blockIdx.x == 0 && threadIdx.x < MPCX_TICKET_WORDS) ticket[threadIdx.x] = 0;      // the queue head and the other per-launch counters
    {       // thread (part, bin) adds up every (ORDER_BLOCK / ORDER_BINS)-th block histogram: the loads of a thread are independent
        const int k = threadIdx.x % ORDER_BINS;
        int all = 0, before = 0;
        for (int b = threadIdx.x / ORDER_BINS; b < (int)gridDim.x; b += ORDER_BLOCK / ORDER_BINS) {
            const int c = block_hist[b * ORDER_BINS + k];
            all += c;
            before += b < (int)blockIdx.x ? c : 0;
        }
        part_all[threadIdx.x] = all; part_before[threadIdx.x] = before;
    }
    __syncthreads();
    if (threadIdx.x < ORDER_BINS) {
        int all = 0, before = 0;
        for (int q = 0; q < ORDER_BLOCK / ORDER_BINS; q++) { all += part_all[q * ORDER_BINS + threadIdx.x]; before += part_before[q * ORDER_BINS + threadIdx.x]; }
        total[threadIdx.x] = all; base[threadIdx.x] = before; h[threadIdx.x] = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {               // bins in descending key order: start of bin k = number of problems with a larger key
        int acc = 0;
        for (int k = ORDER_BINS - 1; k >= 0; k--) { const int t = total[k]; base[k] += acc; acc += t; }
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        const int k = order_key(i, hint, now, prev);
        order[base[k] + atomicAdd(&h[k], 1)] = i;       // rank inside the block (LDS atomic)
    }
}
}  // namespace mpcx

static inline size_t order_blocks(size_t B) { return (B + mpcx::ORDER_BLOCK - 1) / mpcx::ORDER_BLOCK; }
static inline int32_t *order_fail_list(mpcx_ctx *ctx, size_t B) { return ctx->order + B + order_blocks(B) * mpcx::ORDER_BINS; }

int32_t mpcx_ensure_order(mpcx_ctx *ctx, size_t B) {
    const size_t need = 2 * B + order_blocks(B) * mpcx::ORDER_BINS + 2;       // order | block histograms | list of given-up problems | its counter and ticket
    if (need <= ctx->order_cap) return MPCX_OK;
    if (ctx->order) (void)hipFree(ctx->order);
    ctx->order = nullptr; ctx->order_cap = 0;
    if (hipMalloc((void **)&ctx->order, need * sizeof(int32_t)) != hipSuccess)
        return mpcx_fail(ctx, MPCX_E_LAUNCH, "cannot allocate the work-queue order (%zu entries)", B);
    ctx->order_cap = need;
    return MPCX_OK;
}

int32_t mpcx_ensure_ticket(mpcx_ctx *ctx) {
    if (!ctx->ticket && hipMalloc((void **)&ctx->ticket, MPCX_TICKET_WORDS * sizeof(int32_t)) != hipSuccess)
        return mpcx_fail(ctx, MPCX_E_LAUNCH, "cannot allocate the work-queue word");
    return MPCX_OK;
}

// both solvers draw their problems from a queue, longest expected job first: order (ctx->order[0..B)) from ctx->order_hint / order_now / order_prev
int32_t mpcx_qp_build_order(mpcx_ctx *ctx, int32_t B, hipStream_t st) {
    int32_t rc = mpcx_ensure_order(ctx, (size_t)B);
    if (rc != MPCX_OK) return rc;
    rc = mpcx_ensure_ticket(ctx);
    if (rc != MPCX_OK) return rc;
    int32_t *block_hist = ctx->order + B;
    const int nb = (int)order_blocks((size_t)B);
    hipLaunchKernelGGL(mpcx::qp_order_hist_kernel, dim3(nb), dim3(mpcx::ORDER_BLOCK), 0, st, B, ctx->order_hint, ctx->order_now, ctx->order_prev, block_hist);
    hipLaunchKernelGGL(mpcx::qp_order_scatter_kernel, dim3(nb), dim3(mpcx::ORDER_BLOCK), 0, st, B, ctx->order_hint, ctx->order_now, ctx->order_prev,
                       block_hist, ctx->order, ctx->ticket);
    return MPCX_OK;
}

extern "C" int32_t mpcx_qp_solve_batch(mpcx_ctx *ctx, int32_t B, const double *x0, const double *xref,
                                       const double *xbar, const uint8_t *reaches_end, const double *u_warm,
                                       double *x_out, double *u_out, int32_t *status, int32_t *iters, double *kkt) {
    if (!ctx) return MPCX_E_INVALID;
    if (!ctx->have_mpc) return mpcx_fail(ctx, MPCX_E_INVALID, "mpcx_set_mpc_params has not been called");
    if (B == 0) return MPCX_OK;       // empty batch: nothing to do (zero-size tensors have null data pointers)
    if (B < 0 || !x0 || !xref || !xbar || !reaches_end || !x_out || !u_out || !status || !iters || !kkt)
        return mpcx_fail(ctx, MPCX_E_INVALID, "qp_solve_batch: null pointer or negative batch");
    if (B == 0) return MPCX_OK;
    { int32_t rc = mpcx_ensure_ticket(ctx); if (rc != MPCX_OK) return rc; }
    if (!ctx->order_ready) ctx->bins_clean = false;      // this solve draws tickets outside the closed loop's bookkeeping
    // persistent wavefronts: one per SIMD slot the kernel can occupy (1 wave/SIMD, 4 SIMDs/CU), never more than B
    const int grid = B < ctx->n_cu * 4 ? B : ctx->n_cu * 4;
    if (ctx->tune && ctx->tune_rows != B)
        return mpcx_fail(ctx, MPCX_E_INVALID, "qp_solve_batch: %d tuning rows are set but the batch has %d problems", ctx->tune_rows, B);
    const int T = ctx->mpc.T;
    static const int env_solver = [] { const char *e = getenv("MPCX_QP_KERNEL"); return !e ? 0 : !strcmp(e, "wave") ? 1 : !strcmp(e, "stage") ? 2 : 0; }();
    const int solver = ctx->qp_solver ? ctx->qp_solver : env_solver;
    // the five-state problem (mpcx_mpc_params.model == MPCX_MODEL_JERK5) exists in the stage-structured solver only
    if (ctx->mpc.model == MPCX_MODEL_JERK5 && solver == 1)
        return mpcx_fail(ctx, MPCX_E_INVALID, "qp_solve_batch: the condensed solver has no five-state (lib/mpc_jerk.py) variant; use solver 0 or 2");
    const bool use_stage = solver == 2 || ctx->mpc.model == MPCX_MODEL_JERK5 || (solver == 0 && (T > 20 || B >= MPCX_STAGE_MIN_BATCH));
    const int32_t *order = nullptr;
    if (ctx->order_ready) {                       // mpcx_closed_loop_run built it beside the window selection (and the scatter zeroed the ticket)
        ctx->order_ready = false;
        order = ctx->order;
    } else if (ctx->order_hint || ctx->order_now) {      // both solvers draw their problems from a queue: longest expected job first
        int32_t rc = mpcx_qp_build_order(ctx, B, ctx->stream);
        if (rc != MPCX_OK) return rc;
        order = ctx->order;
    } else if (hipMemsetAsync(ctx->ticket, 0, MPCX_TICKET_WORDS * sizeof(int32_t), ctx->stream) != hipSuccess)
        return mpcx_fail(ctx, MPCX_E_LAUNCH, "qp_solve_batch: hipMemsetAsync failed");
    mpcx::QpArgs a{ctx->mpc, B, ctx->ticket, u_warm != nullptr, x0, xref, xbar, u_warm, reaches_end, x_out, u_out, kkt, status, iters,
                   ctx->tune, ctx->tune != nullptr, order, order != nullptr, 0, nullptr, nullptr, nullptr, 0};
    if (!use_stage) {
        // The condensed solver's 64-row LDL' loses a few more digits than the Riccati recursion on the worst-conditioned problems
        // (lam/s ~ 1e10 in M = H + G'DG): about one problem in 2e6 of the benchmark workload breaks down there short of the
        // tolerance while the stage solver (and a dense Cholesky on the CPU) converge.  Those problems get a second chance: the condensed kernel
        // leaves them untouched and lists them, a small stage-solver launch (a few microseconds when the list is empty) solves them.
        int32_t rc = mpcx_ensure_order(ctx, (size_t)B);
        if (rc != MPCX_OK) return rc;
        // the list of given-up problems sits behind the order and its block histograms; its counter and the ticket of the launch that
        // works it off are words 4 and 5 of the context's counters (zeroed with the queue head)
        a.defer_fail = 1; a.fail_list = order_fail_list(ctx, (size_t)B); a.fail_count = ctx->ticket + 4;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (ctx->prof_qp && (hipStreamIsCapturing(ctx->stream, &cap) != hipSuccess || cap == hipStreamCaptureStatusNone)) {
        for (hipEvent_t *e : {&e0, &e1}) {
            if (!ctx->prof_free.empty()) { *e = ctx->prof_free.back(); ctx->prof_free.pop_back(); }
            else if (hipEventCreate(e) != hipSuccess) return mpcx_fail(ctx, MPCX_E_LAUNCH, "qp_solve_batch: hipEventCreate failed");
        }
        (void)hipEventRecord(e0, ctx->stream);
    }
    // stage-structured solver, eight lanes per problem (mpcx_qp_quad.hip) or the condensed solver of this file, one wavefront
    // per problem: mpcx_set_qp_solver / MPCX_QP_KERNEL=wave|stage choose; by default large batches and long horizons take the former
    if (use_stage) mpcx::launch_qp_stage(a, ctx->stream, ctx->n_cu);
    else if (T <= 10) mpcx::launch_qp<10>(a, ctx->stream, grid);
    else if (T <= 13) mpcx::launch_qp<13>(a, ctx->stream, grid);
    else if (T <= 20) mpcx::launch_qp<20>(a, ctx->stream, grid);
    else mpcx::launch_qp<32>(a, ctx->stream, grid);
    if (!use_stage) {
        mpcx::QpArgs f = a;
        f.defer_fail = 0;
        f.order = a.fail_list; f.has_order = 1;
        f.queue_len = a.fail_count; f.has_queue_len = 1;
        f.ticket = ctx->ticket + 5;                         // zeroed together with the counter
        mpcx::launch_qp_stage(f, ctx->stream, 8);           // 32 wavefronts = 256 problems at a time
    }
    if (e0) {
        (void)hipEventRecord(e1, ctx->stream);
        ctx->prof_ev.push_back(e0); ctx->prof_ev.push_back(e1);
    }
    return mpcx_check_launch(ctx, "qp_kernel");
}
