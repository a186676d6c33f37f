// mpcx_misc.hip -- small batched geometry entry points that complete the reference's call surface
// (paths relative to /root/reference/main):
//   lib/linalg.py:4-54 transform_2d_pts  (motion_primitive_at / collision_checking_points_at /
//                                         path_to_full_trajectory, motion_primitive_search.py:77-85,123-135)
//   lib/collision_avoidance.py:107-119   get_cutoff_curve_by_position_idx
//   lib/moving_obstacles_prediction.py:21-47 state_prediction (poses, not disc centres)
#include "mpcx_common.h"

namespace mpcx {

struct XformArgs {
    int n_items, max_pts;
    const double *nodes;        // [n_items][3]
    const int32_t *pts_off;     // [n_items] offset (in points) of the item's source points
    const int32_t *pts_cnt;     // [n_items]
    const double *pts;          // [npts][3] (x, y, theta)
    double *out;                // [n_items][max_pts][3]
};

__global__ __launch_bounds__(256) void transform_kernel(XformArgs a) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)a.n_items * a.max_pts) return;
    const int it = (int)(gid / a.max_pts), j = (int)(gid % a.max_pts);
    double *o = a.out + (size_t)gid * 3;
    if (j >= a.pts_cnt[it]) { o[0] = o[1] = o[2] = 0.0; return; }
    const double x = a.nodes[3 * it], y = a.nodes[3 * it + 1], th = a.nodes[3 * it + 2];
    double s, c;
    sincos(th, &s, &c);
    const bool rot_only = (x == 0.0 && y == 0.0);      // linalg.py:13-17
    const double tx = rot_only ? 0.0 : x, ty = rot_only ? 0.0 : y;
    const double *p = a.pts + 3 * ((size_t)a.pts_off[it] + j);
    if (a.pts_cnt[it] >= 2) {        // (x*m0 + y*m1) + t : first product rounded, second fused (N >= 2 rows)
        o[0] = __dadd_rn(fma(p[1], -s, __dmul_rn(p[0], c)), tx);
        o[1] = __dadd_rn(fma(p[1], c, __dmul_rn(p[0], s)), ty);
    } else {                         // single row: fma(x, m0, y*m1) + t
        o[0] = __dadd_rn(fma(p[0], c, __dmul_rn(p[1], -s)), tx);
        o[1] = __dadd_rn(fma(p[0], s, __dmul_rn(p[1], c)), ty);
    }
    o[2] = __dadd_rn(p[2], th);      // linalg.py:48: theta column is NOT normalised here
}

struct CutArgs {
    int P;
    const double *pts;
    const int32_t *off, *len;
    const double *xy;
    double radius;
    int32_t *out;
};

__global__ __launch_bounds__(64) void cutoff_kernel(CutArgs a) {
    const int p = blockIdx.x, lane = threadIdx.x;
    const double *pts = a.pts + 3 * (size_t)a.off[p];
    const int n = a.len[p];
    const double x = a.xy[2 * p], y = a.xy[2 * p + 1];
    int best = 0x7fffffff;
    for (int j = lane; j < n; j += WAVE) {
        const double dx = __dadd_rn(pts[3 * j], -x), dy = __dadd_rn(pts[3 * j + 1], -y);
        if (__dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy))) <= a.radius) best = j < best ? j : best;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { int o = __shfl_xor(best, s, WAVE); best = o < best ? o : best; }
    if (lane == 0) a.out[p] = (best == 0x7fffffff) ? -1 : best;
}

struct PoseArgs {
    int n, steps;
    double dt, L;
    const double *obs6;
    double *out;   // [n][steps][3]
};

__global__ __launch_bounds__(256) void predict_pose_kernel(PoseArgs a) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= a.n) return;
    const double *s6 = a.obs6 + 6 * (size_t)o;
    double x = s6[0], y = s6[1], v = s6[2], yaw = s6[3];
    const double acc = s6[4], tn = tan(s6[5]);
    double *out = a.out + (size_t)o * a.steps * 3;
    for (int k = 0; k < a.steps; k++) {
        double s, c;
        sincos(yaw, &s, &c);
        x = __dadd_rn(x, __dmul_rn(__dmul_rn(v, c), a.dt));
        y = __dadd_rn(y, __dmul_rn(__dmul_rn(v, s), a.dt));
        v = __dadd_rn(v, __dmul_rn(acc, a.dt));
        yaw = __dadd_rn(yaw, __dmul_rn(__dmul_rn(__ddiv_rn(v, a.L), tn), a.dt));
        out[3 * k] = x; out[3 * k + 1] = y; out[3 * k + 2] = yaw;
    }
}

// self-test of the DPP wave helpers (mpcx_common.h): out[0..63] scan_up32, [64..127] scan_down32, [128..191] lane_next,
// [192..255] lane_prev, [256] wave_sum_dpp, [257] wave_max_dpp, [258..321] frcp1(in) 
__global__ __launch_bounds__(64) void selftest_kernel(const double *in, double *out) {
    const int lane = threadIdx.x;
    const double v = in[lane];
    out[lane] = scan_up32(v);
    out[64 + lane] = scan_down32(v, lane);
    out[128 + lane] = lane_next(v, -1.0);
    out[192 + lane] = lane_prev(v, -1.0);
    const double s = wave_sum_dpp(v), m = wave_max_dpp(v);
    if (lane == 0) { out[256] = s; out[257] = m; }
    out[258 + lane] = frcp1(v);
}

// self-test of the f64 MFMA lane maps the Hessian build relies on: D = A(16x4) * B(4x16), A and B given row-major
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void selftest_mfma_kernel(const double *A, const double *B, double *D) {
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];        // A[i = l&15][k = l>>4]
    const double b = B[(l >> 4) * 16 + (l & 15)];       // B[k = l>>4][j = l&15]
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];   // D[row = (l>>4) + 4r][col = l&15]
}

}  // namespace mpcx

extern "C" int32_t mpcx_selftest_mfma(mpcx_ctx *ctx, const double *A64, const double *B64, double *D256) {
    if (!ctx || !A64 || !B64 || !D256) return MPCX_E_INVALID;
    hipLaunchKernelGGL(mpcx::selftest_mfma_kernel, dim3(1), dim3(64), 0, ctx->stream, A64, B64, D256);
    return mpcx_check_launch(ctx, "selftest_mfma_kernel");
}

extern "C" int32_t mpcx_selftest_wave_ops(mpcx_ctx *ctx, const double *in64, double *out322) {
    if (!ctx || !in64 || !out322) return MPCX_E_INVALID;
    hipLaunchKernelGGL(mpcx::selftest_kernel, dim3(1), dim3(64), 0, ctx->stream, in64, out322);
    return mpcx_check_launch(ctx, "selftest_kernel");
}

extern "C" int32_t mpcx_transform_batch(mpcx_ctx *ctx, int32_t n_items, int32_t max_pts, const double *nodes,
                                        const int32_t *pts_off, const int32_t *pts_cnt, const double *pts, double *out) {
    if (!ctx) return MPCX_E_INVALID;
    if (n_items == 0) return MPCX_OK;
    if (n_items < 0 || max_pts < 1 || !nodes || !pts_off || !pts_cnt || !pts || !out)
        return mpcx_fail(ctx, MPCX_E_INVALID, "transform_batch: null pointer or bad size");
    if (n_items == 0) return MPCX_OK;
    mpcx::XformArgs a{n_items, max_pts, nodes, pts_off, pts_cnt, pts, out};
    const long long total = (long long)n_items * max_pts;
    hipLaunchKernelGGL(mpcx::transform_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, a);
    return mpcx_check_launch(ctx, "transform_kernel");
}

extern "C" int32_t mpcx_cutoff_index_batch(mpcx_ctx *ctx, int32_t P, const double *pts, const int32_t *off,
                                           const int32_t *len, const double *xy, double radius, int32_t *out) {
    if (!ctx) return MPCX_E_INVALID;
    if (P == 0) return MPCX_OK;
    if (P < 0 || !pts || !off || !len || !xy || !out) return mpcx_fail(ctx, MPCX_E_INVALID, "cutoff_index_batch: null pointer");
    if (P == 0) return MPCX_OK;
    mpcx::CutArgs a{P, pts, off, len, xy, radius, out};
    hipLaunchKernelGGL(mpcx::cutoff_kernel, dim3(P), dim3(64), 0, ctx->stream, a);
    return mpcx_check_launch(ctx, "cutoff_kernel");
}

extern "C" int32_t mpcx_predict_obstacles_batch(mpcx_ctx *ctx, int32_t n, int32_t steps, double dt, double L,
                                                const double *obs6, double *out_xyyaw) {
    if (!ctx) return MPCX_E_INVALID;
    if (n == 0) return MPCX_OK;
    if (n < 0 || steps < 1 || !(dt > 0) || !(L > 0) || !obs6 || !out_xyyaw)
        return mpcx_fail(ctx, MPCX_E_INVALID, "predict_obstacles_batch: null pointer or bad size");
    if (n == 0) return MPCX_OK;
    mpcx::PoseArgs a{n, steps, dt, L, obs6, out_xyyaw};
    hipLaunchKernelGGL(mpcx::predict_pose_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, a);
    return mpcx_check_launch(ctx, "predict_pose_kernel");
}
