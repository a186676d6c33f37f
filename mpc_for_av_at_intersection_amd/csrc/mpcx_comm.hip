// mpcx_comm.hip -- the one exchange step of the multi-GPU layouts: an RCCL all-gather over xGMI of per-agent states
// (x, y, v, yaw, accel, steer -- what MovingObstacle*.get() returns, scenarios/mpc_intersection.py:119-122), from which every
// rank predicts the other agents itself (moving_obstacles_prediction.py:21-47 is a deterministic rollout of those six numbers,
// so 48 B per agent travel instead of the 840-B predicted trajectory).  The reference is single-process; this file has no
// counterpart there.
//
// One communicator per context, one rank per GPU.  xGMI is point to point (7 links per GPU): at 8 ranks the payload per rank is
// B * A_loc * 48 B = 196 KB for the headline batch, far below the size where the choice of algorithm matters; the call is
// enqueued on the context's stream between the pool pack and the prediction kernel.
#include "mpcx_common.h"
#include <rccl/rccl.h>
#include <cstring>

namespace mpcx {

struct PermArgs {
    int n_inst, a_loc, world;
    const double *xchg;   // [world][n_inst][a_loc][6]
    double *all;          // [n_inst][world * a_loc][6]
};
// agent-sharded layout: rank r owns agents r*a_loc .. r*a_loc + a_loc - 1 of EVERY instance
__global__ __launch_bounds__(256) void interleave_kernel(PermArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;        // one double each
    const int total = a.world * a.n_inst * a.a_loc * 6;
    if (i >= total) return;
    const int k = i % 6, row = i / 6;
    const int al = row % a.a_loc, b = (row / a.a_loc) % a.n_inst, r = row / (a.a_loc * a.n_inst);
    a.all[((size_t)b * (a.world * a.a_loc) + r * a.a_loc + al) * 6 + k] = a.xchg[i];
}

}  // namespace mpcx

extern "C" int32_t mpcx_comm_unique_id(void *id128) {
    static_assert(sizeof(ncclUniqueId) == MPCX_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id128) return MPCX_E_INVALID;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return MPCX_E_LAUNCH;
    memcpy(id128, &id, sizeof id);
    return MPCX_OK;
}

extern "C" int32_t mpcx_comm_init(mpcx_ctx *ctx, int32_t world, int32_t rank, const void *id128) {
    if (!ctx) return MPCX_E_INVALID;
    if (!id128 || world < 1 || rank < 0 || rank >= world) return mpcx_fail(ctx, MPCX_E_INVALID, "comm_init: bad world/rank or null id");
    if (ctx->comm) return mpcx_fail(ctx, MPCX_E_INVALID, "comm_init: the context already has a communicator");
    if (hipSetDevice(ctx->device) != hipSuccess) return mpcx_fail(ctx, MPCX_E_LAUNCH, "comm_init: hipSetDevice failed");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) return mpcx_fail(ctx, MPCX_E_LAUNCH, "comm_init: ncclCommInitRank: %s", ncclGetErrorString(r));
    ctx->comm = comm; ctx->comm_world = world; ctx->comm_rank = rank;
    return MPCX_OK;
}

extern "C" int32_t mpcx_comm_destroy(mpcx_ctx *ctx) {
    if (!ctx) return MPCX_E_INVALID;
    if (ctx->comm) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)ncclCommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->comm_world = 1; ctx->comm_rank = 0;
    return MPCX_OK;
}

extern "C" int32_t mpcx_allgather_states(mpcx_ctx *ctx, int32_t layout, int32_t n_inst, int32_t agents_local,
                                         const double *local, double *all) {
    if (!ctx) return MPCX_E_INVALID;
    if (n_inst < 0 || agents_local < 0 || (layout != MPCX_SHARD_INSTANCES && layout != MPCX_SHARD_AGENTS))
        return mpcx_fail(ctx, MPCX_E_INVALID, "allgather_states: negative size or unknown layout");
    const size_t rows = (size_t)n_inst * agents_local;
    if (rows == 0) return MPCX_OK;
    if (!local || !all) return mpcx_fail(ctx, MPCX_E_INVALID, "allgather_states: null pointer");
    const int world = ctx->comm ? ctx->comm_world : 1;
    if (!ctx->comm) {       // a single rank: the gathered table is the local one
        if (local != all && hipMemcpyAsync(all, local, rows * 6 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
            return mpcx_fail(ctx, MPCX_E_LAUNCH, "allgather_states: hipMemcpyAsync failed");
        return MPCX_OK;
    }
    double *dst = all;
    if (layout == MPCX_SHARD_AGENTS && world > 1 && agents_local != 0) {
        const size_t need = rows * 6 * world;
        if (need > ctx->xchg_cap) {
            if (ctx->xchg) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->xchg); }
            ctx->xchg = nullptr; ctx->xchg_cap = 0;
            if (hipMalloc((void **)&ctx->xchg, need * sizeof(double)) != hipSuccess)
                return mpcx_fail(ctx, MPCX_E_LAUNCH, "allgather_states: cannot allocate %zu bytes", need * sizeof(double));
            ctx->xchg_cap = need;
        }
        dst = ctx->xchg;
    }
    const ncclResult_t r = ncclAllGather(local, dst, rows * 6, ncclDouble, (ncclComm_t)ctx->comm, ctx->stream);
    if (r != ncclSuccess) return mpcx_fail(ctx, MPCX_E_LAUNCH, "allgather_states: ncclAllGather: %s", ncclGetErrorString(r));
    if (dst != all) {
        mpcx::PermArgs pa{n_inst, agents_local, world, ctx->xchg, all};
        const size_t total = rows * 6 * world;
        hipLaunchKernelGGL(mpcx::interleave_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, pa);
        return mpcx_check_launch(ctx, "interleave_kernel");
    }
    return MPCX_OK;
}
