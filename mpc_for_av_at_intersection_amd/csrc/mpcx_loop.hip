// mpcx_loop.hip -- the reference's scenario loop body (main/scenarios/mpc_intersection.py:95-159) for P agents as a
// device-resident pipeline: n_steps x [pool pack -> predict -> conflict search + path cut -> reference window ->
// rollout -> QP -> plant], enqueued back to back on the context's stream (optionally as a replayed hipGraph), no
// host synchronisation or host arithmetic in between.  Every stage is the kernel behind the per-stage C entry
// point, called with the very buffers the descriptor names, so a run is bit-identical to driving the stages one
// by one from the host.
#include "mpcx_common.h"
#include <cstring>
#include <cstdlib>

namespace mpcx {

struct PackArgs {
    int P;
    const double *state, *applied;
    double *obs6;
};

// MovingObstacle*.get() of the reference (mpc_intersection.py:119-122): (x, y, v, yaw, a, steer)
__global__ __launch_bounds__(256) void pack_pool_kernel(PackArgs a) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.P) return;
    double *o = a.obs6 + 6 * (size_t)q;
    const double *s = a.state + 4 * (size_t)q;
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3];
    o[4] = a.applied[2 * q + 1];
    o[5] = a.applied[2 * q];
}

}  // namespace mpcx

static int32_t enqueue_step(mpcx_ctx *ctx, const mpcx_interaction_params *ip, const mpcx_closed_loop *c) {
    const int P = c->P;
    ctx->bins_clean = false;        // until the plant kernel of this step is enqueued
    int32_t rc, pool_rows = P;
    // the warm-start rollout of this step needs only the states and the previous solution: it runs on the side stream BESIDE the pool pack,
    // the prediction and the conflict search (a chain of T dependent sincos / tan per agent, 35-45 us) and is joined by the window selection
    static const bool late_fork = getenv("MPCX_DEV_LATE_ROLLOUT") != nullptr;      // dev aid: the rollout beside the window selection instead of beside the conflict search
    if (!late_fork) {
        rc = mpcx_rollout_fork(ctx, P, c->state, c->u_sol, c->xbar);
        if (rc != MPCX_OK) return rc;
        ctx->rollout_forked = true;
    }
    if (c->exchange == MPCX_SHARD_AGENTS) {
        // agent-sharded layout: this rank's rows travel to every rank, every rank assembles the whole pool (one RCCL all-gather)
        mpcx::PackArgs pa{P, c->state, c->applied, c->obs_local};
        hipLaunchKernelGGL(mpcx::pack_pool_kernel, dim3((P + 63) / 64), dim3(64), 0, ctx->stream, pa);
        rc = mpcx_allgather_states(ctx, MPCX_SHARD_AGENTS, c->n_inst, c->agents_local, c->obs_local, c->obs6);
        if (rc != MPCX_OK) return rc;
        pool_rows = P * ctx->comm_world;
    } else {
        // local pool: row q is agent q, and the prediction kernel (inside mpcx_interaction_batch) packs it on its way -- no launch of its own
        ctx->pack_state = c->state; ctx->pack_applied = c->applied;
    }
    // the conflict search leaves the cut lengths of the previous step in ctx->prev_cut (the queue of the QP kernel puts the agents whose
    // cut moved at the front) and files every agent under its work-queue key (previous iteration count + "the cut moved"): hard problems first.  The window
    // selection then writes the queue order and the plant kernel resets bins and ticket, so the counting sort costs no launch of its own.
    const bool binned = P < (1 << 24);
    ctx->inter_prev_save = ctx->prev_cut;
    ctx->inter_near = ctx->prev_cut + P;             // (its start index, largest of its three nearest indices) per agent, for the window selection
    ctx->bin_hint = binned ? c->iters : nullptr;
    rc = mpcx_interaction_batch(ctx, ip, P, c->state, c->path_xyyaw, c->path_cs, c->path_off, c->path_len,
                                c->cut_len /* previous step's cut; read before it is rewritten */, pool_rows, c->obs6,
                                c->obs_off, c->obs_cnt, c->obs_skip, c->traj_idx, c->hit_idx, c->hit_xy, c->cut_len);
    ctx->inter_prev_save = nullptr;
    ctx->inter_near = nullptr;
    ctx->bin_hint = nullptr;
    ctx->pack_state = nullptr; ctx->pack_applied = nullptr;
    if (rc != MPCX_OK) return rc;
    // lib/mpc.py:226-237: MAX_ITER passes of (reference window, rollout, QP); from the second pass on the window is spaced by the
    // previous pass's speeds (row 2 of its x) and the rollout uses its inputs.  (Where a pass fails the reference crashes in the next
    // one -- zip over None; here the next pass starts from the untouched warm start, as after a failed step.)
    const int Wd = ctx->mpc.T + 1;
    for (int pass = 0; pass < ctx->lin_passes; pass++) {
        ctx->bin_scatter = binned && pass == 0;
        ctx->window_near = ctx->prev_cut + P; ctx->window_tidx = c->traj_idx;
        rc = mpcx_mpc_prepare_batch_ov(ctx, P, c->state, c->u_sol, c->path_xyyaw, c->path_v, c->path_off, c->cut_len, c->dl,
                                       c->target_ind, pass ? c->x_sol + 2 * Wd : nullptr, 4 * (int64_t)Wd, c->xref, c->reaches_end, c->xbar);
        ctx->window_near = ctx->window_tidx = nullptr;
        if (rc != MPCX_OK) return rc;
        // (further linearisation passes build their order in line, from the iteration counts of the pass before)
        const int32_t *hint_before = ctx->order_hint, *now_before = ctx->order_now, *prev_before = ctx->order_prev;
        ctx->order_hint = c->iters; ctx->order_now = c->cut_len; ctx->order_prev = ctx->prev_cut;
        ctx->order_ready = binned && pass == 0;
        rc = mpcx_qp_solve_batch(ctx, P, c->state, c->xref, c->xbar, c->reaches_end, c->u_sol, c->x_sol, c->u_sol,
                                 c->status, c->iters, c->kkt);
        ctx->order_hint = hint_before; ctx->order_now = now_before; ctx->order_prev = prev_before;
        ctx->order_ready = false;
        if (rc != MPCX_OK) return rc;
    }
    ctx->stats_iters = c->iters;
    ctx->bin_reset = binned;
    rc = mpcx_plant_step_batch(ctx, P, c->state, c->u_sol, c->status, c->applied);
    ctx->stats_iters = nullptr;
    if (rc == MPCX_OK) ctx->bins_clean = binned;
    return rc;
}

extern "C" int32_t mpcx_closed_loop_run(mpcx_ctx *ctx, const mpcx_interaction_params *ip, const mpcx_closed_loop *c,
                                        int32_t n_steps, int32_t use_graph) {
    if (!ctx) return MPCX_E_INVALID;
    if (!ctx->have_mpc) return mpcx_fail(ctx, MPCX_E_INVALID, "mpcx_set_mpc_params has not been called");
    if (!ip || !c || n_steps < 0 || c->P < 0) return mpcx_fail(ctx, MPCX_E_INVALID, "closed_loop_run: null descriptor or negative count");
    if (n_steps == 0 || c->P == 0) return MPCX_OK;
    if (!c->state || !c->applied || !c->obs6 || !c->path_xyyaw || !c->path_cs || !c->path_off || !c->path_len ||
        !c->obs_off || !c->obs_cnt || !c->traj_idx || !c->target_ind || !c->hit_idx || !c->cut_len || !c->hit_xy ||
        !c->xref || !c->xbar || !c->reaches_end || !c->x_sol || !c->u_sol || !c->status || !c->iters || !c->kkt || !(c->dl > 0))
        return mpcx_fail(ctx, MPCX_E_INVALID, "closed_loop_run: null buffer in the descriptor or dl <= 0");
    if (ip->pred_steps < 1 || ip->pred_steps > MPCX_PRED_STEPS_MAX)
        return mpcx_fail(ctx, MPCX_E_INVALID, "closed_loop_run: pred_steps outside 1..%d", MPCX_PRED_STEPS_MAX);
    if (c->exchange != 0 && c->exchange != MPCX_SHARD_AGENTS)
        return mpcx_fail(ctx, MPCX_E_INVALID, "closed_loop_run: exchange must be 0 or MPCX_SHARD_AGENTS");
    if (c->exchange == MPCX_SHARD_AGENTS) {
        if (!c->obs_local || c->n_inst < 0 || c->agents_local < 0 || (long)c->n_inst * c->agents_local != (long)c->P)
            return mpcx_fail(ctx, MPCX_E_INVALID, "closed_loop_run: agent-sharded layout needs obs_local and P = n_inst * agents_local");
        if (use_graph) return mpcx_fail(ctx, MPCX_E_INVALID, "closed_loop_run: the agent-sharded layout (RCCL exchange) is not captured into a graph");
    }
    // everything that allocates happens before the first launch (and outside any capture)
    const size_t pool_rows = (size_t)c->P * (c->exchange == MPCX_SHARD_AGENTS ? (size_t)ctx->comm_world : 1);
    int32_t rc = mpcx_ensure_pred(ctx, pool_rows * ip->pred_steps * 4);
    if (rc != MPCX_OK) return rc;
    rc = mpcx_ensure_ticket(ctx);
    if (rc != MPCX_OK) return rc;
    {
        const size_t slots = ((size_t)c->P + 63) / 64;
        if (slots > ctx->stats_slots) {         // a larger batch: the counters so far are folded into slot 0 of the new table
            int64_t keep[4] = {0, 0, 0, 0};
            if (ctx->stats) { rc = mpcx_closed_loop_stats(ctx, keep, 0); if (rc != MPCX_OK) return rc; (void)hipFree(ctx->stats); ctx->stats = nullptr; ctx->stats_slots = 0; }
            if (hipMalloc((void **)&ctx->stats, 4 * slots * sizeof(unsigned long long)) != hipSuccess ||
                hipMemsetAsync(ctx->stats, 0, 4 * slots * sizeof(unsigned long long), ctx->stream) != hipSuccess ||
                hipMemcpyAsync(ctx->stats, keep, sizeof keep, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                return mpcx_fail(ctx, MPCX_E_LAUNCH, "closed_loop_run: cannot allocate the run statistics");
            ctx->stats_slots = slots;
        }
    }
    rc = mpcx_ensure_order(ctx, (size_t)c->P);
    if (rc != MPCX_OK) return rc;
    {       // queue bins: counters + (key, slot) per agent.  The plant kernel leaves counters and ticket zeroed step by step; they are
            // filled here only when the host cannot know that (first run, a step that failed half way, a solve outside the loop since)
        const size_t need = (size_t)MPCX_ORDER_COPIES * MPCX_ORDER_BINS + (size_t)c->P;
        if (need > ctx->bins_cap) {
            if (ctx->bins) (void)hipFree(ctx->bins);
            ctx->bins = nullptr; ctx->bins_cap = 0; ctx->bins_clean = false;
            if (hipMalloc((void **)&ctx->bins, need * sizeof(int32_t)) != hipSuccess)
                return mpcx_fail(ctx, MPCX_E_LAUNCH, "closed_loop_run: cannot allocate the queue bins of %d agents", c->P);
            ctx->bins_cap = need;
        }
        if (!ctx->bins_clean) {
            if (hipMemsetAsync(ctx->bins, 0, (size_t)MPCX_ORDER_COPIES * MPCX_ORDER_BINS * sizeof(int32_t), ctx->stream) != hipSuccess ||
                hipMemsetAsync(ctx->ticket, 0, MPCX_TICKET_WORDS * sizeof(int32_t), ctx->stream) != hipSuccess)
                return mpcx_fail(ctx, MPCX_E_LAUNCH, "closed_loop_run: hipMemsetAsync failed");
            ctx->bins_clean = true;
        }
    }
    if ((size_t)c->P > ctx->prev_cut_cap) {
        if (ctx->prev_cut) (void)hipFree(ctx->prev_cut);
        ctx->prev_cut = nullptr; ctx->prev_cut_cap = 0;
        if (hipMalloc((void **)&ctx->prev_cut, 4 * (size_t)c->P * sizeof(int32_t)) != hipSuccess)      // P cut lengths + 3 P nearest-index hints
            return mpcx_fail(ctx, MPCX_E_LAUNCH, "closed_loop_run: cannot allocate %d cut lengths", c->P);
        ctx->prev_cut_cap = (size_t)c->P;
    }

    if (!use_graph) {
        for (int s = 0; s < n_steps; s++) {
            rc = enqueue_step(ctx, ip, c);
        ctx->rollout_forked = false;
            if (rc != MPCX_OK) return rc;
        }
        return MPCX_OK;
    }

    if (!ctx->stream) return mpcx_fail(ctx, MPCX_E_INVALID, "closed_loop_run: graph replay needs a non-default stream");
    if (ctx->prof_qp)       // the event pairs of mpcx_profile_qp cannot be recorded inside a replayed graph: say so instead of reporting 0 launches
        return mpcx_fail(ctx, MPCX_E_INVALID, "closed_loop_run: mpcx_profile_qp is on; the QP launches of a replayed graph are not bracketed by events -- run without graph or switch the hook off");
    unsigned char key[sizeof ctx->loop_key];
    static_assert(sizeof(mpcx_closed_loop) + sizeof(mpcx_interaction_params) + sizeof(mpcx_mpc_params) + 9 * sizeof(void *) <= sizeof key,
                  "loop_key too small");
    memset(key, 0, sizeof key);
    size_t o = 0;
    memcpy(key + o, c, sizeof *c); o += sizeof *c;
    memcpy(key + o, ip, sizeof *ip); o += sizeof *ip;
    memcpy(key + o, &ctx->mpc, sizeof ctx->mpc); o += sizeof ctx->mpc;
    memcpy(key + o, &ctx->pred, sizeof ctx->pred); o += sizeof ctx->pred;
    memcpy(key + o, &ctx->tune, sizeof ctx->tune); o += sizeof ctx->tune;
    memcpy(key + o, &ctx->order, sizeof ctx->order); o += sizeof ctx->order;
    memcpy(key + o, &ctx->prev_cut, sizeof ctx->prev_cut); o += sizeof ctx->prev_cut;
    memcpy(key + o, &ctx->bins, sizeof ctx->bins); o += sizeof ctx->bins;
    memcpy(key + o, &ctx->qp_solver, sizeof ctx->qp_solver); o += sizeof ctx->qp_solver;      // the captured launch is the solver chosen at capture time
    memcpy(key + o, &ctx->lin_passes, sizeof ctx->lin_passes); o += sizeof ctx->lin_passes;
    memcpy(key + o, &ctx->stats, sizeof ctx->stats);
    if (!ctx->loop_exec || memcmp(key, ctx->loop_key, sizeof key) != 0) {
        if (ctx->loop_exec) {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipGraphExecDestroy(ctx->loop_exec);
            ctx->loop_exec = nullptr;
        }
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) != hipSuccess)
            return mpcx_fail(ctx, MPCX_E_LAUNCH, "closed_loop_run: hipStreamBeginCapture failed");
        rc = enqueue_step(ctx, ip, c);
        ctx->rollout_forked = false;
        hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
        if (rc != MPCX_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess || !graph) return mpcx_fail(ctx, MPCX_E_LAUNCH, "closed_loop_run: stream capture failed: %s", hipGetErrorString(e));
        e = hipGraphInstantiate(&ctx->loop_exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { ctx->loop_exec = nullptr; return mpcx_fail(ctx, MPCX_E_LAUNCH, "closed_loop_run: hipGraphInstantiate: %s", hipGetErrorString(e)); }
        memcpy(ctx->loop_key, key, sizeof key);
    }
    for (int s = 0; s < n_steps; s++)
        if (hipGraphLaunch(ctx->loop_exec, ctx->stream) != hipSuccess)
            return mpcx_fail(ctx, MPCX_E_LAUNCH, "closed_loop_run: hipGraphLaunch failed");
    return MPCX_OK;
}
