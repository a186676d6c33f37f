// mpcx_api.hip -- context, parameters and error plumbing of libmpcx.so (see include/mpcx.h).
#include "mpcx_common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

int32_t mpcx_fail(mpcx_ctx *ctx, int32_t code, const char *fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

int32_t mpcx_check_launch(mpcx_ctx *ctx, const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mpcx_fail(ctx, MPCX_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return MPCX_OK;
}

extern "C" {

const char *mpcx_version(void) { return "mpcx 0.1 (gfx950)"; }

mpcx_ctx *mpcx_create(int32_t device, void *hip_stream) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    mpcx_ctx *c = new (std::nothrow) mpcx_ctx();
    if (!c) return nullptr;
    c->device = device;
    c->stream = (hipStream_t)hip_stream;
    c->have_mpc = false;
    c->pred = nullptr;
    c->ticket = nullptr;
    {
        hipDeviceProp_t prop;
        c->n_cu = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    c->pred_cap = 0;
    c->loop_exec = nullptr;
    c->prof_qp = false;
    c->qp_solver = 0;
    c->multi = nullptr;
    c->multi_cap = 0;
    c->order_hint = nullptr;
    c->order_now = c->order_prev = nullptr;
    c->prev_cut = nullptr;
    c->prev_cut_cap = 0;
    c->order = nullptr;
    c->order_cap = 0;
    c->tune = nullptr;
    c->tune_rows = 0;
    memset(c->loop_key, 0, sizeof c->loop_key);
    c->err[0] = 0;
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) { mpcx_destroy(c); return nullptr; }
    return c;
}

void mpcx_destroy(mpcx_ctx *ctx) {
    if (!ctx) return;
    if (ctx->loop_exec) { (void)hipStreamSynchronize(ctx->stream); (void)hipGraphExecDestroy(ctx->loop_exec); }
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->prof_free) (void)hipEventDestroy(e);
    if (ctx->order) (void)hipFree(ctx->order);
    if (ctx->multi) (void)hipFree(ctx->multi);
    if (ctx->cs) (void)hipFree(ctx->cs);
    if (ctx->side) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamDestroy(ctx->side); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->prev_cut) (void)hipFree(ctx->prev_cut);
    if (ctx->bins) (void)hipFree(ctx->bins);
    if (ctx->pred) (void)hipFree(ctx->pred);
    if (ctx->ticket) (void)hipFree(ctx->ticket);
    (void)mpcx_comm_destroy(ctx);
    if (ctx->stats) (void)hipFree(ctx->stats);
    if (ctx->xchg) (void)hipFree(ctx->xchg);
    delete ctx;
}

const char *mpcx_last_error(mpcx_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int32_t mpcx_set_mpc_params(mpcx_ctx *ctx, const mpcx_mpc_params *p) {
    if (!ctx || !p) return MPCX_E_INVALID;
    if (p->T < 1 || p->T > MPCX_T_MAX) return mpcx_fail(ctx, MPCX_E_INVALID, "horizon T=%d outside 1..%d", p->T, MPCX_T_MAX);
    if (!(p->dt > 0) || !(p->L > 0) || p->max_iter < 1 || !(p->tol > 0))
        return mpcx_fail(ctx, MPCX_E_INVALID, "dt, L, tol must be positive and max_iter >= 1");
    if (p->model != MPCX_MODEL_BICYCLE4 && p->model != MPCX_MODEL_JERK5)
        return mpcx_fail(ctx, MPCX_E_INVALID, "model %d: MPCX_MODEL_BICYCLE4 (lib/mpc.py) or MPCX_MODEL_JERK5 (lib/mpc_jerk.py)", p->model);
    if (p->model == MPCX_MODEL_JERK5 && !(p->jerk_weight >= 0.0))
        return mpcx_fail(ctx, MPCX_E_INVALID, "jerk_weight must be >= 0");
    ctx->mpc = *p;
    ctx->have_mpc = true;
    return MPCX_OK;
}

int32_t mpcx_set_instance_tuning(mpcx_ctx *ctx, const mpcx_qp_tuning *rows, int32_t n_rows) {
    if (!ctx) return MPCX_E_INVALID;
    if ((rows == nullptr) != (n_rows == 0) || n_rows < 0)
        return mpcx_fail(ctx, MPCX_E_INVALID, "set_instance_tuning: rows and n_rows must both be given or both be empty");
    ctx->tune = rows;
    ctx->tune_rows = n_rows;
    return MPCX_OK;
}

int32_t mpcx_qp_set_order_hint(mpcx_ctx *ctx, const int32_t *prev_iters, const int32_t *ref_now, const int32_t *ref_prev) {
    if (!ctx) return MPCX_E_INVALID;
    if ((ref_now == nullptr) != (ref_prev == nullptr))
        return mpcx_fail(ctx, MPCX_E_INVALID, "qp_set_order_hint: ref_now and ref_prev come as a pair");
    ctx->order_hint = prev_iters;
    ctx->order_now = ref_now;
    ctx->order_prev = ref_prev;
    return MPCX_OK;
}

int32_t mpcx_set_qp_solver(mpcx_ctx *ctx, int32_t which) {
    if (!ctx) return MPCX_E_INVALID;
    if (which < 0 || which > 2) return mpcx_fail(ctx, MPCX_E_INVALID, "set_qp_solver: 0 (automatic), 1 (condensed) or 2 (stage-structured)");
    ctx->qp_solver = which;
    return MPCX_OK;
}

int32_t mpcx_set_linearisation_passes(mpcx_ctx *ctx, int32_t passes) {
    if (!ctx) return MPCX_E_INVALID;
    if (passes < 1 || passes > 16) return mpcx_fail(ctx, MPCX_E_INVALID, "set_linearisation_passes: 1..16 (lib/mpc.py MAX_ITER)");
    ctx->lin_passes = passes;
    return MPCX_OK;
}

int32_t mpcx_profile_qp(mpcx_ctx *ctx, int32_t enable) {
    if (!ctx) return MPCX_E_INVALID;
    ctx->prof_qp = enable != 0;
    return MPCX_OK;
}

int32_t mpcx_profile_qp_read(mpcx_ctx *ctx, double *total_ms, int32_t *launches) {
    if (!ctx || !total_ms || !launches) return MPCX_E_INVALID;
    double sum = 0.0;
    const size_t n = ctx->prof_ev.size() / 2;
    for (size_t i = 0; i < n; i++) {
        float ms = 0.f;
        if (hipEventSynchronize(ctx->prof_ev[2 * i + 1]) != hipSuccess ||
            hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]) != hipSuccess)
            return mpcx_fail(ctx, MPCX_E_LAUNCH, "profile_qp_read: event %zu could not be read", i);
        sum += ms;
    }
    ctx->prof_free.insert(ctx->prof_free.end(), ctx->prof_ev.begin(), ctx->prof_ev.end());
    ctx->prof_ev.clear();
    *total_ms = sum;
    *launches = (int32_t)n;
    return MPCX_OK;
}

}  // extern "C"
