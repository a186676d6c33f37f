// mpcx_expand.hip -- batched motion-primitive successor generation with half-plane collision culling.
//
// Replaces lib/motion_primitive_search.py:87-121 `neighbor_function` (identical in _modified.py:101-135),
// lib/obstacles.py:157-176 `check_collision`, lib/linalg.py:4-54 and lib/maths.py:4-10
// (paths relative to /root/reference/main).  One thread per (node, primitive) pair; obstacle half-planes and
// primitive collision templates are staged once per block in LDS (read as wave-uniform broadcasts), successor
// poses/costs/flags are written as contiguous (node, primitive) records so stores coalesce.
// Row order inside an obstacle is kept (axis-aligned rows come first for both boxes and the circle octagons,
// obstacles.py:87-90,140-148), so the per-point early-out acts as an exact bounding-box cull.
#include "mpcx_common.h"
#include <cmath>
#include <cstring>
#include <vector>

#include "mpcx_expand_core.h"

namespace mpcx {

#ifndef MPCX_EXP_CHUNKS
#define MPCX_EXP_CHUNKS 1
#endif
constexpr int EXP_CHUNKS = MPCX_EXP_CHUNKS;

__device__ __forceinline__ void expand_block(const ExpandArgs &a, unsigned block_in_segment) {
    __shared__ ExpandTables t;
    expand_stage(a, t);
    expand_records(a, t, block_in_segment);          // 256 consecutive records per block, per-lane loops
}

// small launches (a search's frontier), any model size
__global__ __launch_bounds__(256) void expand_kernel(ExpandArgs a) { expand_block(a, blockIdx.x); }

// bulk launches: floor(256 / n_prim) nodes per block, the (record, obstacle) pairs of a wavefront worked off together
// (expand_records_coop), tables in dynamic LDS at the model's sizes
#ifndef MPCX_EXP_WAVES
#define MPCX_EXP_WAVES 6
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MPCX_EXP_WAVES, 8))) void expand_coop_kernel(ExpandArgs a) {
    extern __shared__ double s_dyn[];
    ExpandCoop &co = *reinterpret_cast<ExpandCoop *>(s_dyn);
    ExpandTablesView t = expand_view(a, s_dyn + sizeof(ExpandCoop) / sizeof(double));
    for (int i = threadIdx.x; i < a.n_prim * 5; i += blockDim.x) co.tbox[i / 5][i % 5] = a.tbox[i];
    expand_stage(a, t);                                   // ends with the block's barrier
    // EXP_CHUNKS groups of floor(256 / n_prim) nodes per block: the staging above is paid once for all of them (the wavefronts of a block
    // run their groups independently: nothing below is block-wide)
    const int npb = expand_nodes_per_block(a.n_prim);
    for (int j = 0; j < EXP_CHUNKS; j++) {
        const unsigned g = blockIdx.x * EXP_CHUNKS + j;
        if ((long long)g * npb >= a.n_nodes) break;
        expand_records_coop(a, t, co, g);
    }
}

// cos / sin of every node's heading, once per NODE: the records of a node (one per primitive) would otherwise each evaluate the
// same double-precision sincos, ~150 of the ~1000 instructions of a record in a kernel that is bound by instruction issue
__global__ __launch_bounds__(256) void node_cs_kernel(int n, const double *nodes, double *cs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s, c;
    sincos(nodes[3 * (size_t)i + 2], &s, &c);
    cs[2 * (size_t)i] = c; cs[2 * (size_t)i + 1] = s;
}

// several searches in one launch: segment s = the nodes of one search (its own obstacle / template tables and output ranges);
// every block belongs to exactly one segment, so the LDS staging of expand_block is unchanged
__global__ __launch_bounds__(256) void expand_multi_kernel(const ExpandArgs *segs, const int32_t *blk_seg, const int32_t *blk_first) {
    const int sg = blk_seg[blockIdx.x];
    const ExpandArgs a = segs[sg];
    expand_block(a, blockIdx.x - (unsigned)blk_first[sg]);
}

}  // namespace mpcx

template <typename Tp>
static Tp *to_device(const Tp *h, size_t n) {
    Tp *d = nullptr;
    if (hipMalloc((void **)&d, (n ? n : 1) * sizeof(Tp)) != hipSuccess) return nullptr;
    if (n && hipMemcpy(d, h, n * sizeof(Tp), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    return d;
}

extern "C" mpcx_search_model *mpcx_search_model_create(mpcx_ctx *ctx, int32_t n_prim, const int32_t *tmpl_off,
                                                       const double *tmpl_xy, const double *last_pose,
                                                       const double *edge_cost, int32_t n_obst,
                                                       const int32_t *hp_off, const double *hp) {
    if (!ctx) return nullptr;
    if (n_prim < 1 || n_prim > MPCX_MAX_PRIM || n_obst < 0 || n_obst > mpcx::EXP_MAX_OBST || !tmpl_off || !tmpl_xy ||
        !last_pose || !edge_cost || !hp_off || (n_obst > 0 && !hp)) {
        mpcx_fail(ctx, MPCX_E_INVALID, "search_model_create: bad sizes or null tables");
        return nullptr;
    }
    const int n_pts = tmpl_off[n_prim], n_rows = hp_off[n_obst];
    if (n_pts > mpcx::EXP_MAX_PTS || n_rows > mpcx::EXP_MAX_ROWS || n_pts < 0 || n_rows < 0) {
        mpcx_fail(ctx, MPCX_E_INVALID, "search_model_create: %d template points / %d half-plane rows exceed %d / %d",
                  n_pts, n_rows, mpcx::EXP_MAX_PTS, mpcx::EXP_MAX_ROWS);
        return nullptr;
    }
    mpcx_search_model *m = new mpcx_search_model();
    m->n_prim = n_prim; m->n_obst = n_obst; m->n_pts = n_pts; m->n_rows = n_rows;
    m->d_tmpl_off = to_device(tmpl_off, n_prim + 1);
    m->d_hp_off = to_device(hp_off, n_obst + 1);
    m->d_tmpl_xy = to_device(tmpl_xy, (size_t)n_pts * 2);
    m->d_last_pose = to_device(last_pose, (size_t)n_prim * 3);
    m->d_edge_cost = to_device(edge_cost, n_prim);
    m->d_hp = to_device(hp, (size_t)n_rows * 3);
    {
        // padded to a multiple of four boxes with boxes no record reaches (the bulk kernel culls four at a time)
        std::vector<double> box((size_t)((n_obst + 3) / 4 * 4 + 4) * 4, 0.0);
        for (size_t o = (size_t)n_obst; o < box.size() / 4; o++) { box[4 * o] = INFINITY; box[4 * o + 1] = -INFINITY; box[4 * o + 2] = INFINITY; box[4 * o + 3] = -INFINITY; }
        for (int o = 0; o < n_obst; o++) {
            double xlo = -INFINITY, xhi = INFINITY, ylo = -INFINITY, yhi = INFINITY;
            for (int r = hp_off[o]; r < hp_off[o + 1]; r++) {
                const double ra = hp[3 * r], rb = hp[3 * r + 1], rc = hp[3 * r + 2];
                if (ra == 1.0 && rb == 0.0) xhi = fmin(xhi, -rc);
                else if (ra == -1.0 && rb == 0.0) xlo = fmax(xlo, rc);
                else if (ra == 0.0 && rb == 1.0) yhi = fmin(yhi, -rc);
                else if (ra == 0.0 && rb == -1.0) ylo = fmax(ylo, rc);
            }
            box[4 * o] = xlo; box[4 * o + 1] = xhi; box[4 * o + 2] = ylo; box[4 * o + 3] = yhi;
        }
        m->d_aabb = to_device(box.data(), box.size());
        std::vector<double> rest;
        std::vector<int32_t> roff((size_t)n_obst + 1, 0);
        for (int o = 0; o < n_obst; o++) {
            for (int r = hp_off[o]; r < hp_off[o + 1]; r++) {
                const double ra = hp[3 * r], rb = hp[3 * r + 1];
                const bool axis = ((ra == 1.0 || ra == -1.0) && rb == 0.0) || (ra == 0.0 && (rb == 1.0 || rb == -1.0));
                if (!axis) { rest.push_back(ra); rest.push_back(rb); rest.push_back(hp[3 * r + 2]); }
            }
            roff[(size_t)o + 1] = (int32_t)(rest.size() / 3);
        }
        m->n_rest = (int)(rest.size() / 3);
        std::vector<double> tb((size_t)n_prim * 5);
        for (int k = 0; k < n_prim; k++) {
            double xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY, r = 0.0;
            for (int i = tmpl_off[k]; i < tmpl_off[k + 1]; i++) {
                const double px = tmpl_xy[2 * i], py = tmpl_xy[2 * i + 1];
                xlo = fmin(xlo, px); xhi = fmax(xhi, px); ylo = fmin(ylo, py); yhi = fmax(yhi, py);
                r = fmax(r, fmax(fabs(px), fabs(py)));
            }
            if (tmpl_off[k + 1] <= tmpl_off[k]) { xlo = xhi = ylo = yhi = 0.0; }      // no points: the box is never used for a hit
            tb[5 * k] = xlo; tb[5 * k + 1] = xhi; tb[5 * k + 2] = ylo; tb[5 * k + 3] = yhi; tb[5 * k + 4] = r;
        }
        m->d_tbox = to_device(tb.data(), tb.size());
        m->div_magic = (65536 + n_prim - 1) / n_prim;
        for (int x = 0; x < 256; x++)
            if (((x * m->div_magic) >> 16) != x / n_prim) m->div_magic = 0;
        m->d_rest = to_device(rest.data(), rest.size());
        m->d_rest_off = to_device(roff.data(), roff.size());
    }
    if (!m->d_tmpl_off || !m->d_hp_off || !m->d_tmpl_xy || !m->d_last_pose || !m->d_edge_cost || !m->d_hp || !m->d_aabb || !m->d_rest || !m->d_rest_off || !m->d_tbox) {
        mpcx_fail(ctx, MPCX_E_LAUNCH, "search_model_create: device allocation failed");
        mpcx_search_model_destroy(m);
        return nullptr;
    }
    return m;
}

extern "C" int32_t mpcx_expand_multi_batch(mpcx_ctx *ctx, int32_t n_seg, const mpcx_search_model *const *models, const int32_t *seg_off,
                                           const double *nodes, const double *nodes_cs, double *nbr, double *cost, uint8_t *collide) {
    if (!ctx) return MPCX_E_INVALID;
    if (n_seg == 0) return MPCX_OK;
    if (n_seg < 0 || !models || !seg_off) return mpcx_fail(ctx, MPCX_E_INVALID, "expand_multi_batch: null table or negative segment count");
    const int n_total = seg_off[n_seg];
    if (n_total == 0) return MPCX_OK;
    if (n_total < 0 || !nodes || !nbr || !cost || !collide) return mpcx_fail(ctx, MPCX_E_INVALID, "expand_multi_batch: null pointer");
    const int P = models[0] ? models[0]->n_prim : 0;
    std::vector<mpcx::ExpandArgs> segs((size_t)n_seg);
    std::vector<int32_t> blk_seg, blk_first((size_t)n_seg);
    for (int sg = 0; sg < n_seg; sg++) {
        const mpcx_search_model *m = models[sg];
        const int n0 = seg_off[sg], n = seg_off[sg + 1] - n0;
        if (!m || n < 0 || m->n_prim != P)
            return mpcx_fail(ctx, MPCX_E_INVALID, "expand_multi_batch: segment %d has no model, a negative size or another primitive count", sg);
        segs[sg] = mpcx::ExpandArgs{m->n_prim, m->n_obst, m->n_pts, m->n_rest, n, m->d_tmpl_off, m->d_rest_off, m->d_tmpl_xy, m->d_last_pose,
                                    m->d_edge_cost, m->d_rest, m->d_aabb, nodes + 3 * (size_t)n0, nodes_cs ? nodes_cs + 2 * (size_t)n0 : nullptr,
                                    nbr + 3 * (size_t)n0 * P, cost + (size_t)n0 * P, collide + (size_t)n0 * P};
        blk_first[sg] = (int32_t)blk_seg.size();
        const long long nb = ((long long)n * P + 255) / 256;
        blk_seg.insert(blk_seg.end(), (size_t)nb, sg);
    }
    if (blk_seg.empty()) return MPCX_OK;
    // one upload: [segment descriptors | block -> segment | first block of each segment]
    const size_t b0 = segs.size() * sizeof(mpcx::ExpandArgs), b1 = blk_seg.size() * sizeof(int32_t), b2 = blk_first.size() * sizeof(int32_t);
    const size_t need = b0 + b1 + b2;
    if (need > ctx->multi_cap) {
        if (ctx->multi) (void)hipFree(ctx->multi);
        ctx->multi = nullptr; ctx->multi_cap = 0;
        if (hipMalloc((void **)&ctx->multi, need * 2) != hipSuccess) return mpcx_fail(ctx, MPCX_E_LAUNCH, "expand_multi_batch: cannot allocate %zu bytes", need * 2);
        ctx->multi_cap = need * 2;
    }
    std::vector<unsigned char> host(need);
    memcpy(host.data(), segs.data(), b0); memcpy(host.data() + b0, blk_seg.data(), b1); memcpy(host.data() + b0 + b1, blk_first.data(), b2);
    if (hipMemcpyAsync(ctx->multi, host.data(), need, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)      // `host` goes out of scope: the staging copy must have left it
        return mpcx_fail(ctx, MPCX_E_LAUNCH, "expand_multi_batch: descriptor upload failed");
    const unsigned char *d = (const unsigned char *)ctx->multi;
    hipLaunchKernelGGL(mpcx::expand_multi_kernel, dim3((unsigned)blk_seg.size()), dim3(256), 0, ctx->stream,
                       (const mpcx::ExpandArgs *)d, (const int32_t *)(d + b0), (const int32_t *)(d + b0 + b1));
    return mpcx_check_launch(ctx, "expand_multi_kernel");
}

extern "C" void mpcx_search_model_destroy(mpcx_search_model *m) {
    if (!m) return;
    (void)hipFree(m->d_tmpl_off); (void)hipFree(m->d_hp_off); (void)hipFree(m->d_tmpl_xy);
    (void)hipFree(m->d_last_pose); (void)hipFree(m->d_edge_cost); (void)hipFree(m->d_hp); (void)hipFree(m->d_aabb);
    (void)hipFree(m->d_rest); (void)hipFree(m->d_rest_off); (void)hipFree(m->d_tbox);
    delete m;
}

extern "C" int32_t mpcx_expand_batch(mpcx_ctx *ctx, const mpcx_search_model *m, int32_t n_nodes, const double *nodes,
                                     const double *nodes_cs, double *nbr, double *cost, uint8_t *collide) {
    if (!ctx) return MPCX_E_INVALID;
    if (n_nodes == 0) return MPCX_OK;
    if (!m || n_nodes < 0 || !nodes || !nbr || !cost || !collide)
        return mpcx_fail(ctx, MPCX_E_INVALID, "expand_batch: null pointer or negative node count");
    if (n_nodes == 0) return MPCX_OK;
    if (!nodes_cs && n_nodes >= 4096) {        // bulk expansion with device trigonometry: one sincos per node in front of the records
        const size_t need = (size_t)n_nodes * 2 * sizeof(double);
        if (need > ctx->cs_cap) {
            if (ctx->cs) (void)hipFree(ctx->cs);
            ctx->cs = nullptr; ctx->cs_cap = 0;
            if (hipMalloc((void **)&ctx->cs, need) != hipSuccess) return mpcx_fail(ctx, MPCX_E_LAUNCH, "expand_batch: cannot allocate %zu bytes", need);
            ctx->cs_cap = need;
        }
        hipLaunchKernelGGL(mpcx::node_cs_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, ctx->stream, n_nodes, nodes, ctx->cs);
        nodes_cs = ctx->cs;
    }
    // bulk launches (>= 4096 nodes) take the cooperative records of mpcx_expand_core.h, small ones (a search's frontier) the per-lane loop
    const bool coop = n_nodes >= 4096 && m->div_magic != 0;
    mpcx::ExpandArgs a{m->n_prim, m->n_obst, m->n_pts, m->n_rest, n_nodes, m->d_tmpl_off, m->d_rest_off,
                       m->d_tmpl_xy, m->d_last_pose, m->d_edge_cost, m->d_rest, m->d_aabb, nodes, nodes_cs, nbr, cost, collide,
                       coop ? m->d_tbox : nullptr, m->div_magic};
    const long long total = (long long)n_nodes * m->n_prim;
    const int npb = 256 / m->n_prim;
    const unsigned blocks = coop ? (unsigned)(((long long)n_nodes + npb - 1) / npb) : (unsigned)((total + 255) / 256);
    static_assert(sizeof(mpcx::ExpandCoop) % sizeof(double) == 0, "the tables follow the wavefront scratch in dynamic LDS");
    if (coop) {
        const size_t lds = sizeof(mpcx::ExpandCoop) + mpcx::expand_view_bytes(m->n_rest, m->n_pts, m->n_obst, m->n_prim);
        hipLaunchKernelGGL(mpcx::expand_coop_kernel, dim3((blocks + mpcx::EXP_CHUNKS - 1) / mpcx::EXP_CHUNKS), dim3(256), lds, ctx->stream, a);
    } else
        hipLaunchKernelGGL(mpcx::expand_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a);
    return mpcx_check_launch(ctx, "expand_kernel");
}
