// mpcx_expand_core.h -- the successor generation of lib/motion_primitive_search.py:87-121 (one (node, primitive) record: collision test
// of the primitive's template at the node + pose of its end point), shared by the batched expansion kernels (mpcx_expand.hip) and
// the device-resident search (mpcx_astar.hip).  See mpcx_expand.hip for the arithmetic notes.
#pragma once
#include "mpcx_common.h"

struct mpcx_search_model {
    int n_prim, n_obst, n_pts, n_rows;
    int32_t *d_tmpl_off, *d_hp_off;
    double *d_tmpl_xy, *d_last_pose, *d_edge_cost, *d_hp;
    double *d_aabb;     // per obstacle (xlo, xhi, ylo, yhi) implied by its rows of the form (+-1, 0, c) / (0, +-1, c); +-inf if none
    // rows NOT of that form, per obstacle (boxes: none; circle octagons: the four diagonals): the only rows the kernel evaluates
    // arithmetically -- a point passes every axis-aligned row iff it lies in the box above (exact, see expand_block)
    double *d_rest;
    int32_t *d_rest_off;
    int n_rest;
    double *d_tbox;     // per primitive (xlo, xhi, ylo, yhi, largest |coordinate|) of its template points: the bulk kernels' record boxes
    int div_magic;      // ceil(65536 / n_prim), or 0 if (x * magic) >> 16 != x / n_prim for some x < 256 (then the per-lane kernel runs)
};

namespace mpcx {

constexpr int EXP_MAX_ROWS = 512;   // half-plane rows staged in LDS
constexpr int EXP_MAX_PTS = 512;    // template points staged in LDS
constexpr int EXP_MAX_OBST = 128;

struct ExpandArgs {
    int n_prim, n_obst, n_pts, n_rows, n_nodes;
    const int32_t *tmpl_off, *hp_off;      // hp_off / hp: the NON-axis-aligned rows of every obstacle (mpcx_search_model::d_rest)
    const double *tmpl_xy, *last_pose, *edge_cost, *hp, *aabb, *nodes, *nodes_cs;
    double *nbr, *cost;
    uint8_t *collide;
    const double *tbox;     // per primitive: bounding box of its template points + their largest |coordinate| (xlo, xhi, ylo, yhi, r); nullptr: per-lane path
    int div_magic;          // ceil(65536 / n_prim): (x * div_magic) >> 16 == x / n_prim for x < 256 (checked by the host)
};

// maths.py:4-10 (python float %: result takes the sign of the divisor)
__device__ __forceinline__ double normalize_angle(double th) {
    const double tau = 6.283185307179586, pi = 3.141592653589793;
    if (!(fabs(th) < tau)) th = fmod(th, tau);       // |th| < tau: fmod returns th itself (exact); the sum of two headings is always there
    if (th < 0) th += tau;
    if (th >= tau) th = 0.0;
    if (th >= pi) th -= tau;
    return th;
}

struct ExpandTables {       // the search model's tables in LDS
    double hp[EXP_MAX_ROWS * 3];
    double xy[EXP_MAX_PTS * 2];
    int32_t hoff[EXP_MAX_OBST + 1];
    int32_t toff[MPCX_MAX_PRIM + 1];
    double aabb[EXP_MAX_OBST * 4];
};

// the same tables carved out of dynamic LDS at the model's own sizes (the bulk kernel: 3 KB instead of 25 for the stock intersection, which
// is the difference between four and eight resident workgroups per CU)
struct ExpandTablesView {
    double *hp, *xy, *aabb;
    int32_t *hoff, *toff;
};
static inline __host__ __device__ size_t expand_view_bytes(int n_rows, int n_pts, int n_obst, int n_prim) {
    return ((size_t)n_rows * 3 + (size_t)n_pts * 2 + (size_t)n_obst * 4) * sizeof(double) + ((size_t)(n_obst + 1 + n_prim + 1 + 1) / 2 * 2) * sizeof(int32_t);
}
__device__ __forceinline__ ExpandTablesView expand_view(const ExpandArgs &a, double *base) {
    ExpandTablesView v;
    v.hp = base; v.xy = v.hp + (size_t)a.n_rows * 3; v.aabb = v.xy + (size_t)a.n_pts * 2;
    v.hoff = reinterpret_cast<int32_t *>(v.aabb + (size_t)a.n_obst * 4); v.toff = v.hoff + a.n_obst + 1;
    return v;
}

template <class Tab>
__device__ __forceinline__ void expand_stage(const ExpandArgs &a, Tab &t) {
    // first batch: every load of the thread in flight before its first LDS store (one memory round trip per block instead of five)
    const int i = threadIdx.x, nb = blockDim.x;
    const int n_aabb = a.n_obst * 4, n_hp = a.n_rows * 3, n_xy = a.n_pts * 2;
    const double v0 = i < n_aabb ? a.aabb[i] : 0.0, v1 = i < n_hp ? a.hp[i] : 0.0, v2 = i < n_xy ? a.tmpl_xy[i] : 0.0;
    const int32_t v3 = i <= a.n_obst ? a.hp_off[i] : 0, v4 = i <= a.n_prim ? a.tmpl_off[i] : 0;
    if (i < n_aabb) t.aabb[i] = v0;
    if (i < n_hp) t.hp[i] = v1;
    if (i < n_xy) t.xy[i] = v2;
    if (i <= a.n_obst) t.hoff[i] = v3;
    if (i <= a.n_prim) t.toff[i] = v4;
    for (int j = i + nb; j < n_aabb; j += nb) t.aabb[j] = a.aabb[j];
    for (int j = i + nb; j < n_hp; j += nb) t.hp[j] = a.hp[j];
    for (int j = i + nb; j < n_xy; j += nb) t.xy[j] = a.tmpl_xy[j];
    for (int j = i + nb; j <= a.n_obst; j += nb) t.hoff[j] = a.hp_off[j];
    for (int j = i + nb; j <= a.n_prim; j += nb) t.toff[j] = a.tmpl_off[j];
    __syncthreads();
}

// does primitive k, placed at (tx, ty) with heading cosine / sine (c, s), put a collision point inside obstacle o?  (one obstacle of
// check_collision, obstacles.py:157-176, on the template points transformed as transform_2d_pts does)
template <class Tab>
__device__ __forceinline__ bool primitive_hits_obstacle(const Tab &t, int k, int o, double tx, double ty, double c, double s, int first = 0, int stride = 1) {
    // (first, stride): the points first, first + stride, ... of the template only -- the points are tested independently of each other, so
    // the bulk kernel spreads one (record, obstacle) pair over several lanes when a wavefront has few pairs
    const double *s_hp = t.hp, *s_xy = t.xy;
    const int pa = t.toff[k] + first, pb = t.toff[k + 1];
    const double *bx = t.aabb + 4 * o;
    const int r0 = t.hoff[o], r1 = t.hoff[o + 1];
    bool hit = false;
    for (int p0 = pa; p0 < pb && !hit; p0 += 32 * stride) {    // 32 points at a time (one mask word); the stock templates have 4 .. 14
        // which points pass ALL axis-aligned rows of this obstacle: a row (1, 0, c) is evaluated by the reference as
        // fl(wx + c) <= 0, which holds iff wx <= -c (rounding never changes the sign of a sum of two doubles), so "inside the box
        // the axis-aligned rows imply" is the same decision, taken here without branches
        unsigned inm = 0;
        int n = 0;
        for (int i = p0; i < pb && n < 32; i += stride, n++) {
            const double px = s_xy[2 * i], py = s_xy[2 * i + 1];
            const double wx = __dadd_rn(fma(py, -s, __dmul_rn(px, c)), tx);
            const double wy = __dadd_rn(fma(py, c, __dmul_rn(px, s)), ty);
            inm |= ((unsigned)(wx <= bx[1]) & (unsigned)(wx >= bx[0]) & (unsigned)(wy <= bx[3]) & (unsigned)(wy >= bx[2])) << n;
        }
        if (r0 == r1) { hit = inm != 0; continue; }            // a box: nothing else to test
        while (inm && !hit) {                                  // the remaining rows (octagon diagonals, general half-planes)
            const int i = p0 + (__ffs((int)inm) - 1) * stride;
            inm &= inm - 1;
            const double px = s_xy[2 * i], py = s_xy[2 * i + 1];
            // (x*m0 + y*m1) + t with the first product rounded and the second fused: the order OpenBLAS uses for N>=2 rows
            const double wx = __dadd_rn(fma(py, -s, __dmul_rn(px, c)), tx);
            const double wy = __dadd_rn(fma(py, c, __dmul_rn(px, s)), ty);
            bool inside = true;
            for (int r = r0; r < r1; r++) {
                const double v = __dadd_rn(__dadd_rn(__dmul_rn(s_hp[3 * r], wx), __dmul_rn(s_hp[3 * r + 1], wy)), s_hp[3 * r + 2]);
                if (!(v <= 0.0)) { inside = false; break; }
            }
            hit = inside;
        }
    }
    return hit;
}

// does primitive k, placed at the pose (x, y) with heading cosine / sine (c, s), hit an obstacle?  (check_collision over all obstacles,
// obstacles.py:157-176, on the template points transformed as transform_2d_pts does)
__device__ __forceinline__ bool primitive_collides(const ExpandArgs &a, const ExpandTables &t, int k, double tx, double ty, double c, double s) {
    const double *s_xy = t.xy, *s_aabb = t.aabb;
    const int32_t *s_toff = t.toff;
    // world-space collision points of this primitive: first their bounding box, then, per obstacle, an EXACT cull -- a row
    // (1, 0, c) is evaluated below as fl(wx + c) <= 0, which holds iff wx <= -c (a sum of two doubles is never rounded to
    // zero), so "every point has wx > -c" proves that no point passes that row; likewise for (-1, 0, c), (0, +-1, c).  Only
    // obstacles whose axis-aligned rows the box reaches run the per-point test (same arithmetic as before).
    const int p0 = s_toff[k], p1 = s_toff[k + 1];
    double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int i = p0; i < p1; i++) {
        const double px = s_xy[2 * i], py = s_xy[2 * i + 1];
        const double wx = __dadd_rn(fma(py, -s, __dmul_rn(px, c)), tx);
        const double wy = __dadd_rn(fma(py, c, __dmul_rn(px, s)), ty);
        xmin = fmin(xmin, wx); xmax = fmax(xmax, wx); ymin = fmin(ymin, wy); ymax = fmax(ymax, wy);
    }
    bool hit = false;
    // the cull runs branch-free over all obstacles (32 at a time) into a candidate mask; only the set bits are walked.  The kernel is
    // bound by instruction issue of divergent control flow (profiles/r02_expand_experiments.txt): a cull loop that `continue`s per
    // lane costs every lane of the wavefront the branch code of all 24 obstacles.
    for (int o0 = 0; o0 < a.n_obst && !hit; o0 += 32) {
      unsigned cand = 0;
      const int on = a.n_obst - o0 < 32 ? a.n_obst - o0 : 32;
      for (int j = 0; j < on; j++) {
          const double *bx = s_aabb + 4 * (o0 + j);
          const unsigned out = (unsigned)(xmin > bx[1]) | (unsigned)(xmax < bx[0]) | (unsigned)(ymin > bx[3]) | (unsigned)(ymax < bx[2]);
          cand |= (out ^ 1u) << j;
      }
      while (cand && !hit) {
        const int o = o0 + __ffs((int)cand) - 1;
        cand &= cand - 1;
        hit = primitive_hits_obstacle(t, k, o, tx, ty, c, s);
      }
    }
    return hit;
}

// the pose a primitive ends at (N == 1 row of transform_2d_pts: fma(x, m0, y*m1) + t; heading normalised as maths.py:4-10)
__device__ __forceinline__ void primitive_end_pose(const ExpandArgs &a, int k, double tx, double ty, double th, double c, double s, double *o3) {
    const double lx = a.last_pose[3 * k], ly = a.last_pose[3 * k + 1], lt = a.last_pose[3 * k + 2];
    o3[0] = __dadd_rn(fma(lx, c, __dmul_rn(ly, -s)), tx);
    o3[1] = __dadd_rn(fma(lx, s, __dmul_rn(ly, c)), ty);
    o3[2] = normalize_angle(__dadd_rn(lt, th));
}

// the records block_in_segment * 256 .. + 255 (one thread each) against the staged tables
__device__ __forceinline__ void expand_records(const ExpandArgs &a, const ExpandTables &t, unsigned block_in_segment) {
    const long long gid = (long long)block_in_segment * blockDim.x + threadIdx.x;
    const long long total = (long long)a.n_nodes * a.n_prim;
    if (gid >= total) return;
    const int node = (int)(gid / a.n_prim), k = (int)(gid % a.n_prim);
    const double x = a.nodes[3 * node], y = a.nodes[3 * node + 1], th = a.nodes[3 * node + 2];
    double s, c;
    if (a.nodes_cs) { c = a.nodes_cs[2 * node]; s = a.nodes_cs[2 * node + 1]; }   // host-supplied cos/sin (bit-identical to numpy)
    else sincos(th, &s, &c);
    const bool rot_only = (x == 0.0 && y == 0.0);     // linalg.py:13-17
    const double tx = rot_only ? 0.0 : x, ty = rot_only ? 0.0 : y;
    const bool hit = primitive_collides(a, t, k, tx, ty, c, s);
    primitive_end_pose(a, k, tx, ty, th, c, s, a.nbr + (size_t)gid * 3);
    a.cost[gid] = a.edge_cost[k];
    a.collide[gid] = hit ? 1 : 0;
}

// ---- bulk expansion: the wavefront works its (record, obstacle) pairs off together.
// On the section-8(d) frontier 80 % of the records reach no obstacle box, the rest one to seven: the per-lane loop above runs as long
// as the busiest lane of the wavefront (2.2 passes over <= 14 points on average, 17 live pairs among its 64 lanes x passes).  Here every
// lane culls its record against all obstacle boxes, the surviving pairs of the wavefront are queued in LDS and every lane takes ONE pair
// per pass -- same arithmetic per pair, so the flags are the same bits.  The box of a record comes from the four corners of its
// template's bounding box (interval arithmetic, inflated beyond any rounding difference) instead of all its points: a slightly larger box
// only lets a few more pairs through to the exact test.
constexpr int EXP_QCAP = 256;
struct ExpandCoop {
    double rec[4][4][64];                    // per wavefront: c, s, tx, ty of its records
    unsigned short queue[4][EXP_QCAP];       // per wavefront: lane | obstacle << 6
    unsigned char k[4][64], hit[4][64];
    double tbox[MPCX_MAX_PRIM][5];
};

__device__ __forceinline__ int expand_nodes_per_block(int n_prim) { return 256 / n_prim; }

template <class Tab>
__device__ __forceinline__ void expand_records_coop(const ExpandArgs &a, const Tab &t, ExpandCoop &co, unsigned block_in_segment) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int npb = expand_nodes_per_block(a.n_prim);
    const int nl = ((int)threadIdx.x * a.div_magic) >> 16, k = (int)threadIdx.x - nl * a.n_prim;      // node of the block, primitive
    const long long node = (long long)block_in_segment * npb + nl;
    const bool live = nl < npb && node < a.n_nodes;
    const long long ni = live ? node : 0;
    const double x = a.nodes[3 * ni], y = a.nodes[3 * ni + 1], th = a.nodes[3 * ni + 2];
    const double c = a.nodes_cs[2 * ni], s = a.nodes_cs[2 * ni + 1];      // the caller's cos / sin, or node_cs_kernel's: always there for a bulk launch
    const bool rot_only = (x == 0.0 && y == 0.0);     // linalg.py:13-17
    const double tx = rot_only ? 0.0 : x, ty = rot_only ? 0.0 : y;
    // box that contains every transformed template point: [xlo, xhi] x [ylo, yhi] rotated as intervals, + margin
    const double *tb = co.tbox[live ? k : 0];
    const double cx0 = c * tb[0], cx1 = c * tb[1], sy0 = s * tb[2], sy1 = s * tb[3];
    const double sx0 = s * tb[0], sx1 = s * tb[1], cy0 = c * tb[2], cy1 = c * tb[3];
    const double pad = 1e-12 * (1.0 + fabs(tx) + fabs(ty) + tb[4]);
    const double xmin = tx + fmin(cx0, cx1) - fmax(sy0, sy1) - pad, xmax = tx + fmax(cx0, cx1) - fmin(sy0, sy1) + pad;
    const double ymin = ty + fmin(sx0, sx1) + fmin(cy0, cy1) - pad, ymax = ty + fmax(sx0, sx1) + fmax(cy0, cy1) + pad;
    co.rec[w][0][lane] = c; co.rec[w][1][lane] = s; co.rec[w][2][lane] = tx; co.rec[w][3][lane] = ty;
    co.k[w][lane] = (unsigned char)(live ? k : 0);
    co.hit[w][lane] = 0;
    bool hit = false;
    for (int o0 = 0; o0 < a.n_obst; o0 += 32) {
        unsigned cand = 0;
        const int on = a.n_obst - o0 < 32 ? a.n_obst - o0 : 32;
        // the obstacle boxes come straight from memory: the index is wave-uniform, so these are SCALAR loads (constant cache, SGPR operands
        // of the compares) -- as LDS broadcasts they were 2 x 16 bytes x 64 lanes per obstacle and nearly half of the kernel's LDS time,
        // which is what bounds it (109 LDS instructions per wavefront, 1.6e7 per launch)
        typedef const __attribute__((address_space(4))) double cdouble;     // constant address space: read-only for the kernel's lifetime
        cdouble *gbox = (cdouble *)a.aabb;
        // four boxes per batch of scalar loads (the table is padded to a multiple of four with boxes nothing reaches); per box the four
        // compares narrow EXEC one after the other (v_cmpx), the lanes that are left set their candidate bit, EXEC is restored: 5 vector
        // + 2 scalar instructions per obstacle where the compiler's and-tree of four compare masks took 6 + 8.  !(a > b) forms throughout,
        // so a NaN coordinate keeps the obstacle (as the plain expression `!(xmin > bx1 | ...)` does).
        for (int j = 0; j < on; j += 4) {
            cdouble *bx = gbox + 4 * (o0 + j);
            double b[16];
#pragma unroll
            for (int q = 0; q < 16; q++) b[q] = bx[q];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                unsigned long long saved;
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             "v_cmpx_nlt_f64 vcc, %[b1], %[xmin]\n\t"
                             "v_cmpx_ngt_f64 vcc, %[b0], %[xmax]\n\t"
                             "v_cmpx_nlt_f64 vcc, %[b3], %[ymin]\n\t"
                             "v_cmpx_ngt_f64 vcc, %[b2], %[ymax]\n\t"
                             "v_or_b32 %[cand], %[bit], %[cand]\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [cand] "+v"(cand), [sv] "=&s"(saved)
                             : [b0] "s"(b[4 * q]), [b1] "s"(b[4 * q + 1]), [b2] "s"(b[4 * q + 2]), [b3] "s"(b[4 * q + 3]),
                               [xmin] "v"(xmin), [xmax] "v"(xmax), [ymin] "v"(ymin), [ymax] "v"(ymax), [bit] "s"(1u << (j + q))
                             : "vcc");
            }
        }
        if (!live) cand = 0;
        // queue positions: exclusive prefix of the candidate counts over the lanes
        const int cnt = __popc(cand);
        int pre = cnt;                                        // inclusive scan over the wavefront: row shifts + row broadcasts (DPP, no LDS)
        pre += __builtin_amdgcn_update_dpp(0, pre, 0x111, 0xF, 0xF, false);
        pre += __builtin_amdgcn_update_dpp(0, pre, 0x112, 0xF, 0xF, false);
        pre += __builtin_amdgcn_update_dpp(0, pre, 0x114, 0xF, 0xF, false);
        pre += __builtin_amdgcn_update_dpp(0, pre, 0x118, 0xF, 0xF, false);
        pre += __builtin_amdgcn_update_dpp(0, pre, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
        pre += __builtin_amdgcn_update_dpp(0, pre, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
        const int total = __builtin_amdgcn_readlane(pre, 63);
        if (total == 0) continue;                             // wave-uniform
        if (total > EXP_QCAP) {                               // wave-uniform, not seen on the benchmark: the per-lane loop
            while (cand && !hit) {
                const int o = o0 + __ffs((int)cand) - 1;
                cand &= cand - 1;
                hit = primitive_hits_obstacle(t, k, o, tx, ty, c, s);
            }
            continue;
        }
        int pos = pre - cnt;
        while (cand) {
            const int o = o0 + __ffs((int)cand) - 1;
            cand &= cand - 1;
            co.queue[w][pos++] = (unsigned short)(lane | (o << 6));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wavefront's own LDS writes are in order; keep the compiler from moving the reads up
        // few pairs: 2, 4 or 8 lanes share a pair, each taking every 2nd / 4th / 8th point of the template (the points are independent tests)
        const int sh = total <= 8 ? 3 : total <= 16 ? 2 : total <= 32 ? 1 : 0;
        for (int j = lane; j < (total << sh); j += 64) {
            const int e = co.queue[w][j >> sh], r = e & 63, o = e >> 6;
            if (primitive_hits_obstacle(t, co.k[w][r], o, co.rec[w][2][r], co.rec[w][3][r], co.rec[w][0][r], co.rec[w][1][r], j & ((1 << sh) - 1), 1 << sh))
                co.hit[w][r] = 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    hit = hit || co.hit[w][lane] != 0;
    if (!live) return;
    const long long gid = node * a.n_prim + k;
    primitive_end_pose(a, k, tx, ty, th, c, s, a.nbr + (size_t)gid * 3);
    a.cost[gid] = a.edge_cost[k];
    a.collide[gid] = hit ? 1 : 0;
}

}  // namespace mpcx
