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

__device__ __forceinline__ void expand_stage(const ExpandArgs &a, ExpandTables &t) {
    for (int i = threadIdx.x; i < a.n_obst * 4; i += blockDim.x) t.aabb[i] = a.aabb[i];
    for (int i = threadIdx.x; i < a.n_rows * 3; i += blockDim.x) t.hp[i] = a.hp[i];
    for (int i = threadIdx.x; i < a.n_pts * 2; i += blockDim.x) t.xy[i] = a.tmpl_xy[i];
    for (int i = threadIdx.x; i <= a.n_obst; i += blockDim.x) t.hoff[i] = a.hp_off[i];
    for (int i = threadIdx.x; i <= a.n_prim; i += blockDim.x) t.toff[i] = a.tmpl_off[i];
    __syncthreads();
}

// does primitive k, placed at the pose (x, y) with heading cosine / sine (c, s), hit an obstacle?  (check_collision over all obstacles,
// obstacles.py:157-176, on the template points transformed as transform_2d_pts does)
__device__ __forceinline__ bool primitive_collides(const ExpandArgs &a, const ExpandTables &t, int k, double tx, double ty, double c, double s) {
    const double *s_hp = t.hp, *s_xy = t.xy, *s_aabb = t.aabb;
    const int32_t *s_hoff = t.hoff, *s_toff = t.toff;
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
        const double *bx = s_aabb + 4 * o;
        // which points pass ALL axis-aligned rows of this obstacle: a row (1, 0, c) is evaluated by the reference as
        // fl(wx + c) <= 0, which holds iff wx <= -c (rounding never changes the sign of a sum of two doubles), so "inside the box
        // the axis-aligned rows imply" is the same decision, taken here without branches
        unsigned inm = 0;
        for (int i = p0; i < p1; i++) {
            const double px = s_xy[2 * i], py = s_xy[2 * i + 1];
            const double wx = __dadd_rn(fma(py, -s, __dmul_rn(px, c)), tx);
            const double wy = __dadd_rn(fma(py, c, __dmul_rn(px, s)), ty);
            inm |= ((unsigned)(wx <= bx[1]) & (unsigned)(wx >= bx[0]) & (unsigned)(wy <= bx[3]) & (unsigned)(wy >= bx[2])) << (i - p0);
        }
        const int r0 = s_hoff[o], r1 = s_hoff[o + 1];
        if (r0 == r1) { hit = inm != 0; continue; }           // a box: nothing else to test
        while (inm && !hit) {                                  // the remaining rows (octagon diagonals, general half-planes)
            const int i = p0 + __ffs((int)inm) - 1;
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

}  // namespace mpcx
