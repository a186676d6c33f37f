// mpcx_astar.hip -- device-resident best-first search over motion primitives: the open list, the closed set and the successor
// generation of MANY independent searches live on the GPU, one wavefront per search, no host work between expansions.
//
// Replaces (paths relative to /root/reference/main) lib/a_star.py:31-78 `AStar.run` together with
// lib/motion_primitive_search.py:64-75,87-121 (is_goal, distance_to_goal, neighbor_function),
// lib/motion_primitive_search_modified.py:80-89 (the heuristic every stock MPC scenario uses) and the weighted heuristic / edge terms of
// lib/motion_primitive_search_multi_lane.py:56-108,155-181,226-237, _roundabout.py:131-157,212 and _single_lane.py:145-162,218.
//
// Exactness.  The reference's pop order is the order of the Python tuples (g + h, g, node, predecessor) and its node identity is
// float equality of (x, y, theta), so every number that enters them must come out with the reference's bits:
//  * successor coordinates need cos / sin of the node's heading as numpy computes them: they are LOOKED UP in a table the host
//    fills with numpy (sorted by heading; the headings a search can reach are the closure of its start heading under the nine
//    primitive heading changes, a few 10^5 values to the depth a search goes).  A heading that is not in the table ends the search with
//    MPCX_ASTAR_MISS and the heading in `miss`: the host adds it and runs the search again;
//  * g is a chain of float additions, the goal test and the `base` heuristic are +, -, *, max, abs and a correctly rounded sqrt:
//    reproduced exactly with un-fused IEEE operations;
//  * the `modified` heuristic squares with Python's `**`, i.e. libm pow(x, 2.0), which is NOT always the correctly rounded x * x (one ulp
//    off for ~0.08 % of arguments).  The kernel uses x * x, logs the h of every push, the host re-evaluates them with Python
//    floats and hands the few that differ back as an override table (per search, sorted by node); the search is run again with it;
//  * the edge values of the multi-lane / roundabout / single-lane variants (steering change through Python's float %, 1 / distance to
//    the nearest half-plane with the row norms (a**2 + b**2)**0.5 supplied by the host, np.linalg.norm) enter g and therefore the key:
//    evaluated with un-fused IEEE operations, logged for EVERY free successor (pushed or not: a wrong edge value could also have
//    suppressed a push) and checked / overridden by the host the same way.
// The heap is 8-ary so that a wavefront sifts with one level per memory round trip (children compared by eight lanes, minimum by
// shuffles); keys are compared as the tuples are, field by field.
#include "mpcx_expand_core.h"
#include <vector>

namespace mpcx {

constexpr int HEAP_ARITY = 8;
constexpr int HE = 10;      // doubles per heap entry: f, g, node[3], pred[3], primitive id, (pad)
constexpr int TE = 8;       // doubles per closed-set entry: node[3], g, pred[3], primitive id
constexpr int PLE = 8;      // doubles per successor-log entry: node[3], h (NaN: not pushed), edge value, expansion index, primitive id, g

struct AstarArgs {
    ExpandArgs model;       // tables of this search's model (nodes / outputs unused)
    mpcx_astar_search sp;
    const double *hp_all;   // ALL half-plane rows of the model (a, b, c), n_rows_all of them: the obstacle-distance terms
    int n_rows_all;
};

struct AstarIO {
    int n_search;
    const AstarArgs *searches;
    int n_cs; const double *cs_theta, *cs_val;
    int n_ov; const double *ov_key, *ov_val;
    int heap_cap, table_cap, log_cap, push_cap, path_cap;
    double *heap, *table, *log, *push_log, *path, *cost, *miss;
    int32_t *status, *n_exp, *n_push, *path_len, *path_prim;
};

// (f, g, node, pred) < (f', g', node', pred') as Python compares the tuples
__device__ __forceinline__ bool key_less(const double *a, const double *b) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (a[i] < b[i]) return true;
        if (a[i] > b[i]) return false;
    }
    return false;
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long h) {
    h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL; h ^= h >> 33;
    return h;
}
__device__ __forceinline__ unsigned long long node_hash(double x, double y, double t) {
    // float equality is the reference's node identity: -0.0 and 0.0 are one key
    const unsigned long long a = (unsigned long long)__double_as_longlong(x + 0.0), b = (unsigned long long)__double_as_longlong(y + 0.0),
                             c = (unsigned long long)__double_as_longlong(t + 0.0);
    return mix64(a ^ mix64(b ^ mix64(c)));
}

// closed set: open addressing, empty slots hold NaN in their first field (the caller fills the table with NaN).  Returns the slot of the
// node, or the empty slot where it would go (found tells which), or -1 when the table is full.
__device__ __forceinline__ int table_find(const double *tab, int cap, double x, double y, double t, bool &found) {
    unsigned long long h = node_hash(x, y, t);
    for (int probe = 0; probe < cap; probe++) {
        const int slot = (int)((h + (unsigned long long)probe) & (unsigned long long)(cap - 1));
        const double *e = tab + (size_t)slot * TE;
        const double ex = e[0];
        if (ex != ex) { found = false; return slot; }
        if (ex == x && e[1] == y && e[2] == t) { found = true; return slot; }
    }
    found = false;
    return -1;
}

// sorted table lookup (ascending doubles): index of v or -1
__device__ __forceinline__ int sorted_find(const double *tab, int n, double v) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const double m = tab[mid];
        if (m == v) return mid;
        if (m < v) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}
// override table (one search's slice): rows (x, y, theta, kind) sorted as tuples
__device__ __forceinline__ int key4_find(const double *tab, int n, double x, double y, double t, double kind) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const double *m = tab + 4 * (size_t)mid;
        if (m[0] == x && m[1] == y && m[2] == t && m[3] == kind) return mid;
        const bool less = m[0] < x || (m[0] == x && (m[1] < y || (m[1] == y && (m[2] < t || (m[2] == t && m[3] < kind)))));
        if (less) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

// BoxObstacle.distance_to_point (obstacles.py:95-103): dx = max(x1 - x, 0, x - x2), ...; sqrt(dx*dx + dy*dy)
// hipcc contracts a * b + c into an fma by default and its __dmul_rn / __dadd_rn are plain operators; the reference's Python floats round
// every product.  Where a value's bits matter the product goes through an empty asm that makes it opaque to the contraction pass.
__device__ __forceinline__ double mul_rn(double a, double b) { double r = a * b; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ double box_distance(const double *box, double x, double y) {
    const double dx = fmax(fmax(__dadd_rn(box[0], -x), 0.0), __dadd_rn(x, -box[2]));
    const double dy = fmax(fmax(__dadd_rn(box[1], -y), 0.0), __dadd_rn(y, -box[3]));
    return __dsqrt_rn(__dadd_rn(mul_rn(dx, dx), mul_rn(dy, dy)));
}

// Python's float % for a positive divisor (floatobject.c float_rem: fmod, then the result takes the divisor's sign)
__device__ __forceinline__ double py_mod_pos(double x, double y) {
    double m = fmod(x, y);
    if (m != 0.0) { if (m < 0.0) m = __dadd_rn(m, y); } else m = 0.0;
    return m;
}
// calculate_steering_change_cost(current, next, 1.0) (_multi_lane.py:56-76): |((next - current + pi) % (2 pi)) - pi|
__device__ __forceinline__ double steering_change(double current_th, double next_th) {
    const double pi = 3.141592653589793, tau = 6.283185307179586;
    const double d = __dadd_rn(next_th, -current_th);
    return fabs(__dadd_rn(py_mod_pos(__dadd_rn(d, pi), tau), -pi));
}
// distance_to_nearest_obstacle (_multi_lane.py:78-108): min over every half-plane row of |a x + b y + c| / (a**2 + b**2)**0.5; the whole
// wavefront calls it with the same point, lane l takes rows l, l + 64, ...
__device__ __forceinline__ double nearest_obstacle(const double *rows4, int n_rows, double x, double y, int lane) {      // rows4: (a, b, c, norm) per row, in LDS
    double best = INFINITY;
    for (int r = lane; r < n_rows; r += WAVE) {
        const double v = fabs(__dadd_rn(__dadd_rn(mul_rn(rows4[4 * r], x), mul_rn(rows4[4 * r + 1], y)), rows4[4 * r + 2])) / rows4[4 * r + 3];
        best = v < best ? v : best;
    }
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const double o = __shfl_xor(best, d, WAVE);
        best = o < best ? o : best;
    }
    return best;
}
__host__ __device__ __forceinline__ bool needs_obstacle_term(const mpcx_astar_search &sp) {
    return sp.variant == MPCX_ASTAR_ROUNDABOUT || sp.variant == MPCX_ASTAR_SINGLE_LANE || (sp.variant == MPCX_ASTAR_MULTI_LANE && sp.wh[3] != 0.0);
}

// distance_to_goal of the search's variant at node (x, y, t); obst = 1 / distance to the nearest obstacle there (or 0 if the variant has no such term)
__device__ __forceinline__ double heuristic(const mpcx_astar_search &sp, double x, double y, double t, double obst) {
    if (sp.variant == MPCX_ASTAR_BASE) {            // motion_primitive_search.py:71-75
        const double dxy = box_distance(sp.goal_box, x, y);
        const double dth = fmax(0.0, __dadd_rn(fabs(__dadd_rn(t, -sp.goal_point[2])), -sp.allowed_dtheta));
        return __dadd_rn(dxy, mul_rn(2.7, dth));
    }
    // motion_primitive_search_modified.py:80-89 (x * x stands for Python's x ** 2: see the file header)
    const double ex = __dadd_rn(x, -sp.goal_point[0]), ey = __dadd_rn(y, -sp.goal_point[1]);
    const double dxy = __dsqrt_rn(__dadd_rn(mul_rn(ex, ex), mul_rn(ey, ey)));
    const double ad = fabs(__dadd_rn(t, -sp.goal_point[2]));
    const double alt = __dadd_rn(ad, -__ddiv_rn(sp.allowed_dtheta, 2.0));
    const double dth = alt < ad ? alt : ad;          // Python's min(a, b): b only if b < a
    if (sp.variant == MPCX_ASTAR_MODIFIED || sp.variant == MPCX_ASTAR_ROUNDABOUT)     // _roundabout.py:155: the same two terms
        return __dadd_rn(dxy, mul_rn(2.7, dth));
    const double steer = steering_change(t, sp.goal_point[2]);
    if (sp.variant == MPCX_ASTAR_SINGLE_LANE)       // _single_lane.py:160
        return __dadd_rn(__dadd_rn(dxy, mul_rn(2.7, dth)), mul_rn(15.0, steer));
    // _multi_lane.py:155-181, terms added left to right
    const double centre = sp.wh[4] != 0.0 ? __dsqrt_rn(__dadd_rn(mul_rn(x, x), mul_rn(y, y))) : 0.0;
    double h = __dadd_rn(mul_rn(sp.wh[0], dxy), mul_rn(sp.wh[1], dth));
    h = __dadd_rn(h, mul_rn(sp.wh[2], steer));
    h = __dadd_rn(h, mul_rn(sp.wh[3], obst));
    return __dadd_rn(h, mul_rn(sp.wh[4], centre));
}

// the edge value neighbor_function yields for primitive k (length `len`) from heading nth to the successor (sx, sy, sth)
__device__ __forceinline__ double edge_value(const mpcx_astar_search &sp, double len, double nth, double sx, double sy, double sth, double obst) {
    if (sp.variant == MPCX_ASTAR_BASE || sp.variant == MPCX_ASTAR_MODIFIED) return len;       // motion_primitive_search.py:118
    const double steer = steering_change(nth, sth);
    if (sp.variant == MPCX_ASTAR_ROUNDABOUT)        // _roundabout.py:212: length + 0.1 * obstacle + 5 * steering
        return __dadd_rn(__dadd_rn(len, mul_rn(0.1, obst)), mul_rn(5.0, steer));
    if (sp.variant == MPCX_ASTAR_SINGLE_LANE)       // _single_lane.py:218: length + 5 * steering + 0.1 * obstacle
        return __dadd_rn(__dadd_rn(len, mul_rn(5.0, steer)), mul_rn(0.1, obst));
    // _multi_lane.py:226-237 (np.linalg.norm([x, y]) taken as sqrt(x*x + y*y); the host check corrects the rare ulp)
    const double centre = sp.wc[3] != 0.0 ? __dsqrt_rn(__dadd_rn(mul_rn(sx, sx), mul_rn(sy, sy))) : 0.0;
    double e = __dadd_rn(mul_rn(sp.wc[0], len), mul_rn(sp.wc[1], steer));
    e = __dadd_rn(e, mul_rn(sp.wc[2], obst));
    return __dadd_rn(e, mul_rn(sp.wc[3], centre));
}

__global__ __launch_bounds__(64) void astar_kernel(AstarIO io) {
    __shared__ ExpandTables t;
    const int sidx = blockIdx.x, lane = threadIdx.x;
    const AstarArgs sa = io.searches[sidx];
    const ExpandArgs &a = sa.model;
    const mpcx_astar_search &sp = sa.sp;
    expand_stage(a, t);
    // searches with an obstacle-distance term: ALL half-plane rows of the model with the host's row norms, staged once (dynamic LDS, sized by
    // the launch for the largest model that needs it; nine successors per expansion read every row)
    extern __shared__ double s_rows4[];
    if (needs_obstacle_term(sp)) {
        for (int r = lane; r < sa.n_rows_all; r += WAVE) {
            s_rows4[4 * r] = sa.hp_all[3 * r]; s_rows4[4 * r + 1] = sa.hp_all[3 * r + 1]; s_rows4[4 * r + 2] = sa.hp_all[3 * r + 2];
            s_rows4[4 * r + 3] = sp.hp_norm[r];
        }
        __syncthreads();
    }
    double *heap = io.heap + (size_t)sidx * io.heap_cap * HE;
    double *tab = io.table + (size_t)sidx * io.table_cap * TE;
    double *log = io.log + (size_t)sidx * io.log_cap * 8;
    double *plog = io.push_log + (size_t)sidx * io.push_cap * PLE;
    // pessimistic until an exit says otherwise: a loop that runs out of passes has run out of room, not out of nodes
    int n_heap = 0, n_exp = 0, n_push = 0, status = MPCX_ASTAR_CAPACITY;
    const bool log_all = sp.variant >= MPCX_ASTAR_MULTI_LANE;     // variants whose edge values are computed: every free successor is logged
    const bool want_obst = needs_obstacle_term(sp);
    const bool use_ov = sp.ov_cnt > 0 && sp.variant != MPCX_ASTAR_BASE;
    const double *ovk = io.ov_key + 4 * (size_t)sp.ov_off, *ovv = io.ov_val + (size_t)sp.ov_off;
    double result_cost = 0.0, goal_node[3] = {0.0, 0.0, 0.0};

    // q = [(0, 0, start, start)]
    if (lane == 0) {
        double *e = heap;
        e[0] = 0.0; e[1] = 0.0; e[2] = sp.start[0]; e[3] = sp.start[1]; e[4] = sp.start[2];
        e[5] = sp.start[0]; e[6] = sp.start[1]; e[7] = sp.start[2]; e[8] = -1.0; e[9] = 0.0;
    }
    n_heap = 1;
    __syncthreads();

    for (long guard = 0; guard < (long)io.push_cap + 2; guard++) {      // every pass pops one entry: bounded by the pushes the log can hold (+ the start)
        if (n_heap == 0) { status = MPCX_ASTAR_EXHAUSTED; break; }
        // ---------------------------------------------------------------- heappop: the root leaves, the last entry sifts down from the root
        double top[HE];
#pragma unroll
        for (int i = 0; i < HE; i++) top[i] = heap[i];
        n_heap--;
        if (n_heap > 0) {
            double mv[HE];
#pragma unroll
            for (int i = 0; i < HE; i++) mv[i] = heap[(size_t)n_heap * HE + i];
            int pos = 0;
            for (;;) {
                const int c0 = pos * HEAP_ARITY + 1;
                if (c0 >= n_heap) break;
                // lanes 0..7 hold the children (entries beyond the heap: +inf keys), the smallest is found by three shuffle steps
                const int ci = c0 + (lane & 7);
                double ck[8];
                const bool have = ci < n_heap;
#pragma unroll
                for (int i = 0; i < 8; i++) ck[i] = have ? heap[(size_t)ci * HE + i] : INFINITY;
                int best = ci;
#pragma unroll
                for (int d = 1; d < 8; d <<= 1) {
                    double ok[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) ok[i] = __shfl_xor(ck[i], d, WAVE);
                    const int ob = __shfl_xor(best, d, WAVE);
                    const bool take = key_less(ok, ck) || (!key_less(ck, ok) && ob < best);
                    if (take) {
#pragma unroll
                        for (int i = 0; i < 8; i++) ck[i] = ok[i];
                        best = ob;
                    }
                }
                // every lane of an eight-lane group now holds the group's minimum; group 0 is the one that loaded real children
                double bk[8];
#pragma unroll
                for (int i = 0; i < 8; i++) bk[i] = __shfl(ck[i], 0, WAVE);
                best = __shfl(best, 0, WAVE);
                if (!key_less(bk, mv)) break;
                if (lane < HE) heap[(size_t)pos * HE + lane] = heap[(size_t)best * HE + lane];
                __syncthreads();
                pos = best;
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < HE; i++) heap[(size_t)pos * HE + i] = mv[i];
            }
            __syncthreads();
        }
        const double g = top[1], nx = top[2], ny = top[3], nth = top[4];
        // ---------------------------------------------------------------- seen before with a g at least as good: skip (a_star.py:45-49)
        bool found;
        const int slot = table_find(tab, io.table_cap, nx, ny, nth, found);
        if (slot < 0) { status = MPCX_ASTAR_CAPACITY; break; }
        if (found && g >= tab[(size_t)slot * TE + 3]) continue;
        if (n_exp >= io.log_cap || n_exp >= sp.max_expansions) { status = MPCX_ASTAR_CAPACITY; break; }
        if (lane == 0) {
            double *lg = log + (size_t)n_exp * 8;               // node, g, h = f - g, predecessor (a_star.py:52)
            lg[0] = nx; lg[1] = ny; lg[2] = nth; lg[3] = g; lg[4] = __dadd_rn(top[0], -g); lg[5] = top[5]; lg[6] = top[6]; lg[7] = top[7];
            double *te = tab + (size_t)slot * TE;               // pred_dict[node] = g, predecessor (+ the primitive that led here)
            te[0] = nx; te[1] = ny; te[2] = nth; te[3] = g; te[4] = top[5]; te[5] = top[6]; te[6] = top[7]; te[7] = top[8];
        }
        n_exp++;
        __syncthreads();
        // ---------------------------------------------------------------- goal test at pop time (motion_primitive_search.py:64-69: no angle wrapping)
        if (box_distance(sp.goal_box, nx, ny) <= 1e-5 && fabs(__dadd_rn(nth, -sp.goal_point[2])) <= sp.allowed_dtheta) {
            status = MPCX_ASTAR_FOUND; result_cost = g; goal_node[0] = nx; goal_node[1] = ny; goal_node[2] = nth;
            break;
        }
        // ---------------------------------------------------------------- neighbours: lane k tests primitive k
        const int ic = sorted_find(io.cs_theta, io.n_cs, nth);
        if (ic < 0) { status = MPCX_ASTAR_MISS; if (lane == 0) io.miss[sidx] = nth; break; }
        const double c = io.cs_val[2 * (size_t)ic], s = io.cs_val[2 * (size_t)ic + 1];
        const bool rot_only = (nx == 0.0 && ny == 0.0);     // linalg.py:13-17
        const double tx = rot_only ? 0.0 : nx, ty = rot_only ? 0.0 : ny;
        bool hit = true;
        double pose[3] = {0.0, 0.0, 0.0};
        if (lane < a.n_prim) {
            hit = primitive_collides(a, t, lane, tx, ty, c, s);
            primitive_end_pose(a, lane, tx, ty, nth, c, s, pose);
        }
        bool overflow = false;
        for (int k = 0; k < a.n_prim; k++) {                   // in the order the reference's neighbour generator yields
            const bool free_k = __shfl((int)hit, k, WAVE) == 0;
            const double sx = __shfl(pose[0], k, WAVE), sy = __shfl(pose[1], k, WAVE), sth = __shfl(pose[2], k, WAVE);
            if (!free_k) continue;
            double obst = 0.0;
            if (want_obst) {
                const double dn = nearest_obstacle(s_rows4, sa.n_rows_all, sx, sy, lane);
                obst = dn != 0.0 ? 1.0 / dn : INFINITY;          // 1 / d if d else float('inf')
            }
            double edge = edge_value(sp, a.edge_cost[k], nth, sx, sy, sth, obst);
            if (use_ov && log_all) {
                const int ov = key4_find(ovk, sp.ov_cnt, nx, ny, nth, (double)k);
                if (ov >= 0) edge = ovv[ov];
            }
            const double ng = __dadd_rn(g, edge);                // neighbor_g = g + edge_value
            int li = -1;
            if (log_all) {
                if (n_push >= io.push_cap) { overflow = true; break; }
                li = n_push++;
                if (lane == 0) {
                    double *pl = plog + (size_t)li * PLE;
                    pl[0] = sx; pl[1] = sy; pl[2] = sth; pl[3] = NAN; pl[4] = edge; pl[5] = (double)(n_exp - 1); pl[6] = (double)k; pl[7] = ng;
                }
            }
            bool seen;
            const int sl = table_find(tab, io.table_cap, sx, sy, sth, seen);
            if (sl < 0) { overflow = true; break; }
            if (seen && !(ng < tab[(size_t)sl * TE + 3])) continue;
            double h = heuristic(sp, sx, sy, sth, obst);
            if (use_ov) {
                const int ov = key4_find(ovk, sp.ov_cnt, sx, sy, sth, -1.0);
                if (ov >= 0) h = ovv[ov];
            }
            if (n_heap >= io.heap_cap) { overflow = true; break; }
            if (!log_all) {
                if (n_push >= io.push_cap) { overflow = true; break; }
                li = n_push++;
                if (lane == 0) {
                    double *pl = plog + (size_t)li * PLE;
                    pl[0] = sx; pl[1] = sy; pl[2] = sth; pl[3] = h; pl[4] = edge; pl[5] = (double)(n_exp - 1); pl[6] = (double)k; pl[7] = ng;
                }
            } else if (lane == 0) {
                plog[(size_t)li * PLE + 3] = h;
            }
            // heappush: sift up from the end
            double nk[HE] = {__dadd_rn(ng, h), ng, sx, sy, sth, nx, ny, nth, (double)k, 0.0};
            int pos = n_heap++;
            while (pos > 0) {
                const int par = (pos - 1) / HEAP_ARITY;
                double pk[8];
#pragma unroll
                for (int i = 0; i < 8; i++) pk[i] = heap[(size_t)par * HE + i];
                if (!key_less(nk, pk)) break;
                if (lane < HE) heap[(size_t)pos * HE + lane] = heap[(size_t)par * HE + lane];
                __syncthreads();
                pos = par;
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < HE; i++) heap[(size_t)pos * HE + i] = nk[i];
            }
            __syncthreads();
        }
        if (overflow) { status = MPCX_ASTAR_CAPACITY; break; }
    }

    // ---------------------------------------------------------------- result: the path back through the predecessors (a_star.py:60-68), goal first
    int plen = 0;
    if (status == MPCX_ASTAR_FOUND && lane == 0) {
        double *path = io.path + (size_t)sidx * io.path_cap * 3;
        int32_t *pprim = io.path_prim + (size_t)sidx * io.path_cap;
        double cx = goal_node[0], cy = goal_node[1], cth = goal_node[2];
        bool at_start = false;
        for (; plen < io.path_cap; plen++) {
            bool f;
            const int sl = table_find(tab, io.table_cap, cx, cy, cth, f);
            path[3 * plen] = cx; path[3 * plen + 1] = cy; path[3 * plen + 2] = cth;
            pprim[plen] = (f && sl >= 0) ? (int32_t)tab[(size_t)sl * TE + 7] : -1;
            if (!f || sl < 0) break;
            if (cx == sp.start[0] && cy == sp.start[1] && cth == sp.start[2]) { plen++; at_start = true; break; }
            const double *e = tab + (size_t)sl * TE;
            cx = e[4]; cy = e[5]; cth = e[6];
        }
        if (!at_start) status = MPCX_ASTAR_PATH_CAPACITY;       // longer than path_cap (or a broken chain): never a silently truncated path
    }
    if (lane == 0) {
        io.status[sidx] = status; io.n_exp[sidx] = n_exp; io.n_push[sidx] = n_push; io.cost[sidx] = result_cost; io.path_len[sidx] = plen;
    }
}

}  // namespace mpcx

extern "C" int32_t mpcx_astar_batch(mpcx_ctx *ctx, int32_t n_search, const mpcx_search_model *const *models, const mpcx_astar_search *searches,
                                    int32_t n_cs, const double *cs_theta, const double *cs_val,
                                    int32_t n_ov, const double *ov_key, const double *ov_val,
                                    const mpcx_astar_buffers *b) {
    if (!ctx) return MPCX_E_INVALID;
    if (n_search == 0) return MPCX_OK;
    if (n_search < 0 || !models || !searches || !b || n_cs < 0 || (n_cs > 0 && (!cs_theta || !cs_val)) || n_ov < 0 || (n_ov > 0 && (!ov_key || !ov_val)))
        return mpcx_fail(ctx, MPCX_E_INVALID, "astar_batch: null table or negative count");
    if (b->heap_cap < 16 || b->table_cap < 16 || (b->table_cap & (b->table_cap - 1)) || b->log_cap < 1 || b->push_cap < 1 || b->path_cap < 2 ||
        !b->heap || !b->table || !b->log || !b->push_log || !b->path || !b->path_prim || !b->cost || !b->miss || !b->status || !b->n_exp || !b->n_push || !b->path_len)
        return mpcx_fail(ctx, MPCX_E_INVALID, "astar_batch: buffers missing, capacities too small or table_cap not a power of two");
    std::vector<mpcx::AstarArgs> host((size_t)n_search);
    int max_obst_rows = 0;
    for (int i = 0; i < n_search; i++) {
        const mpcx_search_model *m = models[i];
        const mpcx_astar_search &sp = searches[i];
        if (!m) return mpcx_fail(ctx, MPCX_E_INVALID, "astar_batch: search %d has no model", i);
        if (m->n_prim > 64) return mpcx_fail(ctx, MPCX_E_INVALID, "astar_batch: more than 64 primitives");
        if (sp.variant < MPCX_ASTAR_BASE || sp.variant > MPCX_ASTAR_SINGLE_LANE)
            return mpcx_fail(ctx, MPCX_E_INVALID, "astar_batch: search %d: unknown variant %d", i, sp.variant);
        if (sp.ov_cnt < 0 || sp.ov_off < 0 || (long)sp.ov_off + sp.ov_cnt > n_ov)
            return mpcx_fail(ctx, MPCX_E_INVALID, "astar_batch: search %d: override slice [%d, %d) outside the table of %d rows", i, sp.ov_off, sp.ov_off + sp.ov_cnt, n_ov);
        if (mpcx::needs_obstacle_term(sp) && m->n_rows > 0 && !sp.hp_norm)
            return mpcx_fail(ctx, MPCX_E_INVALID, "astar_batch: search %d: the variant has an obstacle-distance term and needs hp_norm", i);
        host[i].model = mpcx::ExpandArgs{m->n_prim, m->n_obst, m->n_pts, m->n_rest, 0, m->d_tmpl_off, m->d_rest_off, m->d_tmpl_xy, m->d_last_pose,
                                         m->d_edge_cost, m->d_rest, m->d_aabb, nullptr, nullptr, nullptr, nullptr, nullptr};
        host[i].sp = sp;
        host[i].hp_all = m->d_hp;
        host[i].n_rows_all = m->n_rows;
        if (mpcx::needs_obstacle_term(sp) && m->n_rows > max_obst_rows) max_obst_rows = m->n_rows;
    }
    if (max_obst_rows > 1024) return mpcx_fail(ctx, MPCX_E_INVALID, "astar_batch: %d half-plane rows exceed the 1024 the obstacle-distance term stages in LDS", max_obst_rows);
    const size_t need = host.size() * sizeof(mpcx::AstarArgs);
    if (need > ctx->multi_cap) {
        if (ctx->multi) (void)hipFree(ctx->multi);
        ctx->multi = nullptr; ctx->multi_cap = 0;
        if (hipMalloc((void **)&ctx->multi, need * 2) != hipSuccess) return mpcx_fail(ctx, MPCX_E_LAUNCH, "astar_batch: cannot allocate %zu bytes", need * 2);
        ctx->multi_cap = need * 2;
    }
    if (hipMemcpyAsync(ctx->multi, host.data(), need, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)      // `host` goes out of scope
        return mpcx_fail(ctx, MPCX_E_LAUNCH, "astar_batch: descriptor upload failed");
    mpcx::AstarIO io{n_search, (const mpcx::AstarArgs *)ctx->multi, n_cs, cs_theta, cs_val, n_ov, ov_key, ov_val,
                     b->heap_cap, b->table_cap, b->log_cap, b->push_cap, b->path_cap, b->heap, b->table, b->log, b->push_log, b->path, b->cost, b->miss,
                     b->status, b->n_exp, b->n_push, b->path_len, b->path_prim};
    hipLaunchKernelGGL(mpcx::astar_kernel, dim3(n_search), dim3(64), (size_t)max_obst_rows * 4 * sizeof(double), ctx->stream, io);
    return mpcx_check_launch(ctx, "astar_kernel");
}
