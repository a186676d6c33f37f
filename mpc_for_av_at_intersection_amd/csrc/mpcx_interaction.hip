// mpcx_interaction.hip -- ego-vs-moving-cars conflict search and reference-path cut, batched.
//
// Replaces (paths relative to /root/reference/main):
//   lib/moving_obstacles_prediction.py:21-47  MovingObstaclesPrediction.state_prediction -> predict_kernel
//   scenarios/mpc_intersection.py:103-116     nearest index on the full path + ego prediction resampling
//   lib/trajectories.py:58-86                 resample_curve (vector dl)
//   lib/collision_avoidance.py:66-104         check_collision_moving_cars
//   lib/collision_avoidance.py:107-119 + mpc_intersection.py:129-136  cut-off index
// One wavefront per ego.  The reference flattens (frame, agent disc, obstacle x frame-offset, obstacle disc)
// into one pair table and takes the first row within 2*radius; here every obstacle disc position (<= 16*64*2,
// one global load each) is culled exactly against the bounding box of all ego discs and then against the boxes
// of 8 runs of ego frames, compared only with the frames of surviving runs that a +-frame_window shift can
// reach, and the FIRST ROW IN THE REFERENCE'S ORDER is recovered as the minimum of an integer key over all hits
// (bit-exact index outputs).
#include "mpcx_common.h"
#include <type_traits>

namespace mpcx {

struct PredArgs {
    mpcx_interaction_params ip;
    int n;
    const double *obs6;
    double *pred;    // [n][steps][2][2]
    // closed loop, local pool: the pool row of agent o is packed here from its state and applied inputs (MovingObstacle*.get():
    // x, y, v, yaw, a, steer) instead of by a launch of its own; nullptr: obs6 is read as it is
    const double *pack_state, *pack_applied;
    double *pack_out;
};

// moving_obstacles_prediction.py:21-28: v is updated BEFORE yaw; disc centres as trajectories.py:11-37
__global__ __launch_bounds__(256) void predict_kernel(PredArgs a) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= a.n) return;
    double x, y, v, yaw, acc, steer;
    if (a.pack_state) {
        const double *st = a.pack_state + 4 * (size_t)o;
        x = st[0]; y = st[1]; v = st[2]; yaw = st[3]; acc = a.pack_applied[2 * o + 1]; steer = a.pack_applied[2 * o];
        double *row = a.pack_out + 6 * (size_t)o;
        row[0] = x; row[1] = y; row[2] = v; row[3] = yaw; row[4] = acc; row[5] = steer;
    } else {
        const double *s6 = a.obs6 + 6 * (size_t)o;
        x = s6[0]; y = s6[1]; v = s6[2]; yaw = s6[3]; acc = s6[4]; steer = s6[5];
    }
    const double tn = tan(steer);
    const double dt = a.ip.dt;
    double s, c;
    sincos(yaw, &s, &c);
    double *out = a.pred + (size_t)o * a.ip.pred_steps * 4;
    for (int k = 0; k < a.ip.pred_steps; k++) {
        x = __dadd_rn(x, __dmul_rn(__dmul_rn(v, c), dt));
        y = __dadd_rn(y, __dmul_rn(__dmul_rn(v, s), dt));
        v = __dadd_rn(v, __dmul_rn(acc, dt));
        yaw = __dadd_rn(yaw, __dmul_rn(__dmul_rn(__ddiv_rn(v, a.ip.L), tn), dt));
        sincos(yaw, &s, &c);
#pragma unroll
        for (int d = 0; d < 2; d++) {
            const double cx = a.ip.circle_centers[2 * d], cy = a.ip.circle_centers[2 * d + 1];
            out[4 * k + 2 * d] = __dadd_rn(__dadd_rn(__dmul_rn(c, cx), -__dmul_rn(s, cy)), x);
            out[4 * k + 2 * d + 1] = __dadd_rn(__dadd_rn(__dmul_rn(s, cx), __dmul_rn(c, cy)), y);
        }
    }
}

struct InterArgs {
    mpcx_interaction_params ip;
    int P;
    const double *state, *path, *path_cs;
    const int32_t *path_off, *path_len, *prev_cut;
    const double *pred;
    const int32_t *obs_off, *obs_cnt, *obs_skip;
    int32_t *traj_idx, *hit_idx;
    double *hit_xy;
    int32_t *cut_len;
    int max_rem, fcap;    // capacity of this launch: path points ahead of an agent, resampled ego poses (dynamic LDS)
    int32_t *prev_save;   // optional: the previous cut length as read (cut_len may alias prev_cut and is overwritten)
    const int32_t *bin_hint;   // optional (closed loop): file the agent under its work-queue key: bin_cnt[p % COPIES][key]++ -> slot, keyslot[p] = key << 24 | slot
    int32_t *bin_cnt, *keyslot;
    int32_t *near;        // optional (closed loop): near[3p] = the start index of this agent's nearest-index scan, near[3p+1] / [3p+2] = the largest / smallest of its three nearest indices (absolute), -1 = none
};

__device__ __forceinline__ double dist2d(double ax, double ay, double bx, double by) {
    const double dx = __dadd_rn(ax, -bx), dy = __dadd_rn(ay, -by);
    return __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
}
// dist2d(a,b) <= md, decided from the squared distance except within 1e-12 (relative) of the threshold, where the
// reference's own expression sqrt(dx*dx + dy*dy) <= md is evaluated: identical decisions, no sqrt on the bulk of the pairs
__device__ __forceinline__ bool within(double ax, double ay, double bx, double by, double md, double md2lo, double md2hi) {
    const double dx = __dadd_rn(ax, -bx), dy = __dadd_rn(ay, -by);
    const double d2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
    if (d2 > md2hi) return false;
    if (d2 < md2lo) return true;
    return __dsqrt_rn(d2) <= md;
}
__device__ __forceinline__ long long wave_min_ll(long long v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { long long o = __shfl_xor(v, s, WAVE); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { int o = __shfl_xor(v, s, WAVE); v = o < v ? o : v; }
    return v;
}

// collision_avoidance.py:72-104 on prepared disc tables.  s_ego: ego disc centres of the `na` predicted poses;
// pred: obstacle disc centres [pool][steps][2][2]; (rem, rcs, n): the detailed path and cos/sin of its yaw.
// Returns the index of the earliest conflicting pose on the detailed path (and its x,y) or -1 (None).
//
// Work distribution.  The reference's row order is frame-major, so the answer lies in the FIRST run of ego frames that has any hit:
// runs are visited in order and the search stops after the first run with a hit.  Per run, every lane tests the obstacle disc
// positions it holds in registers (8 per lane = 512 per chunk, one global load each) against the run's inflated box into a bit mask and
// works its own set bits off, all lanes on the same ego frame, leaving at the first frame with a hit anywhere in the wavefront
// (round 1: one position per lane through all runs with nested divergent loops; rounds 2-3: survivors compacted into an LDS queue).
constexpr int NSEG = 8;
constexpr int QCAP = 8 * WAVE;      // (sizes the slack behind s_ego that used to hold the rounds 2-3 candidate queue; s_box lives there now)
__device__ __forceinline__ double grp8_min(double v) {
    v = fmin(v, dpp_mov<0xB1>(v, v)); v = fmin(v, dpp_mov<0x4E>(v, v)); v = fmin(v, dpp_mov<0x141>(v, v));
    return v;
}
__device__ __forceinline__ double grp8_max(double v) {
    v = fmax(v, dpp_mov<0xB1>(v, v)); v = fmax(v, dpp_mov<0x4E>(v, v)); v = fmax(v, dpp_mov<0x141>(v, v));
    return v;
}
// j / d and j % d for 0 <= j < 2^20, 1 <= d <= MPCX_PRED_STEPS_MAX through the single-precision reciprocal (+ one correction either way):
// the integer division the compiler emits costs ~25 instructions, and the candidate decode below runs it sixteen times per lane in a
// kernel that is bound by instruction issue
__device__ __forceinline__ void divmod_small(int j, int d, float inv_d, int &quo, int &rem) {
    int q = (int)(((float)j + 0.5f) * inv_d);
    int r = j - q * d;
    if (r < 0) { q -= 1; r += d; }
    if (r >= d) { q += 1; r -= d; }
    quo = q; rem = r;
}

__device__ int first_conflict(const mpcx_interaction_params &ip, double (*s_ego)[4], int na, const double *pred,
                              int ooff, int nobs, int oskip, const double *rem, const double *rcs, int n,
                              double (*s_box)[4], int lane, double &hx, double &hy,
                              bool boxes_ready = false,          // s_box already holds the runs' boxes (mpcx_interaction_params.plan_box)
                              const double *pdisc = nullptr      // disc centres of the poses of `rem` (mpcx_interaction_params.path_disc + 4 * row of rem[0]) or nullptr
#ifdef MPCX_INTER_PROFILE
                              , unsigned long long *fc_prof = nullptr
#endif
                              ) {
#ifdef MPCX_INTER_PROFILE
    unsigned long long fc_t = __builtin_amdgcn_s_memtime();
#define FCSTAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0 && fc_prof) fc_prof[k] += t_ - fc_t; fc_t = t_; } while (0)
    int dbg_queued = 0;      // dev build: candidates that reached a run's queue (returned in hx when there is no conflict)
#else
#define FCSTAMP(k) do {} while (0)
#endif
    const double md = 2.0 * ip.radius;
    const double md2lo = md * md * (1.0 - 1e-12), md2hi = md * md * (1.0 + 1e-12);
    const int steps = ip.pred_steps, w = ip.frame_window;
    const float inv_steps = 1.0f / (float)steps;
    const double slack = md * (1.0 + 1e-9) + 1e-9;      // conservative: never culls a pair within md

    // ---- boxes of NSEG runs of ego frames, inflated by slack > md so that no pair within md is ever skipped.
    // Lane (run = lane / 8, j = lane % 8) folds frames run*SL + j, + 8, ...; an 8-lane butterfly finishes the run.
    const int F = na > steps ? na : steps;
    const int SL = (F + NSEG - 1) / NSEG;                 // frames per run
    if (!boxes_ready) {
        const int sg = lane >> 3, j = lane & 7;
        const int fend = (sg + 1) * SL < F ? (sg + 1) * SL : F;
        double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
        for (int f = sg * SL + j; f < fend; f += 8) {
            const int fe = f < na ? f : na - 1;
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const double ex = s_ego[fe][2 * d], ey = s_ego[fe][2 * d + 1];
                x0 = fmin(x0, ex); x1 = fmax(x1, ex); y0 = fmin(y0, ey); y1 = fmax(y1, ey);
            }
        }
        x0 = grp8_min(x0); x1 = grp8_max(x1); y0 = grp8_min(y0); y1 = grp8_max(y1);
        if (j == 0) { s_box[sg][0] = x0 - slack; s_box[sg][1] = x1 + slack; s_box[sg][2] = y0 - slack; s_box[sg][3] = y1 + slack; }
    }
    __syncthreads();
    FCSTAMP(8);      // run boxes
    const long long NOKEY = 0x7fffffffffffffffLL;
    long long best = NOKEY;
    long long lbest = NOKEY;                               // this lane's own smallest key and the obstacle disc position it belongs to
    double lpx = 0.0, lpy = 0.0;
    int sg_limit = NSEG;                                   // runs >= sg_limit cannot hold the first row any more
    const int ncand_all = nobs * steps * 2;
    for (int cb = 0; cb < ncand_all; cb += 8 * WAVE) {
        double ox[8], oy[8];
        // all eight loads of the lane in flight at once, from addresses that are valid for every lane (clamped); out-of-range
        // candidates become +inf afterwards (a load under a lane predicate is its own exec-masked block with a full wait)
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int cidx = cb + u * WAVE + lane;
            const int cc = cidx < ncand_all ? cidx : ncand_all - 1;
            const int co = cc & 1;
            int g, o;                                                              // o: local obstacle rank
            divmod_small(cc >> 1, steps, inv_steps, o, g);
            int pool = ooff + o;
            if (oskip >= 0 && pool >= oskip) pool += 1;                                    // skip self
            const double *qq = pred + ((size_t)pool * steps + g) * 4 + 2 * co;
            ox[u] = qq[0]; oy[u] = qq[1];
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (cb + u * WAVE + lane >= ncand_all) { ox[u] = INFINITY; oy[u] = INFINITY; }       // fails every box test
        FCSTAMP(9);      // candidate loads
        for (int sg = 0; sg < sg_limit; sg++) {             // wave-uniform
            const double b0 = s_box[sg][0], b1 = s_box[sg][1], b2 = s_box[sg][2], b3 = s_box[sg][3];
            // Round 4: no queue.  Which of the lane's eight positions lie in the run's box is a bit mask, and every lane works its OWN set bits
            // off, one per pass (consecutive candidates -- the frames and discs of one obstacle -- sit on consecutive lanes, so the positions
            // inside a box are spread over the lanes: one or two passes).  The compaction into an LDS queue cost a ballot, two pop counts
            // and a store per position and run, two barriers per run, and a RELOAD of every queued position (one more memory round trip per
            // run visited, in a kernel whose time is a chain of such round trips).
            unsigned mask = 0;
#pragma unroll
            for (int u = 0; u < 8; u++)
                mask |= (unsigned)((ox[u] >= b0) & (ox[u] <= b1) & (oy[u] >= b2) & (oy[u] <= b3)) << u;
            FCSTAMP(10);     // box tests of one run
#ifdef MPCX_INTER_PROFILE
            dbg_queued += (int)wave_sum((double)__popc(mask));
#endif
            if (!__ballot(mask != 0u)) continue;
            const int f0 = sg * SL;
            int f1 = (sg + 1) * SL < F ? (sg + 1) * SL : F;
            bool found = false;
            while (__ballot(mask != 0u)) {                  // wave-uniform: as many passes as the busiest lane has positions in the box
                const bool on = mask != 0u;
                const int u = on ? __ffs((int)mask) - 1 : 0;
                mask &= mask - 1;
                double px = ox[0], py = oy[0];
#pragma unroll
                for (int q = 1; q < 8; q++) { px = (u == q) ? ox[q] : px; py = (u == q) ? oy[q] : py; }
                const int cidx = cb + u * WAVE + lane;
                const int co = cidx & 1;
                int g, o;
                divmod_small(cidx >> 1, steps, inv_steps, o, g);
                // frame-major, all lanes on the same frame: the key is ordered by the frame first, so once ANY lane has a hit at frame f no
                // later frame can hold the first row -- this pass ends with frame f, and later passes need not look beyond it
                for (int f = f0; f < f1; f++) {             // wave-uniform bounds
                    const int ff = f < steps ? f : steps - 1;
                    const int fe = f < na ? f : na - 1;
                    bool hit = false;
                    if (on && abs(g - ff) <= w) {            // else: no offset d in [-w, w] maps padded frame ff onto obstacle frame g
#pragma unroll
                        for (int ca = 0; ca < 2; ca++) {
                            if (within(s_ego[fe][2 * ca], s_ego[fe][2 * ca + 1], px, py, md, md2lo, md2hi)) {
                                // key = reference row order (frame, agent disc, obstacle, offset, obstacle disc); offsets ascend =>
                                // obstacle frames descend; the first offset reaching g is the one that counts
                                const long long key = ((((long long)f * 2 + ca) * MPCX_MAX_OBS + o) * MPCX_PRED_STEPS_MAX + (steps - 1 - g)) * 2 + co;
                                if (key < lbest) { lbest = key; lpx = px; lpy = py; }
                                hit = true;
                            }
                        }
                    }
                    if (__ballot(hit)) { found = true; f1 = f + 1; break; }
                }
            }
            FCSTAMP(11);     // distance tests of one run
            if (__ballot(found)) {                         // rows of later runs come later in the reference's order
                sg_limit = sg + 1;                         // later chunks: only runs up to this one can still win
                break;
            }
        }
    }
    best = wave_min_ll(lbest);
#ifdef MPCX_INTER_PROFILE
    if (best == NOKEY) hx = (double)dbg_queued;
#endif
    if (best == NOKEY) return -1;
    // the obstacle disc of the first row: the lane that found the key still holds its position (a key belongs to one candidate, a
    // candidate to one lane) -- no decode, no load
    const int owner = (int)__ffsll((long long)__ballot(lbest == best)) - 1;
    const double ox = rdlane(lpx, owner), oy = rdlane(lpy, owner);

    // ---- collision_avoidance.py:88-104: earliest pose of the detailed path (front-disc block, then rear-disc block)
    // The answer is the smallest index of the front-disc block if that block has a hit at all, else the smallest of the
    // rear-disc block: each block is walked in index order, 64 poses at a time, and left at the first batch with a hit.
    int first = 0x7fffffff;
    constexpr int PD = 4;      // batches of 64 poses in flight: the scan leaves at its first hit, and one batch per memory round trip made it a chain of 2-5
    for (int d = 0; d < 2 && first == 0x7fffffff; d++) {
        const double cx = ip.circle_centers[2 * d], cy = ip.circle_centers[2 * d + 1];
        for (int i0 = 0; i0 < n && first == 0x7fffffff; i0 += PD * WAVE) {
            double px[PD], py[PD], pc[PD], ps[PD];
#pragma unroll
            for (int k = 0; k < PD; k++) {
                const int i = i0 + k * WAVE + lane;
                const int ic = i < n ? i : n - 1;             // clamped address, masked below
                if (pdisc) { px[k] = pdisc[4 * (size_t)ic + 2 * d]; py[k] = pdisc[4 * (size_t)ic + 2 * d + 1]; pc[k] = 0.0; ps[k] = 0.0; }
                else { px[k] = rem[3 * ic]; py[k] = rem[3 * ic + 1]; pc[k] = rcs[2 * ic]; ps[k] = rcs[2 * ic + 1]; }
            }
#pragma unroll
            for (int k = 0; k < PD; k++) {
                const int i = i0 + k * WAVE + lane;
                // (with the host's table the disc centre is read, not rebuilt: px, py hold it)
                const double ex = pdisc ? px[k] : __dadd_rn(__dadd_rn(__dmul_rn(pc[k], cx), -__dmul_rn(ps[k], cy)), px[k]);
                const double ey = pdisc ? py[k] : __dadd_rn(__dadd_rn(__dmul_rn(ps[k], cx), __dmul_rn(pc[k], cy)), py[k]);
                const bool hit = i < n && within(ox, oy, ex, ey, md, md2lo, md2hi);
                const unsigned long long m = __ballot(hit);
                if (m && first == 0x7fffffff) first = d * n + i0 + k * WAVE + (int)__ffsll((long long)m) - 1;      // wave-uniform
            }
        }
    }
    FCSTAMP(12);     // earliest pose
    first = (first == 0x7fffffff) ? 0 : first % n;      // argmax of an all-False mask is 0
    hx = rem[3 * first]; hy = rem[3 * first + 1];
    return first;
}

constexpr int MAXF_STATIC = MPCX_EGO_FRAMES_MAX;      // moving_collision_kernel (explicit trajectories)

#ifdef MPCX_INTER_PROFILE
#define ISTAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) ((unsigned long long *)(a.hit_xy + 2 * (size_t)a.P))[16 * (size_t)p + (k)] = t_ - t_last; t_last = t_; } while (0)   /* dev build: needs 16 slots per ego behind hit_xy */
#else
#define ISTAMP(k) do {} while (0)
#endif
__global__ __launch_bounds__(64, 5) void interaction_kernel(InterArgs a) {
    // dynamic LDS, sized by the host from the longest path of the call (mpcx_interaction_params.max_path_len):
    //   s_cum [max_rem] doubles   step / cumulative lengths of the remaining path; once the resampling has consumed them the
    //                             same bytes hold s_ego [fcap][4] (ego disc centres per kept pose) and s_box (the runs' boxes)
    //   s_keep [fcap] shorts      indices of the kept poses
    // (round 4: no static LDS, 16-bit indices, no candidate queue: 6464 B at the benchmark's capacity.  Six wavefronts per SIMD -- launch bound 6: 80 VGPRs, 64 B/lane of
    // scratch -- measured 0.138 ms against 0.122 at five: the kernel is bound by instruction issue, more wavefronts only share it)
    extern __shared__ double s_dyn[];
    const int MAXREM = a.max_rem, MAXF = a.fcap;
    double *s_cum = s_dyn;
    unsigned short *s_keep = reinterpret_cast<unsigned short *>(s_dyn + MAXREM);             // (indices < max_rem <= 4096)
    double (*s_ego)[4] = reinterpret_cast<double (*)[4]>(s_cum);
    double (*s_box)[4] = reinterpret_cast<double (*)[4]>(s_cum + (size_t)MAXF * 4);           // bounding boxes of the ego discs per run of frames (256 B behind s_ego)

    const int p = blockIdx.x, lane = threadIdx.x;
#ifdef MPCX_INTER_PROFILE
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
    const mpcx_interaction_params &ip = a.ip;
    const double *path = a.path + 3 * (size_t)a.path_off[p];
    const double *pcs = a.path_cs + 2 * (size_t)a.path_off[p];
    const int len = a.path_len[p];
    const double x = a.state[4 * p], y = a.state[4 * p + 1], v = a.state[4 * p + 2];
    // ---- mpc_intersection.py:103-105: advance traj_agent_idx unless the previous tmp_trajectory collapsed onto it
    int tidx = a.traj_idx[p];
    bool advance = true;
    const int pcut = a.prev_cut ? a.prev_cut[p] : 0;
    if (a.prev_save && lane == 0) a.prev_save[p] = pcut;
    if (a.near && lane == 0) { a.near[3 * p] = tidx; a.near[3 * p + 1] = -1; a.near[3 * p + 2] = -1; }       // until the scan below has an answer
    // lane 0, next to every store of cut_len: the agent's place in the QP work queue of this step
    auto file_key = [&](int cl) {
        if (a.bin_cnt) {
            const int k = order_key_of(a.bin_hint ? a.bin_hint[p] : 0, cl != pcut);
            a.keyslot[p] = (k << 24) | atomicAdd(&a.bin_cnt[(p % MPCX_ORDER_COPIES) * MPCX_ORDER_BINS + k], 1);
        }
    };
    const int t_old = tidx;
    const int n_old = len - t_old;
    // The first batches of the distance pass are requested BEFORE it is known whether the ego advances at all (their addresses need only the
    // path and the old index): they travel together with the six values of the test below instead of one memory round trip later.  The
    // test itself loads all six values and compares them without short-circuit branches -- (a != b) || (c != d) || ... is a chain of up
    // to three dependent round trips for exactly the egos that stand still.
    constexpr int DEPTH = 4;
    double bx[DEPTH], by[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; k++) {
        const int i = k * WAVE + lane;
        const double *q = path + 3 * (size_t)(t_old + ((i < n_old && n_old <= MAXREM) ? i : 0));       // clamped address, selected afterwards
        const double vx = q[0], vy = q[1];
        bx[k] = i < n_old ? vx : 0.0; by[k] = i < n_old ? vy : 0.0;
    }
    {
        const int last = pcut > 0 ? pcut - 1 : tidx;
        const double ax = path[3 * tidx], ay = path[3 * tidx + 1], ath = path[3 * tidx + 2];
        const double lx = path[3 * last], ly = path[3 * last + 1], lth = path[3 * last + 2];
        advance = (pcut <= 0) | (ax != lx) | (ay != ly) | (ath != lth);
    }
    const int nobs = a.obs_cnt[p] - ((a.obs_skip && a.obs_skip[p] >= 0) ? 1 : 0);
    if (n_old > MAXREM || nobs > MPCX_MAX_OBS) {
        if (lane == 0) { a.hit_idx[p] = -2; a.cut_len[p] = len; file_key(len); a.hit_xy[2 * p] = 0; a.hit_xy[2 * p + 1] = 0; }
        return;
    }
    // ONE pass over the remaining path: distances to the ego (per-lane three smallest, ties by lower index) for
    // trajectories.py:100-126, and the step lengths |p_i - p_{i-1}| for resample_curve (trajectories.py:72-75).
    // Each point is loaded once (the predecessor comes from the neighbour lane), the next 64 points are in flight while
    // the current ones are worked on, and the ego distance takes its square root only where the squared distance could
    // enter the lane's three smallest (sqrt is monotone, so a larger square cannot give a smaller distance).
    // With the caller's arc-length table (mpcx_interaction_params.path_cum) the step lengths are not needed here at all -- no
    // square root, no neighbour shuffles, nothing stored -- and an agent that does not advance skips the pass.
    const double *cumtab = ip.path_cum ? ip.path_cum + (size_t)a.path_off[p] : nullptr;
    const bool tab = cumtab != nullptr;
    double b0d = INFINITY, b1d = INFINITY, b2d = INFINITY, b0s = INFINITY, b1s = INFINITY, b2s = INFINITY;
    int b0i = 0x7fffffff, b1i = 0x7fffffff, b2i = 0x7fffffff;
    if (!tab || advance) {
        // the points arrive in batches of DEPTH x 64: the loads of the next batch are all in flight while this one is worked on (one
        // batch deep the pass waited for an L2 round trip per 64 points: it is bound by its loads, not by its arithmetic)
        double lastx = 0.0, lasty = 0.0;
        for (int i0 = 0; i0 < n_old; i0 += DEPTH * WAVE) {
            double nbx[DEPTH], nby[DEPTH];
#pragma unroll
            for (int k = 0; k < DEPTH; k++) {
                const int i = i0 + (DEPTH + k) * WAVE + lane;
                const double *q = path + 3 * (size_t)(t_old + (i < n_old ? i : 0));
                const double vx = q[0], vy = q[1];
                nbx[k] = i < n_old ? vx : 0.0; nby[k] = i < n_old ? vy : 0.0;
            }
#pragma unroll
            for (int k = 0; k < DEPTH; k++) {
                const int i = i0 + k * WAVE + lane;
                const double px = bx[k], py = by[k];
                if (!tab) {
                    double qx = __shfl_up(px, 1, WAVE), qy = __shfl_up(py, 1, WAVE);
                    if (lane == 0) { qx = lastx; qy = lasty; }
                    lastx = rdlane(px, WAVE - 1); lasty = rdlane(py, WAVE - 1);
                    if (i < n_old) s_cum[i] = (i == 0) ? 0.0 : dist2d(px, py, qx, qy);
                }
                if (i < n_old) {
                    if (advance) {
                        const double dx = __dadd_rn(px, -x), dy = __dadd_rn(py, -y);
                        const double d2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
                        if (d2 < b2s || b2i == 0x7fffffff) {
                            const double d = __dsqrt_rn(d2);
                            if (d < b2d || (d == b2d && i < b2i)) {
                                if (d < b1d || (d == b1d && i < b1i)) {
                                    b2d = b1d; b2i = b1i; b2s = b1s;
                                    if (d < b0d || (d == b0d && i < b0i)) { b1d = b0d; b1i = b0i; b1s = b0s; b0d = d; b0i = i; b0s = d2; }
                                    else { b1d = d; b1i = i; b1s = d2; }
                                } else { b2d = d; b2i = i; b2s = d2; }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < DEPTH; k++) { bx[k] = nbx[k]; by[k] = nby[k]; }
        }
    }
    ISTAMP(0);      // distance / step-length pass
    if (advance) {
        if (n_old <= 1) tidx = t_old;
        else if (n_old == 2) tidx = t_old + 1;
        else {
            int bi[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {     // pop the wave-wide minimum three times
                double d = b0d; int ix = b0i;
                wave_argmin(d, ix);
                bi[r] = ix;
                if (b0i == ix) { b0d = b1d; b0i = b1i; b1d = b2d; b1i = b2i; b2d = INFINITY; b2i = 0x7fffffff; }
            }
            if (abs(bi[1] - bi[2]) == 2) tidx = bi[0] + t_old;
            else if (abs(bi[0] - bi[1]) == 1) tidx = max(bi[0], bi[1]) + t_old;
            else tidx = -1;
            if (a.near && lane == 0 && tidx >= 0) { a.near[3 * p + 1] = t_old + max(bi[0], max(bi[1], bi[2])); a.near[3 * p + 2] = t_old + min(bi[0], min(bi[1], bi[2])); }
        }
    }
    if (tidx < 0) {
        if (lane == 0) { a.hit_idx[p] = -3; a.cut_len[p] = len; file_key(len); a.hit_xy[2 * p] = 0; a.hit_xy[2 * p + 1] = 0; }
        return;
    }
    if (lane == 0) a.traj_idx[p] = tidx;
    const double *rem = path + 3 * (size_t)tidx;      // trajectory = trajectory_full[traj_agent_idx:]
    const double *rcs = pcs + 2 * (size_t)tidx;
    const int n = len - tidx;
    const int shift = tidx - t_old;                   // s_cum[shift + i] = step length into point i of the new trajectory
    if (nobs <= 0) {    // collision_avoidance.py:69-70
        if (lane == 0) { a.hit_idx[p] = -1; a.cut_len[p] = len; file_key(len); a.hit_xy[2 * p] = 0; a.hit_xy[2 * p + 1] = 0; }
        return;
    }
    __syncthreads();
    ISTAMP(1);      // three-smallest selection
    // ---- mpc_intersection.py:110-116 + trajectories.py:72-86: ego prediction = resample_curve(trajectory, dl_k).
    // np.cumsum adds strictly left to right.  Replaying that on one lane cost a third of this kernel, so the cumulative
    // lengths first come from a PARALLEL scan (all terms >= 0: it differs from the sequential sum by <= 2.4e-11 for 1024
    // terms summing to <= 200 m) and the bucket floor(c_i / dl_i) of every point is accepted only when c_i / dl_i is farther
    // from an integer than that error can move it (margin 1e-10 / dl_i).  If a single point of this ego is too close to call,
    // the ego is redone with the sequential sum -- same outputs as before in every case, about 1e-6 of the egos take that path.
    const bool accel_phase = v < ip.max_speed;
    const double dl_const = __dmul_rn(ip.dt, ip.max_speed);
    auto prefix_fast = [&]() {
        double carry = 0.0;
        for (int i0 = 0; i0 < n; i0 += WAVE) {
            const int i = i0 + lane;
            double t = (i >= 1 && i < n) ? s_cum[shift + i] : 0.0;     // the first point of the new trajectory has no predecessor
            t += dpp_mov<0x111>(0.0, t);
            t += dpp_mov<0x112>(0.0, t);
            t += dpp_mov<0x114>(0.0, t);
            t += dpp_mov<0x118>(0.0, t);
            t += dpp_mov<0x142, 0xA>(0.0, t);       // row_bcast:15 -> rows 1, 3
            t += dpp_mov<0x143, 0xC>(0.0, t);       // row_bcast:31 -> rows 2, 3
            t += carry;
            carry = rdlane(t, WAVE - 1);
            if (i < n) s_cum[shift + i] = t;
        }
    };
    auto prefix_exact = [&]() {
        for (int i = lane; i < n; i += WAVE) {        // the step lengths again (prefix_fast overwrote them)
            const double *q = rem + 3 * (size_t)i;
            s_cum[shift + i] = (i == 0) ? 0.0 : dist2d(q[0], q[1], q[-3], q[-2]);
        }
        __syncthreads();
        if (lane == 0) {                      // np.cumsum: strictly sequential adds (one lane; loads batched 16 at a time)
            double c = 0.0;
            int i = 1;
            for (; i + 16 <= n; i += 16) {
                double t[16];
#pragma unroll
                for (int q = 0; q < 16; q++) t[q] = s_cum[shift + i + q];
#pragma unroll
                for (int q = 0; q < 16; q++) { c = __dadd_rn(c, t[q]); t[q] = c; }
#pragma unroll
                for (int q = 0; q < 16; q++) s_cum[shift + i + q] = t[q];
            }
            for (; i < n; i++) { c = __dadd_rn(c, s_cum[shift + i]); s_cum[shift + i] = c; }
        }
        __syncthreads();
    };
    // bucket of every point, keep the points where the bucket advances (+ first and last); returns the number kept
    const double cum0 = tab ? cumtab[tidx] : 0.0;
    const double inv_const = frcp(dl_const);
    const double marg = tab ? ip.path_cum_err + 1e-13 : 1.01e-10;     // how far the fast running sum can be from np.cumsum's
    auto resample = [&](auto check_tag, bool &unsure) -> int {
        constexpr bool check = decltype(check_tag)::value;      // two instantiations: the fast pass carries no division and no 64-bit integers
        // check = true: the fast pass -- running sums from the table (or the parallel scan), quotient by reciprocal (<= 2 ulp from the
        // division), bucket accepted only outside the margin; check = false: np.cumsum's own sums (prefix_exact) and the division
        int base = 0;
        long long q_carry = 0;                // bucket of the last element of the previous 64-block
        double qd_carry = 0.0;                // (fast pass: the buckets as doubles -- floor() of a quotient below 2^52 is an exact integer)
        unsure = false;
        for (int i0 = 0; i0 < n; i0 += WAVE) {
            const int i = i0 + lane;
            long long q = 0;
            double qd = 0.0;
            if (i < n) {
                double dl = dl_const, inv = inv_const;
                // the predicted speed v + a (i + 1) is monotone in i: once the FIRST point of a 64-point block has reached max_speed every
                // later one has, and dl is the constant (a car needs four points for that: only the first block takes this branch)
                if (accel_phase && (!(ip.max_accel >= 0.0) || __dadd_rn(__dmul_rn(ip.max_accel, (double)(i0 + 1)), v) < ip.max_speed)) {
                    const double r = __dadd_rn(__dmul_rn(ip.max_accel, (double)(i + 1)), v);   // cumsum of equal terms (exact for 2.0) + v
                    dl = __dmul_rn(ip.dt, fmin(r, ip.max_speed));
                    if (check) inv = frcp(dl);
                }
                const double c = s_cum[shift + i];
                if constexpr (check) {
                    const double r = c * inv;
                    qd = floor(r);
                    if (c != 0.0) {               // c == 0 is exact in every summation order
                        const double room = fabs(r - rint(r));
                        if (!(dl > 0.0) || !(r < 4e15) || !(room > marg * inv + 2e-15 * fabs(r))) unsure = true;
                    }
                } else
                    q = (long long)floor(__ddiv_rn(c, dl));
            }
            bool adv;
            if constexpr (check) {
                const double qprev = lane_prev(qd, qd_carry);       // wave_shr:1, lane 0 takes the previous block's last bucket
                qd_carry = rdlane(qd, WAVE - 1);
                adv = qd - qprev >= 1.0;
            } else {
                long long qprev = __shfl_up(q, 1, WAVE);
                if (lane == 0) qprev = q_carry;
                q_carry = __shfl(q, WAVE - 1, WAVE);
                adv = q - qprev >= 1;
            }
            const bool keep = (i < n) && ((i == 0) || (i == n - 1) || adv);
            const unsigned long long m = __ballot(keep);
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (keep && pos < MAXF) s_keep[pos] = i;
            base += __popcll(m);
        }
        return base;
    };
    // Round 4: the kept poses by SEARCH instead of a scan.  Beyond the first 64 points dl is a constant (the predicted speed has saturated), so
    // the buckets floor(c_i / dl) never decrease along the path and the kept points -- those whose bucket exceeds their predecessor's -- are
    // the FIRST point of every bucket value that occurs: one binary search in the arc-length table per bucket boundary (~36 of them, one per
    // lane) instead of a pass over all ~700 points (which was a third of this kernel's instructions), and the table no longer goes through
    // LDS.  The margin test that guards the fast sums is only needed where it can fail: at the two points around every boundary and at
    // the last point (a quotient close to an integer m elsewhere would put a boundary point at least as close to m).  First 64 points: as
    // before, point by point (dl varies there while the ego accelerates).  Same kept set as the scan, bit for bit (tests + the
    // MPCX_INTER_FORCE_EXACT build, which still takes the sequential path).
    const bool search_ok = tab && n > WAVE &&
                           (!accel_phase || (ip.max_accel >= 0.0 && !(__dadd_rn(__dmul_rn(ip.max_accel, (double)(WAVE + 1)), v) < ip.max_speed)));
    auto resample_search = [&](bool &unsure) -> int {
        const double *ct = cumtab + tidx;
        auto risky = [&](double c, double r, double dl_, double inv_) -> bool {
            return c != 0.0 && (!(dl_ > 0.0) || !(r < 4e15) || !(fabs(r - rint(r)) > marg * inv_ + 2e-15 * fabs(r)));
        };
        unsure = false;
        // ---- points 0..63 (n > 64: none of them is the last point)
        double dl = dl_const, inv = inv_const;
        if (accel_phase && (!(ip.max_accel >= 0.0) || __dadd_rn(__dmul_rn(ip.max_accel, 1.0), v) < ip.max_speed)) {
            const double r = __dadd_rn(__dmul_rn(ip.max_accel, (double)(lane + 1)), v);
            dl = __dmul_rn(ip.dt, fmin(r, ip.max_speed));
            inv = frcp(dl);
        }
        const double c0 = ct[lane] - cum0;
        const double r0 = c0 * inv;
        const double qd = floor(r0);
        if (risky(c0, r0, dl, inv)) unsure = true;
        const double qprev = lane_prev(qd, 0.0);
        const bool keep0 = (lane == 0) || (qd - qprev >= 1.0);
        const unsigned long long m0 = __ballot(keep0);
        {
            const int pos = __popcll(m0 & ((1ull << lane) - 1ull));
            if (keep0 && pos < MAXF) s_keep[pos] = lane;
        }
        int base = __popcll(m0);
        const double Q63 = rdlane(qd, WAVE - 1);
        // ---- point 64 (the first with the constant dl) and the last point
        const double c64 = ct[WAVE] - cum0, cl = ct[n - 1] - cum0;
        const double r64 = c64 * inv_const, rl = cl * inv_const;
        const double Q64 = floor(r64), Ql = floor(rl);
        if (risky(c64, r64, dl_const, inv_const) || risky(cl, rl, dl_const, inv_const)) unsure = true;
        if (__ballot(unsure)) return base;                    // (wave-uniform) the caller redoes the ego with the sequential sums
        if ((Q64 - Q63 >= 1.0) || n - 1 == WAVE) {
            if (lane == 0 && base < MAXF) s_keep[base] = WAVE;
            base++;
        }
        // ---- one target bucket value per lane: idx(b) = first i in [65, n) with floor(c_i / dl) >= b, b = Q64 + 1 .. Ql
        const int ntar = (int)(Ql - Q64);                     // 0 <= ntar: the buckets do not decrease; < 4e15 checked above
        int last_idx = WAVE;                                  // the largest index handled so far
        if (ntar > 0) {
            int span = n - 1 - (WAVE + 1), iters = 0;         // search range [65, n - 1]: r_{n-1} >= Ql >= b, so the answer exists
            while (span > 0) { iters++; span >>= 1; }
            // planner paths are sampled at (nearly) equal arc-length steps, so the boundary is where a straight line through c_64 and
            // c_{n-1} puts it: ONE probe of the two points around the guess -- they are the two points the margin test needs anyway --
            // instead of a chain of ~10 dependent loads; the binary search remains for a path on which the guess misses
            const double hstep = (cl - c64) / (double)(n - 1 - WAVE);
            const double inv_h = hstep > 0.0 ? 1.0 / hstep : 0.0;
            for (int t0 = 0; t0 < ntar; t0 += WAVE) {
                const int t = t0 + lane;
                const bool valid = t < ntar;
                const double b = valid ? Q64 + 1.0 + (double)t : Ql;
                double gd = ceil((b * dl_const - c64) * inv_h) + (double)WAVE;
                gd = fmin(fmax(gd, (double)(WAVE + 1)), (double)(n - 1));
                int idx = (int)gd;
                double cj = ct[idx] - cum0, ci = ct[idx - 1] - cum0;
                const bool miss = valid && !((ci * inv_const < b) && (cj * inv_const >= b));
                if (__ballot(miss)) {                         // wave-uniform
                    int lo = miss ? WAVE + 1 : idx, hi = miss ? n - 1 : idx;
                    for (int it = 0; it < iters; it++) {      // uniform trip count; a lane that has converged repeats its last probe
                        const int mid = lo < hi ? (lo + hi) >> 1 : lo;
                        const double cm = ct[mid] - cum0;
                        const bool ge = cm * inv_const >= b;
                        if (lo < hi) { if (ge) hi = mid; else lo = mid + 1; }
                    }
                    idx = lo;
                    cj = ct[idx] - cum0; ci = ct[idx - 1] - cum0;
                }
                if (valid && (risky(cj, cj * inv_const, dl_const, inv_const) || risky(ci, ci * inv_const, dl_const, inv_const))) unsure = true;
                int prev = __shfl_up(idx, 1, WAVE);
                if (lane == 0) prev = last_idx;
                const bool isnew = valid && idx != prev;
                const unsigned long long mk = __ballot(isnew);
                const int pos = base + __popcll(mk & ((1ull << lane) - 1ull));
                if (isnew && pos < MAXF) s_keep[pos] = idx;
                base += __popcll(mk);
                const int nv = ntar - t0 < WAVE ? ntar - t0 : WAVE;
                last_idx = __shfl(idx, nv - 1, WAVE);
            }
        }
        if (n - 1 > WAVE && last_idx != n - 1) {              // keep_last_point
            if (lane == 0 && base < MAXF) s_keep[base] = n - 1;
            base++;
        }
        return base;
    };
    bool unsure = false;
    int na = 0;
    // Round 4: the ego prediction from the host's table (mpcx_interaction_params.plan_*).  Once the predicted speed has saturated dl is the
    // constant DT * MAX_SPEED; if the points before that (four from standstill with the stock constants) all stay in bucket 0 with their
    // own, smaller dl -- then they do with the constant one too -- the bucket sequence of trajectory_full[tidx:] is the one the host
    // resampled with the constant dl, i.e. the kept poses depend on tidx alone: their number, their disc centres and the run boxes are ONE
    // read of row tidx (one memory round trip) instead of the resampling pass, the disc arithmetic and the box reductions.
    bool plan_ok = false;
    const size_t prow = (size_t)a.path_off[p] + tidx;
#ifndef MPCX_INTER_FORCE_EXACT
    if (ip.plan_cnt && tab && ip.plan_dl == dl_const && ip.plan_steps == ip.pred_steps && ip.plan_radius == ip.radius && ip.plan_cap <= MAXF && ip.plan_cap <= WAVE) {
        const int cnt = ip.plan_cnt[prow];
        // the row's disc centres and boxes are requested together with its count (the table has plan_cap poses per row whatever the count is):
        // one memory round trip, not two
        const double *pd = ip.plan_disc + prow * (size_t)ip.plan_cap * 4;
        const int f0 = lane < 2 * ip.plan_cap ? lane : 0, f1 = lane + WAVE < 2 * ip.plan_cap ? lane + WAVE : 0;
        const double e0x = pd[2 * f0], e0y = pd[2 * f0 + 1], e1x = pd[2 * f1], e1y = pd[2 * f1 + 1];
        const double bxv = ip.plan_box[prow * (4 * NSEG) + (lane < 4 * NSEG ? lane : 0)];
        // lane i: has the predicted speed of point i reached MAX_SPEED?  (monotone in i for max_accel >= 0)
        const bool sat = !accel_phase || !(__dadd_rn(__dmul_rn(ip.max_accel, (double)(lane + 1)), v) < ip.max_speed);
        const unsigned long long sm = __ballot(sat);
        const int isat = sm ? (int)__ffsll((long long)sm) - 1 : WAVE;
        bool bad = false;
        if (lane < isat && lane < n) {       // points with their own dl: bucket 0 needs c_i < dl_i, with the table's error bound on the safe side
            const double r = __dadd_rn(__dmul_rn(ip.max_accel, (double)(lane + 1)), v);
            const double dli = __dmul_rn(ip.dt, r);
            const double ci = cumtab[tidx + lane] - cum0;
            bad = !(ci + marg < dli * (1.0 - 1e-12)) || !(dli > 0.0);
        }
        plan_ok = cnt > 0 && cnt <= ip.plan_cap && ip.max_accel >= 0.0 && isat < WAVE && !__ballot(bad);
        if (plan_ok) {
            na = cnt;
            // lane f handles disc f & 1 of pose f >> 1 (and f + 64 likewise; plan_cap <= 64 poses)
            if (lane < 2 * na) { s_ego[lane >> 1][2 * (lane & 1)] = e0x; s_ego[lane >> 1][2 * (lane & 1) + 1] = e0y; }
            if (lane + WAVE < 2 * na) { s_ego[(lane + WAVE) >> 1][2 * (lane & 1)] = e1x; s_ego[(lane + WAVE) >> 1][2 * (lane & 1) + 1] = e1y; }
            if (lane < 4 * NSEG) (&s_box[0][0])[lane] = bxv;
        }
    }
#endif
    if (plan_ok) {
        ISTAMP(2);
    } else if (search_ok) {
        ISTAMP(2);
        na = resample_search(unsure);
    } else {
        if (tab) {      // running sums from the table, four batches of loads in flight (one by one the pass waited a memory round trip per 64 points)
            for (int i0 = 0; i0 < n; i0 += 4 * WAVE) {
                double t[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { const int i = i0 + k * WAVE + lane; t[k] = cumtab[tidx + (i < n ? i : 0)]; }
#pragma unroll
                for (int k = 0; k < 4; k++) { const int i = i0 + k * WAVE + lane; if (i < n) s_cum[shift + i] = t[k] - cum0; }
            }
        } else prefix_fast();
        __syncthreads();
        ISTAMP(2);      // cumulative lengths
        na = resample(std::true_type{}, unsure);
    }
#ifdef MPCX_INTER_FORCE_EXACT
    unsure = true;                            // dev build: every ego takes the sequential path (tests run both builds)
#endif
    if (__ballot(unsure)) {
        __syncthreads();
        prefix_exact();
        na = resample(std::false_type{}, unsure);
    }
    if (na > MAXF) {
        if (lane == 0) { a.hit_idx[p] = -2; a.cut_len[p] = len; file_key(len); a.hit_xy[2 * p] = 0; a.hit_xy[2 * p + 1] = 0; }
        return;
    }
    __syncthreads();
    ISTAMP(3);      // resample
    // ego disc centres per kept pose
    if (!plan_ok)
    for (int f = lane; f < na; f += WAVE) {
        const int i = s_keep[f];
        const double px = rem[3 * i], py = rem[3 * i + 1], c = rcs[2 * i], s = rcs[2 * i + 1];
#pragma unroll
        for (int d = 0; d < 2; d++) {
            const double cx = ip.circle_centers[2 * d], cy = ip.circle_centers[2 * d + 1];
            s_ego[f][2 * d] = __dadd_rn(__dadd_rn(__dmul_rn(c, cx), -__dmul_rn(s, cy)), px);
            s_ego[f][2 * d + 1] = __dadd_rn(__dadd_rn(__dmul_rn(s, cx), __dmul_rn(c, cy)), py);
        }
    }
    __syncthreads();
    ISTAMP(4);      // ego discs
    const int ooff = a.obs_off[p], oskip = a.obs_skip ? a.obs_skip[p] : -1;
    double hx, hy;
#ifdef MPCX_INTER_PROFILE
    const double *pdisc = ip.plan_cnt ? ip.path_disc + 4 * prow : nullptr;     // disc centres of trajectory_full[tidx:] from the host's table
    const int first = first_conflict(ip, s_ego, na, a.pred, ooff, nobs, oskip, rem, rcs, n, s_box, lane, hx, hy, plan_ok, pdisc,
                                     (unsigned long long *)(a.hit_xy + 2 * (size_t)a.P) + 16 * (size_t)p);
#else
    const double *pdisc = ip.plan_cnt ? ip.path_disc + 4 * prow : nullptr;     // disc centres of trajectory_full[tidx:] from the host's table
    const int first = first_conflict(ip, s_ego, na, a.pred, ooff, nobs, oskip, rem, rcs, n, s_box, lane, hx, hy, plan_ok, pdisc);
#endif
    ISTAMP(5);      // conflict search (+ path scan on a hit)
    if (first < 0) {
#ifdef MPCX_INTER_PROFILE
        if (lane == 0) { a.hit_idx[p] = -1; a.cut_len[p] = len; file_key(len); a.hit_xy[2 * p] = hx; a.hit_xy[2 * p + 1] = 0; }
#else
        if (lane == 0) { a.hit_idx[p] = -1; a.cut_len[p] = len; file_key(len); a.hit_xy[2 * p] = 0; a.hit_xy[2 * p + 1] = 0; }
#endif
        return;
    }
    // ---- collision_avoidance.py:107-119 on trajectory_full, then mpc_intersection.py:130-134
    int cut = 0x7fffffff;
    if (ip.path_first_within) {
        // (hx, hy) IS path point tidx + first: the first point within 1 mm of it is a property of the path, tabulated by the host with
        // the reference's own expression (mpcx_interaction_params.path_first_within) -- no scan of the path up to the conflict
        cut = ip.path_first_within[(size_t)a.path_off[p] + tidx + first];
    } else {
        const double cr = 0.001, cr2lo = cr * cr * (1.0 - 1e-12), cr2hi = cr * cr * (1.0 + 1e-12);
        // (hx, hy) IS path point tidx + first, so the first index within 1 mm cannot lie beyond it: scan [0, tidx + first]
        const int jend = tidx + first + 1 < len ? tidx + first + 1 : len;
        for (int j = lane; j < jend; j += WAVE)         // same decision as sqrt(dx*dx + dy*dy) <= 0.001, no sqrt on the bulk
            if (within(path[3 * j], path[3 * j + 1], hx, hy, cr, cr2lo, cr2hi)) cut = j < cut ? j : cut;
        cut = wave_min_i(cut);
    }
    int cl = len;
    if (cut != 0x7fffffff) { cl = cut - ip.cutoff_margin; cl = cl > tidx + 1 ? cl : tidx + 1; }
    if (lane == 0) { a.hit_idx[p] = first; a.hit_xy[2 * p] = hx; a.hit_xy[2 * p + 1] = hy; a.cut_len[p] = cl; file_key(cl); }
    ISTAMP(6);      // cut index
}

// ------------------------------------------------------------------------------------------------------------
// check_collision_moving_cars on EXPLICIT trajectories (the reference's own signature, collision_avoidance.py:66):
// the caller has already resampled the ego and predicted the obstacles (mpc_intersection.py:110-122).
struct PoseDiscArgs {
    mpcx_interaction_params ip;
    int n;                       // number of poses
    const double *pose, *cs;     // [n][3] (x,y,yaw), [n][2] (cos,sin of yaw)
    double *out;                 // [n][4] disc centres
};
__global__ __launch_bounds__(256) void pose_disc_kernel(PoseDiscArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const double px = a.pose[3 * i], py = a.pose[3 * i + 1], c = a.cs[2 * i], s = a.cs[2 * i + 1];
#pragma unroll
    for (int d = 0; d < 2; d++) {
        const double cx = a.ip.circle_centers[2 * d], cy = a.ip.circle_centers[2 * d + 1];
        a.out[4 * i + 2 * d] = __dadd_rn(__dadd_rn(__dmul_rn(c, cx), -__dmul_rn(s, cy)), px);
        a.out[4 * i + 2 * d + 1] = __dadd_rn(__dadd_rn(__dmul_rn(s, cx), __dmul_rn(c, cy)), py);
    }
}

struct MovArgs {
    mpcx_interaction_params ip;
    int P;
    const double *ego, *ego_cs;
    const int32_t *ego_off, *ego_len;
    const double *path, *path_cs;
    const int32_t *path_off, *path_len;
    const double *pred;
    const int32_t *obs_off, *obs_cnt;
    int32_t *hit_idx;
    double *hit_xy;
};

__global__ __launch_bounds__(64) void moving_collision_kernel(MovArgs a) {
    constexpr int MAXF = MAXF_STATIC;
    __shared__ double s_ego[MAXF][4];
    __shared__ double s_box[NSEG][4];
    const int p = blockIdx.x, lane = threadIdx.x;
    const mpcx_interaction_params &ip = a.ip;
    const int na = a.ego_len[p], n = a.path_len[p], nobs = a.obs_cnt[p];
    if (nobs <= 0) { if (lane == 0) { a.hit_idx[p] = -1; a.hit_xy[2 * p] = 0; a.hit_xy[2 * p + 1] = 0; } return; }
    if (na > MAXF || na < 1 || n < 1 || nobs > MPCX_MAX_OBS) { if (lane == 0) { a.hit_idx[p] = -2; a.hit_xy[2 * p] = 0; a.hit_xy[2 * p + 1] = 0; } return; }
    const double *ego = a.ego + 3 * (size_t)a.ego_off[p], *ecs = a.ego_cs + 2 * (size_t)a.ego_off[p];
    for (int f = lane; f < na; f += WAVE) {
        const double px = ego[3 * f], py = ego[3 * f + 1], c = ecs[2 * f], s = ecs[2 * f + 1];
#pragma unroll
        for (int d = 0; d < 2; d++) {
            const double cx = ip.circle_centers[2 * d], cy = ip.circle_centers[2 * d + 1];
            s_ego[f][2 * d] = __dadd_rn(__dadd_rn(__dmul_rn(c, cx), -__dmul_rn(s, cy)), px);
            s_ego[f][2 * d + 1] = __dadd_rn(__dadd_rn(__dmul_rn(s, cx), __dmul_rn(c, cy)), py);
        }
    }
    __syncthreads();
    double hx, hy;
    const int first = first_conflict(ip, s_ego, na, a.pred, a.obs_off[p], nobs, -1,
                                     a.path + 3 * (size_t)a.path_off[p], a.path_cs + 2 * (size_t)a.path_off[p], n, s_box, lane, hx, hy);
    if (lane == 0) { a.hit_idx[p] = first; a.hit_xy[2 * p] = first < 0 ? 0.0 : hx; a.hit_xy[2 * p + 1] = first < 0 ? 0.0 : hy; }
}

}  // namespace mpcx


extern "C" int32_t mpcx_interaction_batch(mpcx_ctx *ctx, const mpcx_interaction_params *ip, int32_t P,
                                          const double *state, const double *path_xyyaw, const double *path_cs,
                                          const int32_t *path_off, const int32_t *path_len, const int32_t *prev_cut_len,
                                          int32_t n_obs_pool, const double *obs6, const int32_t *obs_off,
                                          const int32_t *obs_cnt, const int32_t *obs_skip,
                                          int32_t *traj_idx, int32_t *hit_idx, double *hit_xy, int32_t *cut_len) {
    if (!ctx) return MPCX_E_INVALID;
    if (P == 0) return MPCX_OK;
    if (!ip || P < 0 || n_obs_pool < 0 || !state || !path_xyyaw || !path_cs || !path_off || !path_len || !obs_off ||
        !obs_cnt || !traj_idx || !hit_idx || !hit_xy || !cut_len || (n_obs_pool > 0 && !obs6))
        return mpcx_fail(ctx, MPCX_E_INVALID, "interaction_batch: null pointer or negative size");
    if (ip->pred_steps < 1 || ip->pred_steps > MPCX_PRED_STEPS_MAX || ip->frame_window < 0 || !(ip->dt > 0) || !(ip->L > 0))
        return mpcx_fail(ctx, MPCX_E_INVALID, "interaction_batch: pred_steps outside 1..%d or bad dt/L/frame_window", MPCX_PRED_STEPS_MAX);
    if (P == 0) return MPCX_OK;
    { int32_t rc = mpcx_ensure_pred(ctx, (size_t)(n_obs_pool > 0 ? n_obs_pool : 1) * ip->pred_steps * 4); if (rc != MPCX_OK) return rc; }
    if (n_obs_pool > 0) {
        mpcx::PredArgs pa{*ip, n_obs_pool, obs6, ctx->pred, ctx->pack_state, ctx->pack_applied, ctx->pack_state ? const_cast<double *>(obs6) : nullptr};
        hipLaunchKernelGGL(mpcx::predict_kernel, dim3((n_obs_pool + 63) / 64), dim3(64), 0, ctx->stream, pa);
    }
    // capacity: max_path_len path points (0 = MPCX_MAX_REMAINING; never below 512), rounded up to whole wavefronts; the LDS that
    // holds their cumulative lengths later holds the ego discs of max_rem / 4 - 32 resampled poses and the runs' boxes.
    // The kernel hides its memory latency with resident wavefronts (17 -> 11 blocks per CU costs 36 %), so the LDS follows the
    // call's longest path instead of a fixed 1024 points.
    int max_rem = ip->max_path_len > 0 ? ip->max_path_len : MPCX_MAX_REMAINING;
    if (max_rem < 512) max_rem = 512;
    max_rem = (max_rem + 63) / 64 * 64;
    if (max_rem > MPCX_MAX_PATH_LEN)
        return mpcx_fail(ctx, MPCX_E_INVALID, "interaction_batch: max_path_len %d exceeds %d", ip->max_path_len, MPCX_MAX_PATH_LEN);
    const int fcap = max_rem / 4 - mpcx::QCAP * 2 / 32;
    const size_t lds = (size_t)max_rem * sizeof(double) + ((size_t)fcap * sizeof(unsigned short) + 7) / 8 * 8;
    mpcx::InterArgs ia{*ip, P, state, path_xyyaw, path_cs, path_off, path_len, prev_cut_len, ctx->pred,
                       obs_off, obs_cnt, obs_skip, traj_idx, hit_idx, hit_xy, cut_len, max_rem, fcap, ctx->inter_prev_save,
                       ctx->bin_hint, ctx->bin_hint ? ctx->bins : nullptr, ctx->bin_hint ? ctx->bins + MPCX_ORDER_COPIES * MPCX_ORDER_BINS : nullptr,
                       ctx->inter_near};
    hipLaunchKernelGGL(mpcx::interaction_kernel, dim3(P), dim3(64), lds, ctx->stream, ia);
    return mpcx_check_launch(ctx, "interaction kernels");
}

int32_t mpcx_ensure_pred(mpcx_ctx *ctx, size_t need) {
    if (need <= ctx->pred_cap) return MPCX_OK;
    if (ctx->pred) (void)hipFree(ctx->pred);
    ctx->pred = nullptr; ctx->pred_cap = 0;
    if (hipMalloc((void **)&ctx->pred, need * sizeof(double)) != hipSuccess)
        return mpcx_fail(ctx, MPCX_E_LAUNCH, "cannot allocate %zu bytes of prediction scratch", need * sizeof(double));
    ctx->pred_cap = need;
    return MPCX_OK;
}

extern "C" int32_t mpcx_moving_collision_batch(mpcx_ctx *ctx, const mpcx_interaction_params *ip, int32_t P,
                                               const double *ego_xyyaw, const double *ego_cs, const int32_t *ego_off,
                                               const int32_t *ego_len, const double *path_xyyaw, const double *path_cs,
                                               const int32_t *path_off, const int32_t *path_len,
                                               int32_t n_obs_pool, const double *obs_xyyaw, const double *obs_cs,
                                               const int32_t *obs_off, const int32_t *obs_cnt,
                                               int32_t *hit_idx, double *hit_xy) {
    if (!ctx) return MPCX_E_INVALID;
    if (P == 0) return MPCX_OK;
    if (!ip || P < 0 || n_obs_pool < 0 || !ego_xyyaw || !ego_cs || !ego_off || !ego_len || !path_xyyaw || !path_cs ||
        !path_off || !path_len || !obs_off || !obs_cnt || !hit_idx || !hit_xy || (n_obs_pool > 0 && (!obs_xyyaw || !obs_cs)))
        return mpcx_fail(ctx, MPCX_E_INVALID, "moving_collision_batch: null pointer or negative size");
    if (ip->pred_steps < 1 || ip->pred_steps > MPCX_PRED_STEPS_MAX || ip->frame_window < 0)
        return mpcx_fail(ctx, MPCX_E_INVALID, "moving_collision_batch: pred_steps outside 1..%d or negative frame_window", MPCX_PRED_STEPS_MAX);
    if (P == 0) return MPCX_OK;
    const size_t nposes = (size_t)n_obs_pool * ip->pred_steps;
    int32_t rc = mpcx_ensure_pred(ctx, (nposes ? nposes : 1) * 4);
    if (rc != MPCX_OK) return rc;
    if (nposes) {
        mpcx::PoseDiscArgs pd{*ip, (int)nposes, obs_xyyaw, obs_cs, ctx->pred};
        hipLaunchKernelGGL(mpcx::pose_disc_kernel, dim3((unsigned)((nposes + 255) / 256)), dim3(256), 0, ctx->stream, pd);
    }
    mpcx::MovArgs ma{*ip, P, ego_xyyaw, ego_cs, ego_off, ego_len, path_xyyaw, path_cs, path_off, path_len, ctx->pred,
                     obs_off, obs_cnt, hit_idx, hit_xy};
    hipLaunchKernelGGL(mpcx::moving_collision_kernel, dim3(P), dim3(64), 0, ctx->stream, ma);
    return mpcx_check_launch(ctx, "moving collision kernels");
}
