// mpcx_qp_stage.h -- stage-structured interior-point solver for the MPC QP of lib/mpc.py:138-208.
//
// Same Mehrotra predictor-corrector iteration as the condensed solver (same residuals, step rules, safeguards and
// exit conditions), but the Newton system  (H + G'DG) du = -(rd + G'w)  is never formed: because every cost and
// constraint of the problem is local to one stage (x_t, u_t, u_{t-1}), the step is the solution of a
// time-varying LQ problem and comes out of one backward Riccati sweep (6-state: dx, dy, dv, dpsi and the previous
// input pair, which carries the input-rate cost and the steering-rate constraint) plus one forward sweep;
// the two right-hand sides of the predictor-corrector reuse the stored gains.  O(T) work per iteration instead of
// O(T^3), and a working set small enough that A FEW LANES (eight on the GPU) hold one problem: lane q of a group owns SPL consecutive
// stages, sweeps run as LQ "turns" (lane q works while the others wait) with the 6x6 cost-to-go handed to the
// neighbour lane by DPP, and everything that is local to a stage (slacks, multipliers, residuals, step lengths)
// runs on all lanes at once.  A wavefront therefore carries 64/LQ problems.
//
// Cx::JERK selects the five-state problem of lib/mpc_jerk.py:143-208 instead: the extra state x4 integrates the acceleration
// input and feeds the speed (v' = v + dt (x4 + a), x4' = x4 + dt a, mpc_jerk.py:73-78), its initial value is a free unknown
// (only x[:4, 0] is pinned, line 193) and the jerk term of line 190 is w (x4' - x4)^2 = w dt^2 a^2 on the stages it covers.
// The sweeps then carry seven states (x4 last); everything local to a row is unchanged (no row involves x4).
//
// The code is written against a small policy class (group geometry, neighbour hand-off, group reductions, row
// storage) so that the SAME source runs on the host with one "lane" per problem; tests/ compiles that build with
// g++ to check the algebra against the dense CPU restatement and under the sanitizers.  It is not a product path.
#pragma once
#include <math.h>
#include <stdint.h>
#include "mpcx.h"

#if defined(__HIPCC__)
#define MPCX_HD __host__ __device__ __forceinline__
#else
#define MPCX_HD inline
#endif
#ifndef MPCX_STEP_FRACTION
#define MPCX_STEP_FRACTION 0.999     /* fraction of the step to the boundary; see mpcx_common.h (the host build of this header has no other source) */
#endif
#ifndef MPCX_SLACK_FLOOR
#define MPCX_SLACK_FLOOR 0.5        /* starting point of the iteration: s = max(slack, floor), lam = MPCX_LAM0 */
#endif
#ifndef MPCX_LAM0
#define MPCX_LAM0 3.0
#endif
#ifndef MPCX_TRIAL_STEP
#define MPCX_TRIAL_STEP 1            /* try the unconstrained minimiser before the interior-point iteration (see `trial` below) */
#endif
/* Active-set polish (round 3; same rule in the condensed solver and in the tests' CPU checker): an interior-point iterate sits ~sqrt(mu) from the optimum
   on weakly active rows, and the low curvature of the input cost (2R = 0.02) amplifies that -- up to 1e-3 on the hard closed-loop
   problems at the reduced-accuracy exit.  Once the iterate is close (mu <= MPCX_POLISH_MU with small residuals, or at any exit) the
   rows with s < lam are taken as the active set and ONE augmented-Lagrangian solve is made on it -- a round whose barrier weights
   are rho on the active rows and 0 elsewhere, with lam_a + rho gap_a as the rows' linear term: the trial pass below is the special
   case "no active row".  "Close" is decided at the END of the step that produces the iterate (the new mu is known exactly there, the
   residuals shrink by one minus the step lengths), so that the polish round takes the place of the iterate's first row pass.  The end point is accepted only if it is a KKT point (new multipliers lam_a + rho gap_a' >= 0, no other
   row violated): then it is the minimiser up to |lam - lam*| / rho.  Otherwise rows with a negative multiplier leave the set,
   violated rows enter it, and the round is repeated, MPCX_POLISH_TRIES times in all; after that nothing is kept and the iteration goes on
   (or ends with its own iterate).  Polish rounds are not counted as iterations. */
#ifndef MPCX_POLISH
#define MPCX_POLISH 1
#endif
#ifndef MPCX_POLISH_MU
#define MPCX_POLISH_MU 1e-5         /* entry: mu, primal residual / hnorm, dual residual / gnorm predicted below these (or any exit) */
#define MPCX_POLISH_RP 1e-6
#define MPCX_POLISH_RD 1e-3
#define MPCX_POLISH_RHO 1e8         /* penalty of the augmented-Lagrangian solve */
#define MPCX_POLISH_TRIES 3
#define MPCX_POLISH_EPS_L 1e-9      /* a new multiplier below -EPS_L / a gap above EPS_G rejects the point */
#define MPCX_POLISH_EPS_G 1e-9
#endif
#define MPCX_UNROLL _Pragma("unroll")
#define MPCX_NOUNROLL _Pragma("nounroll")

namespace mpcx_stage {

// slot t (0..T-1) owns: input u_t, dynamics x_t -> x_{t+1}, the state x_{t+1} with its tracking cost, and 8 rows:
//   0: a_t <= amax   1: -a_t <= -amin   2: d_t <= smax   3: -d_t <= smax
//   4: d_t - d_{t-1} <= rmax   5: -(d_t - d_{t-1}) <= rmax   (t >= 1)
//   6: v_{t+1} <= vmax   7: -v_{t+1} <= -vmin
enum { ROWS = 8 };

struct Problem {            // one QP (pointers to that problem's rows of the batch arrays)
    const double *x0, *xref, *xbar, *u_warm;   // u_warm may be null
    const uint8_t *re;
    double *x_out, *u_out, *kkt;
    int32_t *status, *iters;
};

template <class Cx, class Src>
MPCX_HD void solve_queue(Cx &cx, Src &src) {
    constexpr int LQ = Cx::LQ, SPL = Cx::SPL;
    constexpr bool JERK = Cx::JERK;
    const int q = cx.q;
    mpcx_mpc_params P = src.params();       // weights / limits may be replaced per problem by fetch(); T, dt, L never change
    int pbi = 0;                            // index of the group's problem: its pointers are rebuilt where they are needed (set-up, hand-in)
    double PH[SPL] = {};                                      // yaw of the linearisation point (for C_t)
    double xs[4] = {0.0, 0.0, 0.0, 0.0};                      // x0
    const int T = P.T, W = T + 1;
    const double dt = P.dt;
    const double minv = 1.0 / (double)(8 * T - 2);
    const int NTURN = (T + SPL - 1) / SPL < LQ ? (T + SPL - 1) / SPL : LQ;   // lanes q >= NTURN own no stage of this horizon
    bool have = false, drained = false;     // this group holds a problem / found the queue empty (uniform within a group)

    // ------------------------------------------------------------------ per-slot constants and iterate
    double A0[SPL] = {}, A1[SPL] = {}, A2[SPL] = {}, A3[SPL] = {}, B3[SPL] = {};   // a02, a03, a12, a13, b3 of mpc.py:58-79 (delta_bar = 0)
    // 2*W_{t+1} xy block (mpc.py:157-170): the off-diagonal entries in registers, the diagonal ones in the policy's spare row storage
    // (cx.ld_w / cx.st_w, slots ls and SPL + ls): constants of a problem that are read three times per round -- the register
    // allocator kept them in scratch (HBM-backed) anyway, at a memory round trip per Riccati sweep
    double Wxy[SPL] = {};
#define WXX(ls) cx.ld_w(ls)
#define WYY(ls) cx.ld_w(SPL + (ls))
    bool ended[SPL] = {}, uend[SPL] = {};                     // x_{t+1} / u_t fall on the clipped tail of the reference (Qf / R_end)
    double U0[SPL] = {}, U1[SPL] = {};                        // a_t, delta_t
    // x_{t+1} as tracking error: X0, X1, X3 = (x, y, psi) - reference, X2 = v itself (the speed rows need it), XRV = reference speed.
    // The reference window is read once per problem; the iterate moves by alpha * dx either way.
    double X0[SPL] = {}, X1[SPL] = {}, X2[SPL] = {}, X3[SPL] = {}, XRV[SPL] = {};
    bool act[SPL] = {}, rate[SPL] = {};
    double Z0 = 0.0, jw2 = 0.0;                               // JERK: the iterate of x4_0 (uniform within a group), 2 w dt^2
    double Rda = 0.0, Rds = 0.0, wv_run = 0.0, wp_run = 0.0, wv_end = 0.0, wp_end = 0.0;
    double ra_run = 1.0, rs_run = 1.0, ra_end = 1.0, rs_end = 1.0, rmax = 0.0, hnorm = 1.0, gnorm = 1.0, tol_loose = 1e-7;
    int status = MPCX_QP_MAXITER, it = 0, loose_run = 0, max_iter = -1;
    double res_d = 0.0, res_p = 0.0, mu = 0.0;
    bool loose = false, running = false;    // uniform within a group; other groups of the wave may be in another state
    // Trial step (same rule in the condensed solver and in the tests' CPU checker): a new problem's first round is run with every multiplier taken
    // as zero -- the Riccati sweep then factorises the plain LQ problem and the predictor direction leads to its unconstrained
    // minimiser.  If that point violates no row it is the solution (lam = 0 is its exact multiplier): the group takes the full
    // step, reports 0 iterations and is done; about two thirds of the closed-loop problems end this way instead of spending four
    // interior-point iterations walking lam from 1 to 1e-10.  Otherwise nothing is kept and the iteration starts as before.
    bool trial = false, accepted = false;
    // polish rounds (see MPCX_POLISH above): the current active set, one byte of row bits per slot; rounds tried; what the group does
    // if no round is accepted (0: the iteration goes on, 1: it ends OPTIMAL with its own iterate, 2: it ends as it is -- MAXITER)
    // The active set of a polishing group is kept in the SIGN of the stored slacks (an iterate's slacks are positive): a row is in the
    // set iff the group is polishing and its stored s is negative.  Every pass takes |s| (a free source modifier) and the set costs no
    // registers.  The signs go away with the next step (of an accepted round, or of the iteration after the group gave up).
    bool polish = false, pinit = false;     // pinit: the set is still to be taken from the iterate (rows with s < lam)
    int ptries = 0, pend = 0, ptested = -1; // ptested: the iterate (by its count) that has had its polish rounds
#ifdef MPCX_STAGE_TRACE
    double trace_alpha = 0.0, trace_aff = 0.0, trace_sigma = 0.0;       // dev build: per-iteration history of one problem
    int trace_n = 0;
#endif

#define WV(ls) (act[ls] ? (ended[ls] ? wv_end : wv_run) : 0.0)
#define WP(ls) (act[ls] ? (ended[ls] ? wp_end : wp_run) : 0.0)
#define RA_(ls) (uend[ls] ? ra_end : ra_run)
#define RS_(ls) (uend[ls] ? rs_end : rs_run)
#define JW_(ls) ((JERK && act[ls] && q * SPL + (ls) + 1 < T) ? jw2 : 0.0)      /* jerk term on a_t, t <= T-2 (mpc_jerk.py:188-190) */


    // ---- serial sweeps are written as "turns": lane `turn` works on its slots, then hands its carry to the neighbour
    // forward rollout of the linear model from (u0, u1): fills (X0..X3); `free` = true uses u = 0 (free response)
    auto rollout = [&](bool free_resp, double (&Y0)[SPL], double (&Y1)[SPL], double (&Y2)[SPL], double (&Y3)[SPL]) {
        double c0 = xs[0], c1 = xs[1], c2 = xs[2], c3 = xs[3];
        double c4 = (JERK && !free_resp) ? Z0 : 0.0;
        for (int turn = 0; turn < NTURN; turn++) {
            if (q == turn) {
                MPCX_UNROLL
                for (int ls = 0; ls < SPL; ls++) {
                    {
                        const double ph = PH[ls];
                        const double ua = free_resp ? 0.0 : U0[ls], us = free_resp ? 0.0 : U1[ls];
                        // C_t = (dt v sin(phi) phi, -dt v cos(phi) phi, 0, 0) = (-a03 phi, -a13 phi, 0, 0)
                        const double n0 = c0 + A0[ls] * c2 + A1[ls] * c3 - A1[ls] * ph;
                        const double n1 = c1 + A2[ls] * c2 + A3[ls] * c3 - A3[ls] * ph;
                        double n2 = c2 + dt * ua;
                        const double n3 = c3 + B3[ls] * us;
                        if constexpr (JERK) { if (act[ls]) { n2 += dt * c4; c4 += dt * ua; } }
                        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
                    }
                    Y0[ls] = c0; Y1[ls] = c1; Y2[ls] = c2; Y3[ls] = c3;
                }
            }
            if (turn + 1 < NTURN) {
                const double t0 = cx.prv(c0), t1 = cx.prv(c1), t2 = cx.prv(c2), t3 = cx.prv(c3);
                c0 = t0; c1 = t1; c2 = t2; c3 = t3;      // every lane shifts: only the lane whose turn comes next uses what it got
                if constexpr (JERK) c4 = cx.prv(c4);
            }
        }
    };
    // backward costate sweep: p_t = qx_t + A_t' p_{t+1};  out[ls] = ru[ls] + B_t' p_{t+1} (condensed gradient entries)
    auto costate = [&](const double (&Q0)[SPL], const double (&Q1)[SPL], const double (&Q2)[SPL], const double (&Q3)[SPL],
                       const double (&G0)[SPL], const double (&G1)[SPL], double (&O0)[SPL], double (&O1)[SPL]) -> double {
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
        double p4 = 0.0;                                      // JERK: costate of x4; what lane 0 ends with is the gradient entry of x4_0
        for (int turn = NTURN - 1; turn >= 0; turn--) {
            if (q == turn) {
                MPCX_UNROLL
                for (int ls = SPL - 1; ls >= 0; ls--) {
                    {
                        p0 += Q0[ls]; p1 += Q1[ls]; p2 += Q2[ls]; p3 += Q3[ls];          // gradient of x_{t+1}
                        O0[ls] = G0[ls] + dt * p2;
                        if constexpr (JERK) { O0[ls] += dt * p4; p4 += dt * p2; }      // B' p and A' p of the fifth row / column
                        O1[ls] = G1[ls] + B3[ls] * p3;
                        const double n2 = A0[ls] * p0 + A2[ls] * p1 + p2, n3 = A1[ls] * p0 + A3[ls] * p1 + p3;
                        p2 = n2; p3 = n3;
                    }
                }
            }
            if (turn > 0) {
                const double t0 = cx.nxt(p0), t1 = cx.nxt(p1), t2 = cx.nxt(p2), t3 = cx.nxt(p3);
                p0 = t0; p1 = t1; p2 = t2; p3 = t3;
                if constexpr (JERK) p4 = cx.nxt(p4);
            }
        }
        return (JERK && q == 0) ? p4 : 0.0;
    };

    // neighbours of a per-slot value: previous slot (t-1) / next slot (t+1), across lanes where needed
    auto prev_of = [&](const double (&V)[SPL], double (&O)[SPL]) {
        const double from = cx.prv(V[SPL - 1]);
        O[0] = (q > 0) ? from : 0.0;
        MPCX_UNROLL
        for (int ls = 1; ls < SPL; ls++) O[ls] = V[ls - 1];
    };
    auto next_of = [&](const double (&V)[SPL], double (&O)[SPL]) {
        const double from = cx.nxt(V[0]);
        O[SPL - 1] = (q + 1 < LQ) ? from : 0.0;
        MPCX_UNROLL
        for (int ls = 0; ls + 1 < SPL; ls++) O[ls] = V[ls + 1];
    };

    // rows: residual of row r at the current iterate WITHOUT the slack: g_r(u, x) - h_r
    double Dprev[SPL];
    auto row_gap = [&](int ls, int r, double dprev) -> double {
        switch (r) {
            case 0: return U0[ls] - P.max_accel;
            case 1: return -U0[ls] + P.max_decel;
            case 2: return U1[ls] - P.max_steer;
            case 3: return -U1[ls] - P.max_steer;
            case 4: return (U1[ls] - dprev) - rmax;
            case 5: return -(U1[ls] - dprev) - rmax;
            case 6: return X2[ls] - P.max_speed;
            default: return -X2[ls] + P.min_speed;
        }
    };
    auto row_on = [&](int ls, int r) -> bool { return (r == 4 || r == 5) ? rate[ls] : act[ls]; };
    // slacks and multipliers of one slot, all 16 loads in flight before the first is used (left to itself the compiler loads each
    // row right before its use and waits for it: 24 exposed LDS round trips per pass at one wavefront per SIMD)
    auto rows_of = [&](int ls, double (&sv)[ROWS], double (&lv)[ROWS]) {
        MPCX_UNROLL
        for (int r = 0; r < ROWS; r++) { sv[r] = cx.ld_s(ls * ROWS + r); lv[r] = cx.ld_l(ls * ROWS + r); }
    };
    // gradient of the cost (no multipliers) wrt x_{t+1} and u_t at the current iterate; recomputed where needed rather than
    // kept across the Riccati sweep (30 doubles per lane that the sweep needs for the cost-to-go)
    auto cost_grad = [&](double (&G0)[SPL], double (&G1)[SPL], double (&G2)[SPL], double (&G3)[SPL], double (&H0)[SPL], double (&H1)[SPL]) {
        double Un0[SPL], Un1[SPL], Up0[SPL], Up1[SPL];
        prev_of(U1, Up1); prev_of(U0, Up0); next_of(U0, Un0); next_of(U1, Un1);
        MPCX_UNROLL
        for (int ls = 0; ls < SPL; ls++) {
            const int t = q * SPL + ls;
            const double e0 = X0[ls], e1 = X1[ls], e2 = X2[ls] - XRV[ls], e3 = X3[ls];
            G0[ls] = WXX(ls) * e0 + Wxy[ls] * e1; G1[ls] = Wxy[ls] * e0 + WYY(ls) * e1; G2[ls] = WV(ls) * e2; G3[ls] = WP(ls) * e3;
            const bool has_next = act[ls] && (t + 1 < T);
            double g0 = (RA_(ls) + JW_(ls)) * U0[ls], g1 = RS_(ls) * U1[ls];
            if (rate[ls]) { g0 += Rda * (U0[ls] - Up0[ls]); g1 += Rds * (U1[ls] - Up1[ls]); }
            if (has_next) { g0 -= Rda * (Un0[ls] - U0[ls]); g1 -= Rds * (Un1[ls] - U1[ls]); }
            H0[ls] = g0; H1[ls] = g1;               // zero on slots beyond the horizon (u = 0 there, no rate terms)
        }
    };

    // ------------------------------------------------------------------ a new problem enters the group: constants, scaling norms,
    // initial point (same rules as the condensed solver).  With have == false everything is zeroed, so the group idles harmlessly.
    auto setup = [&]() {
        const bool valid = have;
        const Problem pb = src.at(pbi);
        // every global load of the new problem is issued here, before anything waits for one (the refill is run ~16 times per
        // wavefront with all eight groups waiting for it; the reference window used to be read twice, each time behind a sweep)
        double in_vb[SPL], in_ph[SPL], in_yr[SPL], in_r0[SPL], in_r1[SPL], in_r2[SPL], in_u0[SPL], in_u1[SPL];
        bool in_end[SPL], in_uend[SPL];
        {
            // loads first, unconditionally and from addresses that are valid for every lane (a group without a problem points at
            // problem 0, slots beyond the horizon at a clamped column): a load under a lane predicate becomes its own exec-masked
            // block with a full wait behind it, and this path had ten of those in a row.  Selects afterwards.
            uint8_t e1[SPL], e0[SPL];
            MPCX_UNROLL
            for (int ls = 0; ls < SPL; ls++) {
                const int t = q * SPL + ls;
                const int tc = t < T ? t : T - 1;
                const int tn = t + 1 <= T ? t + 1 : T;
                in_vb[ls] = pb.xbar[2 * W + tc]; in_ph[ls] = pb.xbar[3 * W + tc];
                e1[ls] = pb.re[tc + 1]; e0[ls] = pb.re[tc];
                in_r0[ls] = pb.xref[0 * W + tn]; in_r1[ls] = pb.xref[1 * W + tn];
                in_r2[ls] = pb.xref[2 * W + tn]; in_yr[ls] = pb.xref[3 * W + tn];
                in_u0[ls] = 0.0; in_u1[ls] = 0.0;
            }
            if (pb.u_warm) {            // the same for every lane of the wavefront (a launch has warm starts or it has none)
                MPCX_UNROLL
                for (int ls = 0; ls < SPL; ls++) {
                    const int t = q * SPL + ls;
                    const int tc = t < T ? t : T - 1;
                    in_u0[ls] = pb.u_warm[tc]; in_u1[ls] = pb.u_warm[T + tc];
                }
            }
            MPCX_UNROLL
            for (int ls = 0; ls < SPL; ls++) {
                const int t = q * SPL + ls;
                const bool a_ = valid && t < T;
                in_vb[ls] = a_ ? in_vb[ls] : 0.0; in_ph[ls] = a_ ? in_ph[ls] : 0.0;
                in_end[ls] = a_ && e1[ls] != 0; in_uend[ls] = a_ && e0[ls] != 0;
                in_r0[ls] = valid ? in_r0[ls] : 0.0; in_r1[ls] = valid ? in_r1[ls] : 0.0;
                in_r2[ls] = valid ? in_r2[ls] : 0.0; in_yr[ls] = valid ? in_yr[ls] : 0.0;
                in_u0[ls] = a_ ? in_u0[ls] : 0.0; in_u1[ls] = a_ ? in_u1[ls] : 0.0;
            }
        }
        {
            const double v0 = pb.x0[0], v1 = pb.x0[1], v2 = pb.x0[2], v3 = pb.x0[3];
            xs[0] = valid ? v0 : 0.0; xs[1] = valid ? v1 : 0.0; xs[2] = valid ? v2 : 0.0; xs[3] = valid ? v3 : 0.0;
        }
        const double x02 = xs[2];
    
        MPCX_UNROLL
        for (int ls = 0; ls < SPL; ls++) {
            const int t = q * SPL + ls;
            act[ls] = valid && t < T;
            rate[ls] = act[ls] && t >= 1;
            const double vb = in_vb[ls], ph = in_ph[ls];
            double sn, cs;
            sincos(ph, &sn, &cs);
            A0[ls] = dt * cs; A1[ls] = -dt * vb * sn; A2[ls] = dt * sn; A3[ls] = dt * vb * cs; B3[ls] = dt * vb / P.L;
            ended[ls] = in_end[ls];
            const double yr = act[ls] ? in_yr[ls] : 0.0;
            sincos(yr, &sn, &cs);
            // perpendicular projector [[s^2, -sc], [-sc, c^2]] * w_perp + parallel projector [[c^2, cs], [cs, s^2]] * w_para
            double wxx = 2.0 * (ended[ls] ? P.Qf[0] : (sn * sn) * P.w_perp + (cs * cs) * P.w_para);
            Wxy[ls] = 2.0 * (ended[ls] ? 0.0 : (-sn * cs) * P.w_perp + (cs * sn) * P.w_para);
            double wyy = 2.0 * (ended[ls] ? P.Qf[1] : (cs * cs) * P.w_perp + (sn * sn) * P.w_para);
            uend[ls] = in_uend[ls];
            U0[ls] = in_u0[ls];
            U1[ls] = in_u1[ls];
            // slots beyond the horizon (and groups beyond the batch) carry all-zero data: every sweep below passes through them
            // unchanged (zero dynamics, zero weights, rows off), so the code needs no per-slot branches
            if (!act[ls]) { A0[ls] = A1[ls] = A2[ls] = A3[ls] = B3[ls] = 0.0; wxx = Wxy[ls] = wyy = 0.0; }
            cx.st_w(ls, wxx); cx.st_w(SPL + ls, wyy);
            PH[ls] = ph;
        }
        Rda = 2.0 * P.Rd[0]; Rds = 2.0 * P.Rd[1];
        if constexpr (JERK) { jw2 = 2.0 * P.jerk_weight * dt * dt; Z0 = 0.0; }
        wv_run = 2.0 * P.Q_v_yaw[0]; wp_run = 2.0 * P.Q_v_yaw[1]; wv_end = 2.0 * P.Qf[2]; wp_end = 2.0 * P.Qf[3];
        ra_run = 2.0 * P.R[0]; rs_run = 2.0 * P.R[1]; ra_end = 2.0 * P.R_end[0]; rs_end = 2.0 * P.R_end[1];
        status = MPCX_QP_MAXITER; it = 0;
        res_d = 0.0; res_p = 0.0; mu = 0.0;
        const bool feasible0 = !(x02 > P.max_speed + 1e-9 || x02 < P.min_speed - 1e-9);
        gnorm = 1.0;
        {
            double F0[SPL] = {}, F1[SPL] = {}, F2[SPL] = {}, F3[SPL] = {}, Q0[SPL], Q1[SPL], Q2[SPL], Q3[SPL], Z[SPL], O0[SPL] = {}, O1[SPL] = {};
            rollout(true, F0, F1, F2, F3);
            MPCX_UNROLL
            for (int ls = 0; ls < SPL; ls++) {
                const double e0 = F0[ls] - in_r0[ls], e1 = F1[ls] - in_r1[ls];
                const double e2 = F2[ls] - in_r2[ls], e3 = F3[ls] - in_yr[ls];
                Q0[ls] = WXX(ls) * e0 + Wxy[ls] * e1; Q1[ls] = Wxy[ls] * e0 + WYY(ls) * e1; Q2[ls] = WV(ls) * e2; Q3[ls] = WP(ls) * e3;
                Z[ls] = 0.0;
            }
            const double gz = costate(Q0, Q1, Q2, Q3, Z, Z, O0, O1);
            double gm = fabs(gz);
            MPCX_UNROLL
            for (int ls = 0; ls < SPL; ls++) gm = fmax(gm, fmax(fabs(O0[ls]), fabs(O1[ls])));
            gnorm = fmax(1.0, cx.gmax(gm));
        }
        rmax = P.max_dsteer * dt;
        hnorm = fmax(fmax(fmax(1.0, fabs(P.max_accel)), fmax(fabs(P.max_decel), fabs(P.max_steer))),
                                  fmax(fabs(rmax), fmax(fabs(P.max_speed - x02), fabs(x02 - P.min_speed))));
        rollout(false, X0, X1, X2, X3);
        MPCX_UNROLL
        for (int ls = 0; ls < SPL; ls++) {
            X0[ls] -= in_r0[ls]; X1[ls] -= in_r1[ls]; X3[ls] -= in_yr[ls];
            XRV[ls] = in_r2[ls];
        }
        prev_of(U1, Dprev);
        MPCX_UNROLL
        for (int ls = 0; ls < SPL; ls++)
            MPCX_UNROLL
            for (int r = 0; r < ROWS; r++) {
                const double si = -row_gap(ls, r, Dprev[ls]);
                cx.st_s(ls * ROWS + r, row_on(ls, r) ? (si > MPCX_SLACK_FLOOR ? si : MPCX_SLACK_FLOOR) : 1.0);
                cx.st_l(ls * ROWS + r, row_on(ls, r) ? MPCX_LAM0 : 0.0);
            }
    
        tol_loose = P.tol > 1e-7 ? P.tol : 1e-7;
        loose_run = 0; loose = false;
        max_iter = (valid && feasible0) ? P.max_iter : -1;
        if (!feasible0) status = MPCX_QP_INFEASIBLE;
        running = valid && feasible0;
        trial = running && MPCX_TRIAL_STEP != 0;
        accepted = false;
        polish = false; pinit = false; ptries = 0; pend = 0; ptested = -1;
    };
    // ------------------------------------------------------------------ a finished problem leaves: u, x = rollout of the linear model
    auto emit = [&]() {
        const Problem pb = src.at(pbi);
        rollout(false, X0, X1, X2, X3);
        {
            MPCX_UNROLL
            for (int ls = 0; ls < SPL; ls++)
                if (act[ls]) {
                    const int t = q * SPL + ls;
                    pb.u_out[t] = U0[ls]; pb.u_out[T + t] = U1[ls];
                    pb.x_out[0 * W + t + 1] = X0[ls]; pb.x_out[1 * W + t + 1] = X1[ls];
                    pb.x_out[2 * W + t + 1] = X2[ls]; pb.x_out[3 * W + t + 1] = X3[ls];
                }
            if (q == 0) {
                pb.x_out[0] = xs[0]; pb.x_out[W] = xs[1]; pb.x_out[2 * W] = xs[2]; pb.x_out[3 * W] = xs[3];
                *pb.status = status; *pb.iters = it;
                pb.kkt[0] = res_d; pb.kkt[1] = res_p; pb.kkt[2] = mu; pb.kkt[3] = 0.0;
            }
        }
    };

    // ------------------------------------------------------------------ main loop: one interior-point iteration per round for every group
    // that is running; a group that is not (finished, or never started) first hands in its solution and draws the next
    // problem.  `cx.any` keeps the wavefront together (cross-lane operations need every lane in the loop); the loop is
    // bounded by construction: each round either advances an iteration counter or consumes a ticket.
    const long rounds = src.max_rounds();
    have = src.fetch(cx, P, pbi);           // every group draws its first problem
    drained = !have;
    setup();
    for (long guard = 0; guard < rounds; guard++) {
        double DSa[SPL], DSd[SPL], DSr[SPL], DSv[SPL];  // barrier weights d = lam/s summed over the row pairs
        double PG0[SPL], PG1[SPL], PG2[SPL], PG3[SPL], PH0[SPL], PH1[SPL];     // predictor gradient wrt x_{t+1}, u_t
        double n_mu = 0.0;
        bool fresh = guard == 0;        // the group's problem entered in this round (uniform within a group)
        // A round opens with the residuals of every group's iterate and the exit tests.  Groups that are done hand in their solution
        // and draw the next problem RIGHT THERE, and the residual pass is repeated (for everybody: the running groups get the
        // same numbers again), so that a new problem starts its first iteration in the round it arrives in.  (Until round 2 the
        // refill sat in front of the residual pass, and every problem paid one whole round just to be told it had converged.)
        MPCX_NOUNROLL
        for (int pass = 0; pass < 2; pass++) {
            if (cx.any(pinit)) {            // groups that start polishing at an exit mark the rows their iterate holds active (rare: the usual
                MPCX_NOUNROLL               // entry is decided by the step, which writes the marks itself; rows that are off hold s = 1, lam = 0)
                for (int k = 0; k < SPL * ROWS; k++) {
                    const double sk = fabs(cx.ld_s(k)), lk = cx.ld_l(k);
                    if (pinit) cx.st_s(k, sk < lk ? -sk : sk);
                }
                pinit = false;
                cx.fence();
            }
            // ---- local pass A: rows, complementarity, gradient pieces
            const bool ipm = !(trial || polish);
            prev_of(U1, Dprev);
            double L45[SPL], N45[SPL];                    // (lam4 - lam5), (nu4 - nu5) of the predictor
            double QL2[SPL], QA2[SPL], RL0[SPL], RL1[SPL], RA0[SPL], RA1[SPL];
            double mu_s = 0.0, rp_m = 0.0;
            cost_grad(PG0, PG1, PG2, PG3, PH0, PH1);
            MPCX_UNROLL
            for (int ls = 0; ls < SPL; ls++) {
                double lam[ROWS], nu[ROWS], dd[ROWS], sv[ROWS], lv[ROWS];
                rows_of(ls, sv, lv);
                MPCX_UNROLL
                for (int r = 0; r < ROWS; r++) {
                    const bool on = row_on(ls, r);
                    // A polish round puts weight rho on the rows of the active set, with lam + rho gap as their linear term (lam = the
                    // iterate's multiplier), and nothing on the other rows (a trial round has no active row at all).  That is what the
                    // iteration's own formulas d = lam / s, nu = d (s + gap) give for a row with the slack lam / rho -- so an active
                    // row is an ordinary row with that slack, an inactive one a row with a zero multiplier, and no pass below has polish
                    // arithmetic of its own.
                    const bool pa = polish & (sv[r] < 0.0);  // the row is in a polishing group's active set (bitwise: a lane-varying && / || compiles to a branch per row)
                    const double s = pa ? lv[r] * (1.0 / MPCX_POLISH_RHO) : fabs(sv[r]);
                    const double l = (pa | (on & ipm)) ? lv[r] : 0.0;
                    const double is = cx.rcp(s);
                    const double rp = on ? s + row_gap(ls, r, Dprev[ls]) : 0.0;
                    const double d = l * is;
                    dd[r] = d;
                    lam[r] = l;
                    nu[r] = d * rp;
                    mu_s += s * l;
                    rp_m = fmax(rp_m, fabs(rp));
                }
                DSa[ls] = dd[0] + dd[1]; DSd[ls] = dd[2] + dd[3]; DSr[ls] = dd[4] + dd[5]; DSv[ls] = dd[6] + dd[7];
                RL0[ls] = lam[0] - lam[1]; RL1[ls] = lam[2] - lam[3]; L45[ls] = lam[4] - lam[5]; QL2[ls] = lam[6] - lam[7];
                RA0[ls] = nu[0] - nu[1]; RA1[ls] = nu[2] - nu[3]; N45[ls] = nu[4] - nu[5]; QA2[ls] = nu[6] - nu[7];
            }
            double L45n[SPL], N45n[SPL];
            next_of(L45, L45n); next_of(N45, N45n);
            cx.stamp(1);                    // [local pass A]
            // ---- dual residual: costate sweep with the multipliers
            double O0[SPL] = {}, O1[SPL] = {};
            double rd_z = 0.0;
            {
                double Q2t[SPL], R0t[SPL], R1t[SPL];
                MPCX_UNROLL
                for (int ls = 0; ls < SPL; ls++) {
                    Q2t[ls] = PG2[ls] + QL2[ls];
                    R0t[ls] = PH0[ls] + RL0[ls];
                    R1t[ls] = PH1[ls] + RL1[ls] + L45[ls] - L45n[ls];
                }
                rd_z = costate(PG0, PG1, Q2t, PG3, R0t, R1t, O0, O1);
            }
            // from here on PG / PH hold the predictor's linear terms (cost gradient + G' nu with nu = d * rp)
            MPCX_UNROLL
            for (int ls = 0; ls < SPL; ls++) {
                PG2[ls] += QA2[ls];
                PH0[ls] += RA0[ls];
                PH1[ls] += RA1[ls] + N45[ls] - N45n[ls];
            }
            double rd_m = fabs(rd_z);
            MPCX_UNROLL
            for (int ls = 0; ls < SPL; ls++) rd_m = fmax(rd_m, fmax(fabs(O0[ls]), fabs(O1[ls])));
            const double n_rd = cx.gmax(rd_m), n_rp = cx.gmax(rp_m);
            n_mu = cx.gsum(mu_s) * minv;
            if (running && !polish) { res_d = n_rd; res_p = n_rp; mu = n_mu; }
            const bool test = running && !trial && !polish && (pass == 0 || fresh);      // a group takes the exit tests once per iterate
            if (test && accepted) { status = MPCX_QP_OPTIMAL; running = false; }     // residuals of the accepted trial point: measured above, for the report
#ifdef MPCX_STAGE_TRACE
            if (test) cx.trace(it, res_d, res_p, mu, trace_alpha, trace_aff, trace_sigma);
#endif
            cx.stamp(2);                    // [costate sweep]
            // ---- exit tests (uniform per group)
            // a group that reaches an exit without having polished this iterate polishes it now: its row pass is redone with the polish
            // weights (rare: the usual entry is decided at the end of the step, see below)
            bool entered = false;
            if (test && running) {
                const bool conv = res_d <= P.tol * gnorm && res_p <= P.tol * hnorm && mu <= P.tol;
                loose = (res_d <= tol_loose * gnorm && res_p <= tol_loose * hnorm && mu <= tol_loose);
                loose_run = loose ? loose_run + 1 : 0;
                const bool stop_ok = conv || loose_run >= 4 || (it == max_iter && loose);
                const bool stop_fail = it == max_iter && !loose;
                if (MPCX_POLISH != 0 && (stop_ok || stop_fail) && ptested != it) {
                    polish = true; pinit = true; entered = true; ptries = 0; pend = stop_ok ? 1 : 2;
                } else if (stop_ok) { status = MPCX_QP_OPTIMAL; running = false; }
                else if (stop_fail) running = false;
            }
            if (pass == 1) break;
            // hand-in / draw / set-up costs the whole wavefront a few thousand instructions: do it when at least `refill_min`
            // groups are waiting, or when nobody is running any more
            const bool need = !running && !drained;
            if (!(cx.count(need) >= src.refill_min() || (cx.any(need) && !cx.any(running)) || cx.any(entered))) break;
            fresh = false;
            if (need) {                  // uniform within a group: the DPP operations inside stay inside the group
#ifdef MPCX_STAGE_PROFILE
                if (have) cx.lifetime(pbi, 1, (int)guard);
#endif
                if (have) emit();
                have = src.fetch(cx, P, pbi);
#ifdef MPCX_STAGE_PROFILE
                if (have) cx.lifetime(pbi, 0, (int)guard);
#endif
                drained = !have;         // the queue is empty: zero the group's data once and idle from now on
                setup();
                fresh = true;
            }
            cx.stamp(0);                    // [refill / set-up]
        }
        if (!cx.any(have)) break;
#ifdef MPCX_STAGE_PROFILE
        cx.occupancy((int)guard, cx.count(running) / Cx::LQ, cx.count(running && (trial || polish)) / Cx::LQ);      // dev build: groups at work in this round
#endif

        cx.fence();
        // ---- backward sweep: Riccati factorisation + predictor gains.  Carry: cost-to-go Hessian Pm (6x6 symmetric, 21
        // entries, row-major upper) and gradient pv (6) at z_{t+1} = (dx_{t+1}, du_t), EXCLUDING x_{t+1}'s own stage cost.
        // gains: the 2x4 block of K that multiplies (dx, dy, dv, dpsi) lives in the policy's storage (cx.st_k / cx.ld_k, 8 per
        // slot); the 2x2 block that multiplies the previous input pair is Huu^-1 diag(rho_a, rho_d) and is rebuilt from the
        // pivots (I1, I2, LL) and the rate weights where it is used
        double KA0[SPL] = {}, KA1[SPL] = {}, I1[SPL] = {}, I2[SPL] = {}, LL[SPL] = {}, RHd[SPL] = {};
        bool RHon[SPL] = {};
        double KZa[SPL] = {}, KZd[SPL] = {};       // JERK: the gain column that multiplies dx4 (registers), 1 / (cost-to-go curvature in x4_0),
        double I66 = 1.0, DZA = 0.0;               // and the predictor's step of x4_0
        auto prev_block = [&](int ls, double &ka4, double &ka5, double &kd4, double &kd5) {
            const double ra = RHon[ls] ? Rda : 0.0, rd = RHd[ls];
            kd4 = -LL[ls] * ra * I2[ls]; ka4 = ra * I1[ls] - LL[ls] * kd4;
            kd5 = rd * I2[ls];           ka5 = -LL[ls] * kd5;
        };
        bool bad = false;
        {
            constexpr int NZ = JERK ? 7 : 6, NP = NZ * (NZ + 1) / 2;
            double Pm[NP], pv[NZ];
            MPCX_UNROLL
            for (int i = 0; i < NP; i++) Pm[i] = 0.0;
            MPCX_UNROLL
            for (int i = 0; i < NZ; i++) pv[i] = 0.0;
            for (int turn = NTURN - 1; turn >= 0; turn--) {
                if (q == turn) {
                    MPCX_UNROLL
                    for (int ls = SPL - 1; ls >= 0; ls--) {
                      if constexpr (JERK) {
                        // seven states z = (dx, dy, dv, dpsi, a-, d-, dx4).  Columns of F = d z_{t+1} / d (z_t, a, delta):
                        //   x: e0   y: e1   v: a0 e0 + a2 e1 + e2   psi: a1 e0 + a3 e1 + e3   a-, d-: 0
                        //   x4: dt e2 + e6   a: dt e2 + e4 + dt e6   delta: b3 e3 + e5
#define PX7(i, j) ((i) * 7 - (i) * ((i) + 1) / 2 + (j))
                        double Pf[7][7], g[7];
                        MPCX_UNROLL
                        for (int i = 0; i < 7; i++)
                            MPCX_UNROLL
                            for (int j = i; j < 7; j++) { Pf[i][j] = Pm[PX7(i, j)]; Pf[j][i] = Pf[i][j]; }
                        // 1. own state cost of x_{t+1} (+ speed barrier) and its gradient
                        Pf[0][0] += WXX(ls); Pf[0][1] += Wxy[ls]; Pf[1][0] += Wxy[ls]; Pf[1][1] += WYY(ls);
                        Pf[2][2] += WV(ls) + DSv[ls]; Pf[3][3] += WP(ls);
                        MPCX_UNROLL
                        for (int i = 0; i < 7; i++) g[i] = pv[i];
                        g[0] += PG0[ls]; g[1] += PG1[ls]; g[2] += PG2[ls]; g[3] += PG3[ls];
                        const double a0 = A0[ls], a1 = A1[ls], a2 = A2[ls], a3 = A3[ls], b3 = B3[ls];
                        // 2. G = P F, one 7-vector per column of F
                        double Gv[7], Gp[7], Gz[7], Ga[7], Gd[7];
                        MPCX_UNROLL
                        for (int i = 0; i < 7; i++) {
                            Gv[i] = a0 * Pf[i][0] + a2 * Pf[i][1] + Pf[i][2];
                            Gp[i] = a1 * Pf[i][0] + a3 * Pf[i][1] + Pf[i][3];
                            Gz[i] = dt * Pf[i][2] + Pf[i][6];
                            Ga[i] = dt * (Pf[i][2] + Pf[i][6]) + Pf[i][4];
                            Gd[i] = b3 * Pf[i][3] + Pf[i][5];
                        }
                        // 3. Phi = F' G through the column dot products
                        auto Fv = [&](const double (&w)[7]) { return a0 * w[0] + a2 * w[1] + w[2]; };
                        auto Fp = [&](const double (&w)[7]) { return a1 * w[0] + a3 * w[1] + w[3]; };
                        auto Fz = [&](const double (&w)[7]) { return dt * w[2] + w[6]; };
                        auto Fa = [&](const double (&w)[7]) { return dt * (w[2] + w[6]) + w[4]; };
                        auto Fd = [&](const double (&w)[7]) { return b3 * w[3] + w[5]; };
                        // 4. input, rate, jerk and barrier terms
                        const double rho_a = rate[ls] ? Rda : 0.0;
                        const double rho_d = rate[ls] ? Rds + DSr[ls] : 0.0;
                        const double Faa = Fa(Ga) + RA_(ls) + JW_(ls) + DSa[ls] + rho_a;
                        const double Fad = Fa(Gd);
                        const double Fdd = Fd(Gd) + RS_(ls) + DSd[ls] + rho_d;
                        // 5. eliminate (a, delta)
                        bad = bad || !(Faa > 0.0);
                        const double i1 = cx.rcp(Faa > 0.0 ? Faa : 1.0);
                        const double l = Fad * i1;
                        const double s2 = Fdd - l * Fad;
                        bad = bad || !(s2 > 0.0);
                        const double i2 = cx.rcp(s2 > 0.0 ? s2 : 1.0);
                        I1[ls] = i1; I2[ls] = i2; LL[ls] = l; RHd[ls] = rho_d; RHon[ls] = rate[ls];
                        const double za[7] = {Ga[0], Ga[1], Fv(Ga), Fp(Ga), -rho_a, 0.0, Fz(Ga)};
                        const double zd[7] = {Gd[0], Gd[1], Fv(Gd), Fp(Gd), 0.0, -rho_d, Fz(Gd)};
                        double ka[7], kd[7];
                        MPCX_UNROLL
                        for (int c = 0; c < 7; c++) {
                            const double wd = zd[c] - l * za[c];
                            kd[c] = -wd * i2;
                            ka[c] = -za[c] * i1 - l * kd[c];
                            if (c < 4) { cx.st_k(ls * 8 + c, ka[c]); cx.st_k(ls * 8 + 4 + c, kd[c]); }
                        }
                        KZa[ls] = ka[6]; KZd[ls] = kd[6];
                        const double hu0 = Fa(g) + PH0[ls];
                        const double hu1 = Fd(g) + PH1[ls];
                        const double wd = hu1 - l * hu0;
                        const double k1 = -wd * i2, k0 = -hu0 * i1 - l * k1;
                        KA0[ls] = k0; KA1[ls] = k1;
                        const double hz[7] = {g[0], g[1], Fv(g), Fp(g), 0.0, 0.0, Fz(g)};
                        // 6. new cost-to-go at z_t (without x_t's own cost): Hzz + Huz' K,  hz + K' hu
                        const double zz[28] = {Pf[0][0], Pf[0][1], Gv[0], Gp[0], 0.0, 0.0, Gz[0],
                                               Pf[1][1], Gv[1], Gp[1], 0.0, 0.0, Gz[1],
                                               Fv(Gv), Fv(Gp), 0.0, 0.0, Fv(Gz),
                                               Fp(Gp), 0.0, 0.0, Fp(Gz),
                                               rho_a, 0.0, 0.0,
                                               rho_d, 0.0,
                                               Fz(Gz)};
                        MPCX_UNROLL
                        for (int i = 0; i < 7; i++)
                            MPCX_UNROLL
                            for (int j = i; j < 7; j++) Pm[PX7(i, j)] = zz[PX7(i, j)] + za[i] * ka[j] + zd[i] * kd[j];
                        MPCX_UNROLL
                        for (int i = 0; i < 7; i++) pv[i] = hz[i] + ka[i] * hu0 + kd[i] * hu1;
                      } else {
                        // index of (i,j), i<=j, in the packed upper triangle of a 6x6
#define PX(i, j) ((i) * 6 - (i) * ((i) + 1) / 2 + (j))
                        // 1. own state cost of x_{t+1} (+ speed barrier) and its gradient
                        const double p00 = Pm[PX(0, 0)] + WXX(ls), p01 = Pm[PX(0, 1)] + Wxy[ls], p11 = Pm[PX(1, 1)] + WYY(ls);
                        const double p02 = Pm[PX(0, 2)], p03 = Pm[PX(0, 3)], p04 = Pm[PX(0, 4)], p05 = Pm[PX(0, 5)];
                        const double p12 = Pm[PX(1, 2)], p13 = Pm[PX(1, 3)], p14 = Pm[PX(1, 4)], p15 = Pm[PX(1, 5)];
                        const double p22 = Pm[PX(2, 2)] + WV(ls) + DSv[ls];
                        const double p23 = Pm[PX(2, 3)], p24 = Pm[PX(2, 4)], p25 = Pm[PX(2, 5)];
                        const double p33 = Pm[PX(3, 3)] + WP(ls), p34 = Pm[PX(3, 4)], p35 = Pm[PX(3, 5)];
                        const double p44 = Pm[PX(4, 4)], p45 = Pm[PX(4, 5)], p55 = Pm[PX(5, 5)];
                        const double g0 = pv[0] + PG0[ls], g1 = pv[1] + PG1[ls], g2 = pv[2] + PG2[ls], g3 = pv[3] + PG3[ls];
                        const double g4 = pv[4], g5 = pv[5];
                        const double a0 = A0[ls], a1 = A1[ls], a2 = A2[ls], a3 = A3[ls], b3 = B3[ls];
                        // 2. G = P * F  (columns v, psi, a, delta of F = d z_{t+1} / d (x, y, v, psi, a, delta))
                        const double Gv0 = a0 * p00 + a2 * p01 + p02, Gv1 = a0 * p01 + a2 * p11 + p12, Gv2 = a0 * p02 + a2 * p12 + p22;
                        const double Gp0 = a1 * p00 + a3 * p01 + p03, Gp1 = a1 * p01 + a3 * p11 + p13, Gp2 = a1 * p02 + a3 * p12 + p23;
                        const double Gp3 = a1 * p03 + a3 * p13 + p33;
                        const double Ga0 = dt * p02 + p04, Ga1 = dt * p12 + p14, Ga2 = dt * p22 + p24, Ga3 = dt * p23 + p34, Ga4 = dt * p24 + p44;
                        const double Gd0 = b3 * p03 + p05, Gd1 = b3 * p13 + p15, Gd2 = b3 * p23 + p25, Gd3 = b3 * p33 + p35;
                        const double Gd4 = b3 * p34 + p45, Gd5 = b3 * p35 + p55;
                        // 3. Phi = F' G
                        const double Fvv = a0 * Gv0 + a2 * Gv1 + Gv2, Fvp = a0 * Gp0 + a2 * Gp1 + Gp2;
                        const double Fva = a0 * Ga0 + a2 * Ga1 + Ga2, Fvd = a0 * Gd0 + a2 * Gd1 + Gd2;
                        const double Fpp = a1 * Gp0 + a3 * Gp1 + Gp3, Fpa = a1 * Ga0 + a3 * Ga1 + Ga3, Fpd = a1 * Gd0 + a3 * Gd1 + Gd3;
                        // 4. input, rate and barrier terms
                        const double rho_a = rate[ls] ? Rda : 0.0;
                        const double rho_d = rate[ls] ? Rds + DSr[ls] : 0.0;
                        const double Faa = dt * Ga2 + Ga4 + RA_(ls) + DSa[ls] + rho_a;
                        const double Fad = dt * Gd2 + Gd4;
                        const double Fdd = b3 * Gd3 + Gd5 + RS_(ls) + DSd[ls] + rho_d;
                        // 5. eliminate (a, delta): Huu = [[Faa, Fad], [Fad, Fdd]] = L D L'
                        bad = bad || !(Faa > 0.0);
                        const double i1 = cx.rcp(Faa > 0.0 ? Faa : 1.0);
                        const double l = Fad * i1;
                        const double s2 = Fdd - l * Fad;
                        bad = bad || !(s2 > 0.0);
                        const double i2 = cx.rcp(s2 > 0.0 ? s2 : 1.0);
                        I1[ls] = i1; I2[ls] = i2; LL[ls] = l; RHd[ls] = rho_d; RHon[ls] = rate[ls];
                        // Huz rows (columns x, y, v, psi, a-, d-)
                        const double za[6] = {Ga0, Ga1, Fva, Fpa, -rho_a, 0.0};
                        const double zd[6] = {Gd0, Gd1, Fvd, Fpd, 0.0, -rho_d};
                        double ka[6], kd[6];
                        MPCX_UNROLL
                        for (int c = 0; c < 6; c++) {
                            const double wd = zd[c] - l * za[c];
                            kd[c] = -wd * i2;
                            ka[c] = -za[c] * i1 - l * kd[c];
                            if (c < 4) { cx.st_k(ls * 8 + c, ka[c]); cx.st_k(ls * 8 + 4 + c, kd[c]); }
                        }
                        // predictor gradient: hu = ru + B'p, hz = A'p
                        const double hu0 = dt * g2 + g4 + PH0[ls];
                        const double hu1 = b3 * g3 + g5 + PH1[ls];
                        const double wd = hu1 - l * hu0;
                        const double k1 = -wd * i2, k0 = -hu0 * i1 - l * k1;
                        KA0[ls] = k0; KA1[ls] = k1;
                        const double hz[6] = {g0, g1, a0 * g0 + a2 * g1 + g2, a1 * g0 + a3 * g1 + g3, 0.0, 0.0};
                        // 6. new cost-to-go at z_t (without x_t's own cost): Hzz + Huz' K,  hz + K' hu
                        const double zz[21] = {p00, p01, Gv0, Gp0, 0.0, 0.0,
                                               p11, Gv1, Gp1, 0.0, 0.0,
                                               Fvv, Fvp, 0.0, 0.0,
                                               Fpp, 0.0, 0.0,
                                               rho_a, 0.0,
                                               rho_d};
                        MPCX_UNROLL
                        for (int i = 0; i < 6; i++)
                            MPCX_UNROLL
                            for (int j = i; j < 6; j++) Pm[PX(i, j)] = zz[PX(i, j)] + za[i] * ka[j] + zd[i] * kd[j];
                        MPCX_UNROLL
                        for (int i = 0; i < 6; i++) pv[i] = hz[i] + ka[i] * hu0 + kd[i] * hu1;
                      }
                    }
                }
                if (turn > 0) {
                    MPCX_UNROLL
                    for (int i = 0; i < NP; i++) Pm[i] = cx.nxt(Pm[i]);       // every lane shifts; lanes that had their turn no longer need theirs
                    MPCX_UNROLL
                    for (int i = 0; i < NZ; i++) pv[i] = cx.nxt(pv[i]);
                }
            }
            if constexpr (JERK) {
                // lane 0 now holds the cost-to-go at t = 0, where (dx, dy, dv, dpsi) = 0 and the rate terms are off: what is left is
                // 1/2 P66 dz^2 + pv6 dz in the free initial value of the fifth state
                const double p66 = Pm[PX7(6, 6)];
                bad = bad || (q == 0 && !(p66 > 0.0));
                I66 = cx.rcp(p66 > 0.0 ? p66 : 1.0);
                DZA = cx.gsum(q == 0 ? -pv[6] * I66 : 0.0);
            }
        }
        const bool any_bad = cx.gany(bad);
        if (running && !trial && !polish && any_bad) { status = loose ? MPCX_QP_OPTIMAL : MPCX_QP_NUMERIC; running = false; }

        // forward sweep with gains (K, kk): fills the direction (du, dx_{t+1})
        auto forward = [&](const double (&k0)[SPL], const double (&k1)[SPL], double dz0, double (&D0)[SPL], double (&D1)[SPL],
                           double (&E0)[SPL], double (&E1)[SPL], double (&E2)[SPL], double (&E3)[SPL]) {
            double z[JERK ? 7 : 6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if constexpr (JERK) z[6] = dz0;
            for (int turn = 0; turn < NTURN; turn++) {
                if (q == turn) {
                    double kk[SPL][8];           // this lane's gains, all in flight before the first is used
                    MPCX_UNROLL
                    for (int ls = 0; ls < SPL; ls++)
                        MPCX_UNROLL
                        for (int c = 0; c < 8; c++) kk[ls][c] = cx.ld_k(ls * 8 + c);
                    MPCX_UNROLL
                    for (int ls = 0; ls < SPL; ls++) {
                        {
                            double da = k0[ls], dd = k1[ls];
                            MPCX_UNROLL
                            for (int c = 0; c < 4; c++) { da += kk[ls][c] * z[c]; dd += kk[ls][4 + c] * z[c]; }
                            {
                                double ka4, ka5, kd4, kd5;
                                prev_block(ls, ka4, ka5, kd4, kd5);
                                da += ka4 * z[4] + ka5 * z[5]; dd += kd4 * z[4] + kd5 * z[5];
                            }
                            if constexpr (JERK) { da += KZa[ls] * z[6]; dd += KZd[ls] * z[6]; }
                            const double n0 = z[0] + A0[ls] * z[2] + A1[ls] * z[3], n1 = z[1] + A2[ls] * z[2] + A3[ls] * z[3];
                            double n2 = z[2] + dt * da;
                            const double n3 = z[3] + B3[ls] * dd;
                            if constexpr (JERK) { if (act[ls]) { n2 += dt * z[6]; z[6] += dt * da; } }
                            z[0] = n0; z[1] = n1; z[2] = n2; z[3] = n3; z[4] = da; z[5] = dd;
                            D0[ls] = da; D1[ls] = dd;
                        }
                        E0[ls] = z[0]; E1[ls] = z[1]; E2[ls] = z[2]; E3[ls] = z[3];
                    }
                }
                if (turn + 1 < NTURN) {
                    MPCX_UNROLL
                    for (int i = 0; i < (JERK ? 7 : 6); i++) z[i] = cx.prv(z[i]);
                }
            }
        };
        // row direction g_r' (du, dx)
        auto row_dir = [&](int r, double da, double dd, double ddprev, double dv) -> double {
            switch (r) {
                case 0: return da;
                case 1: return -da;
                case 2: return dd;
                case 3: return -dd;
                case 4: return dd - ddprev;
                case 5: return -(dd - ddprev);
                case 6: return dv;
                default: return -dv;
            }
        };

        cx.fence();
        cx.stamp(3);                    // [Riccati sweep]
        // ---- predictor direction, affine step length, centring parameter
        double DA0[SPL] = {}, DA1[SPL] = {}, EA0[SPL] = {}, EA1[SPL] = {}, EA2[SPL] = {}, EA3[SPL] = {}, DAp[SPL];
        forward(KA0, KA1, DZA, DA0, DA1, EA0, EA1, EA2, EA3);
        cx.stamp(4);                    // [forward sweep 1]
        prev_of(DA1, DAp);
        double al = 1.0, c1 = 0.0, c2 = 0.0;
        bool viol = false;              // trial / polish: the predictor's end point is no KKT point (a row violated, a multiplier negative)
        const bool ipm = !(trial || polish);
        const double veps = trial ? 0.0 : MPCX_POLISH_EPS_G;
        // per row, recomputed from (s, lam, u, x) wherever needed instead of being kept: rp = s + gap, d = lam / s
        MPCX_UNROLL
        for (int ls = 0; ls < SPL; ls++) {
            double sv[ROWS], lv[ROWS];
            rows_of(ls, sv, lv);
            MPCX_UNROLL
            for (int r = 0; r < ROWS; r++) {
                const bool on = row_on(ls, r);
                const bool pa = polish & (sv[r] < 0.0);
                const double sa = fabs(sv[r]);
                const double s = pa ? lv[r] * (1.0 / MPCX_POLISH_RHO) : sa;
                const double l = (pa | (on & ipm)) ? lv[r] : 0.0;
                const double gap = row_gap(ls, r, Dprev[ls]), dir = row_dir(r, DA0[ls], DA1[ls], DAp[ls], EA2[ls]);
                const double rp = on ? s + gap : 0.0;
                const double d = l * cx.rcp(s);
                const double dsa = on ? -rp - dir : 0.0;
                const double dla = -l - d * dsa;
                // the end point of a trial / polish round: rows outside the active set must hold, the new multipliers of the rows inside it
                // (lam + dla = lam + rho gap') must not be negative
                const bool o_in = l + dla < -MPCX_POLISH_EPS_L, o_out = !(gap + dir <= veps);      // (both evaluated: no branch per row)
                viol = viol | (pa & o_in) | (!pa & on & o_out);
                (void)sa;
                // ratio tests: any value <= the exact ratio is a valid step bound, the 0.001 margin of MPCX_STEP_FRACTION is 12 orders above rcp_fast's error
                al = fmin(al, (on && dsa < 0.0) ? -s * cx.rcp_fast(dsa) : 1.0);
                al = fmin(al, (on && dla < 0.0) ? -l * cx.rcp_fast(dla) : 1.0);
                c1 += on ? s * dla + l * dsa : 0.0;
                c2 += dsa * dla;
            }
        }
        const double alpha_aff = cx.gmin(al);
        // A trial group goes through the rest of the round with its multipliers still taken as zero: the corrector's right-hand side is
        // then the predictor's, the second forward sweep repeats the first, and the ordinary step below -- at full length, if no row
        // was violated -- IS the move to the unconstrained minimiser.  A rejected trial steps nowhere.
        const bool was_trial = trial;
        if (trial) {
            accepted = running && !(cx.gany(viol) || any_bad);
            trial = false;
        }
        // a polish round: accepted if the end point is a KKT point; otherwise the active set is corrected and the round repeated, or the
        // group gives up polishing -- the iterate has not been touched
        const bool was_polish = polish;
        if (polish) {
            const bool rej = cx.gany(viol) || any_bad;
            accepted = running && !rej;
#ifdef MPCX_STAGE_TRACE
            cx.trace(24 + trace_n++, (double)it, (double)ptries, rej ? 1.0 : 0.0, 0.0, 0.0, (any_bad ? 100.0 : 0.0) + (double)guard);
#endif
            if (cx.any(rej && !any_bad && ptries + 1 < MPCX_POLISH_TRIES)) {
                // rare: the active set of the next polish round -- rows with a negative multiplier leave, violated rows enter.  Rolled loop
                // over the rows of a slot: small code, few registers.
                const bool fix = rej && !any_bad;
                MPCX_UNROLL
                for (int ls = 0; ls < SPL; ls++) {
                    MPCX_NOUNROLL
                    for (int r = 0; r < ROWS; r++) {
                        const double sr = cx.ld_s(ls * ROWS + r), lr = cx.ld_l(ls * ROWS + r);
                        const bool on = row_on(ls, r);
                        const bool pa = sr < 0.0;
                        const double sa = fabs(sr);
                        const double s = pa ? lr * (1.0 / MPCX_POLISH_RHO) : sa;
                        const double l = pa ? lr : 0.0;
                        const double gd = row_gap(ls, r, Dprev[ls]) + row_dir(r, DA0[ls], DA1[ls], DAp[ls], EA2[ls]);
                        const double dsa = on ? -(s + gd) : 0.0;
                        const double dla = -l - (l * cx.rcp(s)) * dsa;
                        const bool out = pa ? (l + dla < -MPCX_POLISH_EPS_L) : (on && !(gd <= MPCX_POLISH_EPS_G));
                        if (fix) cx.st_s(ls * ROWS + r, (pa != out) ? -sa : sa);
                    }
                }
                cx.fence();
            }
            if (rej) {
                ptries++;
                if (ptries >= MPCX_POLISH_TRIES) {
                    polish = false; ptested = it;       // pend == 0: the iterate takes its exit tests in the next round and the iteration goes on
                    if (pend == 1) { status = MPCX_QP_OPTIMAL; running = false; }
                    else if (pend == 2) running = false;
                }
            } else polish = false;
        }
        // mu_aff = sum (s + a dsa)(lam + a dla) / m = mu + a c1/m + a^2 c2/m
        const double mu_aff = n_mu + alpha_aff * (cx.gsum(c1) * minv) + alpha_aff * alpha_aff * (cx.gsum(c2) * minv);
        double sigma = mu_aff / (n_mu > 0.0 ? n_mu : 1.0);
        sigma = sigma * sigma * sigma;
        // centring target, never below a tenth of the tolerance: once mu has converged, driving it further down only worsens the
        // conditioning (lam/s grows without bound) while the stationarity residual sits at its rounding floor -- a reversing ego on
        // a clipped reference (tests/golden/qp_hard2.npz) lost its fourth consecutive reduced-accuracy iterate that way and ran on
        // into garbage
        const double smu = fmax(sigma * n_mu, 0.1 * P.tol);
#ifdef MPCX_STAGE_TRACE
        trace_aff = alpha_aff; trace_sigma = sigma;
#endif

        cx.fence();
        cx.stamp(5);                    // [local pass C]
        // ---- corrector: nu = (lam rp - alpha_aff dsa dla + sigma mu) / s ; backward vector sweep with the stored gains
        double KC0[SPL] = {}, KC1[SPL] = {};
        double DZC = 0.0;               // JERK: the corrector's step of x4_0
        double RC[SPL][ROWS];           // rc / s of the corrector (kept in registers across the two corrector sweeps)
        {
            double C01[SPL], C23[SPL], C45[SPL], C67[SPL], C45n[SPL];
            MPCX_UNROLL
            for (int ls = 0; ls < SPL; ls++) {
                double nu[ROWS], sv[ROWS], lv[ROWS];
                rows_of(ls, sv, lv);
                MPCX_UNROLL
                for (int r = 0; r < ROWS; r++) {
                    const bool on = row_on(ls, r);
                    const bool pa = was_polish & (sv[r] < 0.0);
                    const bool full = on & !was_trial & !was_polish;             // a row of an ordinary iteration
                    const double s = pa ? lv[r] * (1.0 / MPCX_POLISH_RHO) : fabs(sv[r]);
                    const double l = (pa | full) ? lv[r] : 0.0;
                    const double is = cx.rcp(s);
                    const double rp = on ? s + row_gap(ls, r, Dprev[ls]) : 0.0;
                    const double dsa = on ? -rp - row_dir(r, DA0[ls], DA1[ls], DAp[ls], EA2[ls]) : 0.0;
                    const double dla = -l - (l * is) * dsa;
                    // a trial / polish round has no corrector (no second-order term, no centring): its right-hand side is the predictor's
                    // again, the second forward sweep repeats the first, and the step below -- at full length, if the round was accepted --
                    // IS the move to the end point that was checked
                    const double rc = s * l + (full ? alpha_aff * (dsa * dla) - smu : 0.0);
                    RC[ls][r] = rc * is;                                         // rc / s
                    nu[r] = l + (l * rp) * is - rc * is;
                }
                C01[ls] = nu[0] - nu[1]; C23[ls] = nu[2] - nu[3]; C45[ls] = nu[4] - nu[5]; C67[ls] = nu[6] - nu[7];
            }
            next_of(C45, C45n);
            double CG0[SPL], CG1[SPL], CG2[SPL], CG3[SPL], CH0[SPL], CH1[SPL];
            cost_grad(CG0, CG1, CG2, CG3, CH0, CH1);
            MPCX_UNROLL
            for (int ls = 0; ls < SPL; ls++) { CG2[ls] += C67[ls]; CH0[ls] += C01[ls]; CH1[ls] += C23[ls] + C45[ls] - C45n[ls]; }
            double pv[JERK ? 7 : 6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            for (int turn = NTURN - 1; turn >= 0; turn--) {
                if (q == turn) {
                    double kk[SPL][8];           // this lane's gains, all in flight before the first is used
                    MPCX_UNROLL
                    for (int ls = 0; ls < SPL; ls++)
                        MPCX_UNROLL
                        for (int c = 0; c < 8; c++) kk[ls][c] = cx.ld_k(ls * 8 + c);
                    MPCX_UNROLL
                    for (int ls = SPL - 1; ls >= 0; ls--) {
                        const double g0 = pv[0] + CG0[ls], g1 = pv[1] + CG1[ls], g2 = pv[2] + CG2[ls], g3 = pv[3] + CG3[ls];
                        double hu0 = dt * g2 + pv[4] + CH0[ls];
                        if constexpr (JERK) hu0 += dt * pv[6];
                        const double hu1 = B3[ls] * g3 + pv[5] + CH1[ls];
                        const double i1 = I1[ls], i2 = I2[ls], l = LL[ls];
                        const double wd = hu1 - l * hu0;
                        const double k1 = -wd * i2, k0 = -hu0 * i1 - l * k1;
                        KC0[ls] = k0; KC1[ls] = k1;
                        const double hz[6] = {g0, g1, A0[ls] * g0 + A2[ls] * g1 + g2, A1[ls] * g0 + A3[ls] * g1 + g3, 0.0, 0.0};
                        MPCX_UNROLL
                        for (int i = 0; i < 4; i++) pv[i] = hz[i] + kk[ls][i] * hu0 + kk[ls][4 + i] * hu1;
                        {
                            double ka4, ka5, kd4, kd5;
                            prev_block(ls, ka4, ka5, kd4, kd5);
                            pv[4] = ka4 * hu0 + kd4 * hu1; pv[5] = ka5 * hu0 + kd5 * hu1;
                        }
                        if constexpr (JERK) pv[6] = (dt * g2 + pv[6]) + KZa[ls] * hu0 + KZd[ls] * hu1;
                    }
                }
                if (turn > 0) {
                    MPCX_UNROLL
                    for (int i = 0; i < (JERK ? 7 : 6); i++) pv[i] = cx.nxt(pv[i]);
                }
            }
            if constexpr (JERK) DZC = cx.gsum(q == 0 ? -pv[6] * I66 : 0.0);
        }
        double D0[SPL] = {}, D1[SPL] = {}, E0[SPL] = {}, E1[SPL] = {}, E2[SPL] = {}, E3[SPL] = {}, Dp[SPL];
        cx.stamp(6);                    // [local pass D + corrector vector sweep]
        forward(KC0, KC1, DZC, D0, D1, E0, E1, E2, E3);
        cx.stamp(7);                    // [forward sweep 2]
        prev_of(D1, Dp);
        // the gains are dead from here on: their storage takes the multiplier step dl (the slack step ds is recomputed)
        cx.fence();
        double am_p = 1e300, am_d = 1e300;      // step bounds of the primal side (slacks) and of the multipliers, taken separately
        auto slack_step = [&](int ls, int r, double s) -> double {
            const bool on = row_on(ls, r);
            const double rp = on ? s + row_gap(ls, r, Dprev[ls]) : 0.0;
            return on ? -rp - row_dir(r, D0[ls], D1[ls], Dp[ls], E2[ls]) : 0.0;
        };
        MPCX_UNROLL
        for (int ls = 0; ls < SPL; ls++) {
            double sv[ROWS], lv[ROWS];
            rows_of(ls, sv, lv);
            MPCX_UNROLL
            for (int r = 0; r < ROWS; r++) {
                const bool on = row_on(ls, r);
                const bool pa = was_polish & (sv[r] < 0.0);
                const bool full = on & !was_trial & !was_polish;
                const double s = pa ? lv[r] * (1.0 / MPCX_POLISH_RHO) : fabs(sv[r]);
                const double l = (pa | full) ? lv[r] : 0.0;
                const double ds = slack_step(ls, r, s);
                // the multiplier step; in a trial / polish round the rows outside the active set end with a zero multiplier (written as a
                // step from the stored one, so that the update below has no special case), the rows inside it with lam + rho gap'
                const double dl = (pa | full) ? -RC[ls][r] - (l * cx.rcp(s)) * ds : -lv[r];
                cx.st_k(ls * ROWS + r, dl);
                am_p = fmin(am_p, (on && ds < 0.0) ? -s * cx.rcp_fast(ds) : 1e300);
                am_d = fmin(am_d, (on && dl < 0.0) ? -l * cx.rcp_fast(dl) : 1e300);
            }
        }
        cx.fence();
        // separate step lengths for (u, x, s) and for lam: the creeping instances are blocked by a slack in one iteration and by a
        // multiplier in the next; a common step length pays for both every time (maximum 20 -> 16 iterations on the closed-loop
        // corpus, and a launch lasts as long as its slowest problem)
        double alpha = MPCX_STEP_FRACTION * cx.gmin(am_p), alpha_d = MPCX_STEP_FRACTION * cx.gmin(am_d);
        if (alpha > 1.0 || was_trial || was_polish) alpha = 1.0;
        if (alpha_d > 1.0 || was_trial || was_polish) alpha_d = 1.0;
        // centrality safeguard: shorten until min s*lam >= 1e-3 * mean at the new point (at most 6 times)
        double mu_next = 0.0;           // mu at the point the step leads to (of the last evaluation: exact unless the step was shortened after it)
        for (int tr = 0; tr < 6; tr++) {
            double pmin = 1e300, psum = 0.0;
            MPCX_UNROLL
            for (int ls = 0; ls < SPL; ls++) {
                double sv[ROWS], lv[ROWS], dv[ROWS];
                rows_of(ls, sv, lv);
                MPCX_UNROLL
                for (int r = 0; r < ROWS; r++) dv[r] = cx.ld_k(ls * ROWS + r);
                MPCX_UNROLL
                for (int r = 0; r < ROWS; r++) {
                    const bool on = row_on(ls, r);
                    const double s = fabs(sv[r]), l = lv[r];
                    const double pr = (s + alpha * slack_step(ls, r, s)) * (l + alpha_d * dv[r]);
                    pmin = fmin(pmin, on ? pr : 1e300); psum += on ? pr : 0.0;
                }
            }
            const double gmn = cx.gmin(pmin), gsm = cx.gsum(psum);
            mu_next = gsm * minv;
            const bool ok = gmn >= 1e-3 * (gsm * minv);
            if (!cx.any(running && !was_trial && !was_polish && !ok)) break;
            if (!ok && !was_trial && !was_polish) { alpha *= 0.7; alpha_d *= 0.7; }      // (a trial / polish group takes the full step or none)
        }
#ifdef MPCX_STAGE_TRACE
        trace_alpha = alpha;
#endif
        cx.stamp(8);                    // [local pass E + safeguard]
        // ---- step
        if (running && (!(was_trial || was_polish) || accepted)) {
            // is the new iterate close enough to polish?  (same rule in the condensed solver: mu of the new point, the primal residual shrinks by
            // 1 - alpha, the dual one by about the smaller of the two step lengths.)  If so the group polishes from the next round on,
            // and the step itself marks the rows the new iterate holds active (s < lam) by the sign of the stored slack.
            const bool pnext = MPCX_POLISH != 0 && !was_trial && !was_polish && mu_next <= MPCX_POLISH_MU &&
                               (1.0 - alpha) * res_p <= MPCX_POLISH_RP * hnorm &&
                               (1.0 - (alpha < alpha_d ? alpha : alpha_d)) * res_d <= MPCX_POLISH_RD * gnorm;
            {
                double sv[SPL * ROWS], lv[SPL * ROWS], dv[SPL * ROWS];       // every load in flight before the first store
                MPCX_UNROLL
                for (int k = 0; k < SPL * ROWS; k++) { sv[k] = cx.ld_s(k); lv[k] = cx.ld_l(k); dv[k] = cx.ld_k(k); }
                MPCX_UNROLL
                for (int ls = 0; ls < SPL; ls++)
                    MPCX_UNROLL
                    for (int r = 0; r < ROWS; r++) {          // rows that are off have ds = dl = 0
                        const double s = fabs(sv[ls * ROWS + r]);
                        const double sn = s + alpha * slack_step(ls, r, s), ln = lv[ls * ROWS + r] + alpha_d * dv[ls * ROWS + r];
                        // an accepted trial / polish point: the true slack (>= 0 up to rounding; floored so that lam / s stays finite)
                        const double sw = ((was_trial | was_polish) & row_on(ls, r)) ? fmax(sn, 1e-30) : sn;
                        cx.st_s(ls * ROWS + r, (pnext & (sn < ln)) ? -sw : sw);
                        cx.st_l(ls * ROWS + r, ln);
                    }
            }
            MPCX_UNROLL
            for (int ls = 0; ls < SPL; ls++) {
                U0[ls] += alpha * D0[ls]; U1[ls] += alpha * D1[ls];
                X0[ls] += alpha * E0[ls]; X1[ls] += alpha * E1[ls]; X2[ls] += alpha * E2[ls]; X3[ls] += alpha * E3[ls];
            }
            if constexpr (JERK) Z0 += alpha * DZC;
            if (!was_trial && !was_polish) it++;
            if (pnext) { polish = true; ptries = 0; pend = 0; }
        }
        cx.stamp(9);                    // [update]
    }
#undef PX
#undef WXX
#undef WYY
#undef WV
#undef WP
#undef RA_
#undef RS_
#undef JW_
#ifdef PX7
#undef PX7
#endif
}

}  // namespace mpcx_stage
