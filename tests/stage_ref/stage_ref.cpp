// Host build of csrc/mpcx_qp_stage.h with ONE lane per problem (LQ = 1): test infrastructure that checks the algebra of
// the stage-structured solver against the oracle and lets the sanitizers see it.  Never loaded by the product path.
#include <cmath>
#include <cstring>
#include "mpcx_qp_stage.h"

namespace {
template <bool JERK_>
struct HostCx {
    static constexpr int LQ = 1, SPL = MPCX_T_MAX;
    static constexpr bool JERK = JERK_;
    int q = 0;
    double s[SPL * 8], l[SPL * 8], k[SPL * 8], w[2 * SPL];
    double prv(double v) const { return v; }
    double nxt(double v) const { return v; }
    double gmax(double v) const { return v; }
    double gmin(double v) const { return v; }
    double gsum(double v) const { return v; }
    bool gany(bool b) const { return b; }
    bool any(bool b) const { return b; }
    int count(bool b) const { return b ? 1 : 0; }
    double rcp(double v) const { return 1.0 / v; }
#ifdef STAGE_REF_EMULATE_RCP
    // the GPU's one-Newton-step reciprocal: a seed good to ~5e-8 (here: the float-rounded quotient), refined once
    double rcp_fast(double v) const { const double r = (double)(float)(1.0 / v); return fma(fma(-v, r, 1.0), r, r); }
#else
    double rcp_fast(double v) const { return 1.0 / v; }
#endif
    double rcp_seed(double v) const { return 1.0 / v; }
    void fence() const {}
    void stamp(int) const {}
    double ld_s(int k) const { return s[k]; }
    double ld_l(int k) const { return l[k]; }
    double ld_k(int j) const { return k[j]; }
    void st_s(int k, double v) { s[k] = v; }
    void st_l(int k, double v) { l[k] = v; }
    void st_k(int j, double v) { k[j] = v; }
    double ld_w(int j) const { return w[j]; }
    void st_w(int j, double v) { w[j] = v; }
};
}  // namespace

namespace {
struct HostSrc {            // a queue of exactly one problem
    const mpcx_mpc_params *p;
    mpcx_stage::Problem pb;
    int taken = 0;
    mpcx_mpc_params params() const { return *p; }
    mpcx_stage::Problem at(int) const { return pb; }
    long max_rounds() const { return (long)(p->max_iter + 6) * (MPCX_POLISH_TRIES + 1); }
    int refill_min() const { return 1; }
    template <class Cx>
    bool fetch(Cx &, mpcx_mpc_params &, int &idx) { idx = 0; return taken++ == 0; }
};
}  // namespace

extern "C" int stage_ref_solve(const mpcx_mpc_params *p, const double *x0, const double *xref, const double *xbar,
                               const uint8_t *re, const double *u_warm, double *x_out, double *u_out, int32_t *status,
                               int32_t *iters, double *kkt) {
    mpcx_stage::Problem pb{x0, xref, xbar, u_warm, re, x_out, u_out, kkt, status, iters};
    HostSrc src{p, pb};
    if (p->model == MPCX_MODEL_JERK5) {
        HostCx<true> cx;
        memset(cx.s, 0, sizeof cx.s); memset(cx.l, 0, sizeof cx.l); memset(cx.k, 0, sizeof cx.k);
        mpcx_stage::solve_queue(cx, src);
    } else {
        HostCx<false> cx;
        memset(cx.s, 0, sizeof cx.s); memset(cx.l, 0, sizeof cx.l); memset(cx.k, 0, sizeof cx.k);
        mpcx_stage::solve_queue(cx, src);
    }
    return 0;
}

#ifdef STAGE_REF_MAIN
// Sanitizer harness: reads n problems from a flat binary file (written by tests/test_stage_ref.py), solves them, writes the
// solutions.  Built with -fsanitize=address,undefined.
#include <cstdio>
#include <vector>
int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    mpcx_mpc_params p;
    int32_t n = 0;
    if (fread(&p, sizeof p, 1, f) != 1 || fread(&n, sizeof n, 1, f) != 1) return 4;
    const int T = p.T, W = T + 1;
    std::vector<double> x0(4), xref(4 * W), xbar(4 * W), uw(2 * T), x(4 * W), u(2 * T), kkt(4);
    std::vector<uint8_t> re(W);
    FILE *o = fopen(argv[2], "wb");
    for (int i = 0; i < n; i++) {
        int32_t has_warm = 0;
        if (fread(x0.data(), 8, 4, f) != 4 || fread(xref.data(), 8, 4 * W, f) != (size_t)(4 * W) || fread(xbar.data(), 8, 4 * W, f) != (size_t)(4 * W) ||
            fread(re.data(), 1, W, f) != (size_t)W || fread(&has_warm, 4, 1, f) != 1 || fread(uw.data(), 8, 2 * T, f) != (size_t)(2 * T)) return 5;
        int32_t st = -1, it = -1;
        stage_ref_solve(&p, x0.data(), xref.data(), xbar.data(), re.data(), has_warm ? uw.data() : nullptr, x.data(), u.data(), &st, &it, kkt.data());
        fwrite(&st, 4, 1, o); fwrite(&it, 4, 1, o); fwrite(u.data(), 8, 2 * T, o); fwrite(x.data(), 8, 4 * W, o); fwrite(kkt.data(), 8, 4, o);
    }
    fclose(o); fclose(f);
    return 0;
}
#endif
