// accuracy of v_rcp_f64 and of its Newton refinements on gfx950 (what frcp / frcp1 of mpcx_common.h deliver)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const double *x, double *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    double d = x[i];
    double r0 = __builtin_amdgcn_rcp(d);
    double r1 = fma(fma(-d, r0, 1.0), r0, r0);
    double r2 = fma(fma(-d, r1, 1.0), r1, r1);
    out[3 * i] = r0; out[3 * i + 1] = r1; out[3 * i + 2] = r2;
}
int main() {
    const int n = 1 << 20;
    double *h = (double *)malloc(n * sizeof(double)), *ho = (double *)malloc(3 * n * sizeof(double));
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); int e = (int)((s >> 3) % 600) - 300; h[i] = -(1.0 + u) * pow(2.0, e); }
    double *d, *o; hipMalloc(&d, n * sizeof(double)); hipMalloc(&o, 3 * n * sizeof(double));
    hipMemcpy(d, h, n * sizeof(double), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o, n);
    hipMemcpy(ho, o, 3 * n * sizeof(double), hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) { double t = 1.0 / h[i]; e0 = fmax(e0, fabs(ho[3*i]/t - 1)); e1 = fmax(e1, fabs(ho[3*i+1]/t - 1)); e2 = fmax(e2, fabs(ho[3*i+2]/t - 1)); }
    printf("max relative error over 2^20 values spanning 2^-300..2^300: seed %.3e, one Newton step %.3e, two %.3e\n", e0, e1, e2);
    return 0;
}
