// Microbenchmark 3: f64 FMA cost vs EXEC mask AND instruction-level parallelism (NACC independent accumulators), VGPR operands only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int NACC>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *cyc, int iters, const double *cin, unsigned long long mask) {
    const int lane = threadIdx.x;
    const bool on = (mask >> lane) & 1ull;
    double a[NACC];
#pragma unroll
    for (int j = 0; j < NACC; j++) a[j] = lane + j;
    const double c = cin[lane], d = cin[64 + lane];     // per-lane VGPR operands
    unsigned long long t0 = 0, t1 = 0;
    if (on) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 64 / NACC; u++) {
#pragma unroll
                for (int j = 0; j < NACC; j++) a[j] = fma(a[j], c, d);
            }
        }
        asm volatile("s_nop 0" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == __ffsll((long long)mask) - 1) cyc[blockIdx.x] = t1 - t0;
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < NACC; j++) s += a[j];
    out[blockIdx.x * 64 + lane] = s;
}

template <int NACC>
void run(int grid, int iters, unsigned long long mask, const double *cin) {
    double *out; unsigned long long *cyc;
    (void)hipMalloc(&out, grid * 64 * sizeof(double)); (void)hipMalloc(&cyc, grid * sizeof(unsigned long long));
    (void)hipMemset(cyc, 0, grid * sizeof(unsigned long long));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(64), 0, 0, out, cyc, iters, cin, mask);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(64), 0, 0, out, cyc, iters, cin, mask);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long *h = (unsigned long long *)malloc(grid * sizeof(unsigned long long));
    (void)hipMemcpy(h, cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < grid; i++) mean += (double)h[i]; mean /= grid;
    printf("ILP %2d mask %016llx (%2d lanes) grid %4d: %7.3f ms, %6.2f cycles/instr\n", NACC, mask, __builtin_popcountll(mask), grid, ms, mean / ((double)iters * 64.0));
    free(h); (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cu = p.multiProcessorCount;
    double hc[128]; for (int i = 0; i < 64; i++) { hc[i] = 0.999; hc[64 + i] = 1.0; }
    double *cin; (void)hipMalloc(&cin, sizeof(hc)); (void)hipMemcpy(cin, hc, sizeof(hc), hipMemcpyHostToDevice);
    const int iters = 5000;
    const unsigned long long masks[] = {~0ull, 0x0101010101010101ull, 0x0303030303030303ull, 0x1ull, 0xFFFFull};
    for (int wps = 1; wps <= 2; wps++) {
        const int grid = cu * 4 * wps;
        for (unsigned long long m : masks) {
            run<1>(grid, iters, m, cin); run<2>(grid, iters, m, cin); run<4>(grid, iters, m, cin); run<8>(grid, iters, m, cin); run<16>(grid, iters, m, cin); run<32>(grid, iters, m, cin);
        }
    }
    // one wave on the whole chip: no neighbours at all
    for (unsigned long long m : masks) { run<8>(1, iters, m, cin); run<32>(1, iters, m, cin); }
    return 0;
}
