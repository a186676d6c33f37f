// Microbenchmark: does an f64 FMA wave-instruction cost less when most lanes are masked off?
// Decides whether the 1/8-active "turns" of qp_quad_kernel can be made cheaper by lane placement.
// Build: hipcc -O3 --offload-arch=gfx950 masked_fma.hip -o ../../build/masked_fma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *cyc, int iters, double c) {
    const int lane = threadIdx.x;
    bool on;
    if (MODE == 0) on = true;                    // all 64 lanes
    else if (MODE == 1) on = (lane & 7) == 3;    // one lane of every group of 8 (what the turns do)
    else if (MODE == 2) on = lane < 8;           // 8 lanes inside one 16-lane row
    else if (MODE == 3) on = lane < 16;          // one full row
    else if (MODE == 4) on = lane < 32;          // half a wave
    else on = (lane & 1) == 0;                   // every other lane
    double a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    unsigned long long t0 = 0, t1 = 0;
    if (on) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a0 = fma(a0, c, 1.0); a1 = fma(a1, c, 1.0); a2 = fma(a2, c, 1.0); a3 = fma(a3, c, 1.0);
                a4 = fma(a4, c, 1.0); a5 = fma(a5, c, 1.0); a6 = fma(a6, c, 1.0); a7 = fma(a7, c, 1.0);
            }
        }
        asm volatile("s_nop 0" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
    }
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (lane == 0 && on) cyc[blockIdx.x] = t1 - t0;
    if (MODE == 1 && lane == 3) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int grid, int iters) {
    double *out; unsigned long long *cyc;
    hipMalloc(&out, grid * 64 * sizeof(double)); hipMalloc(&cyc, grid * sizeof(unsigned long long));
    hipMemset(cyc, 0, grid * sizeof(unsigned long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, cyc, iters, 0.999);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, cyc, iters, 0.999);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long *h = (unsigned long long *)malloc(grid * sizeof(unsigned long long));
    hipMemcpy(h, cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < grid; i++) mean += (double)h[i]; mean /= grid;
    printf("%-28s grid %5d: %.3f ms, %.2f shader cycles per f64 FMA wave-instruction\n", name, grid, ms, mean / ((double)iters * 64.0));
    free(h); hipFree(out); hipFree(cyc);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cu = p.multiProcessorCount;
    printf("device %s, %d CUs\n", p.name, cu);
    const int iters = 20000;
    for (int wps = 1; wps <= 2; wps++) {
        const int grid = cu * 4 * wps;
        printf("-- %d wave(s) per SIMD\n", wps);
        run<0>("all 64 lanes", grid, iters);
        run<1>("1 of 8 lanes (interleaved)", grid, iters);
        run<2>("lanes 0-7 (half a row)", grid, iters);
        run<3>("lanes 0-15 (one row)", grid, iters);
        run<4>("lanes 0-31", grid, iters);
        run<5>("even lanes", grid, iters);
    }
    return 0;
}
