// Microbenchmark 2: cost of f64 / f32 VALU wave-instructions as a function of the EXEC mask (runtime mask).
// Build: hipcc -O3 --offload-arch=gfx950 masked_fma2.hip -o ../../build/masked_fma2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int OP>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *cyc, int iters, double c, unsigned long long mask) {
    const int lane = threadIdx.x;
    const bool on = (mask >> lane) & 1ull;
    double a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    float f0 = lane, f1 = lane + 1, f2 = lane + 2, f3 = lane + 3, f4 = lane + 4, f5 = lane + 5, f6 = lane + 6, f7 = lane + 7;
    const float cf = (float)c;
    unsigned long long t0 = 0, t1 = 0;
    if (on) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (OP == 0) {
                    a0 = fma(a0, c, 1.0); a1 = fma(a1, c, 1.0); a2 = fma(a2, c, 1.0); a3 = fma(a3, c, 1.0);
                    a4 = fma(a4, c, 1.0); a5 = fma(a5, c, 1.0); a6 = fma(a6, c, 1.0); a7 = fma(a7, c, 1.0);
                } else if (OP == 1) {
                    a0 = a0 * c; a1 = a1 * c; a2 = a2 * c; a3 = a3 * c; a4 = a4 * c; a5 = a5 * c; a6 = a6 * c; a7 = a7 * c;
                } else if (OP == 2) {
                    a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; a4 = a4 + c; a5 = a5 + c; a6 = a6 + c; a7 = a7 + c;
                } else {
                    f0 = fmaf(f0, cf, 1.0f); f1 = fmaf(f1, cf, 1.0f); f2 = fmaf(f2, cf, 1.0f); f3 = fmaf(f3, cf, 1.0f);
                    f4 = fmaf(f4, cf, 1.0f); f5 = fmaf(f5, cf, 1.0f); f6 = fmaf(f6, cf, 1.0f); f7 = fmaf(f7, cf, 1.0f);
                }
            }
        }
        asm volatile("s_nop 0" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
        if (lane == __ffsll((long long)mask) - 1) cyc[blockIdx.x] = t1 - t0;
    }
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
}

template <int OP>
void run(const char *name, int grid, int iters, unsigned long long mask) {
    double *out; unsigned long long *cyc;
    (void)hipMalloc(&out, grid * 64 * sizeof(double)); (void)hipMalloc(&cyc, grid * sizeof(unsigned long long));
    (void)hipMemset(cyc, 0, grid * sizeof(unsigned long long));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, out, cyc, iters, 0.999, mask);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, out, cyc, iters, 0.999, mask);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long *h = (unsigned long long *)malloc(grid * sizeof(unsigned long long));
    (void)hipMemcpy(h, cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < grid; i++) mean += (double)h[i]; mean /= grid;
    printf("%-10s mask %016llx (%2d lanes): %7.3f ms, %6.2f cycles/instr\n", name, mask, __builtin_popcountll(mask), ms, mean / ((double)iters * 64.0));
    free(h); (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cu = p.multiProcessorCount;
    const int iters = 5000;
    const unsigned long long masks[] = {
        0x1FFull, 0x3FFull, 0x7FFull, 0x0101010101010103ull, 0x0101010101010107ull, 0x010101010101010Full, 0x0101010101010181ull, 0x0101010101018181ull, 0x0101010181818181ull,
        ~0ull, 0x1ull, 0x3ull, 0xFull, 0xFFull, 0xFFFull, 0x7FFFull, 0xFFFFull, 0x1FFFFull, 0xFFFFFFull, 0xFFFFFFFFull,
        0x0101010101010101ull,   // 1 of 8
        0x0303030303030303ull,   // 2 of 8 adjacent
        0x1111111111111111ull,   // 1 of 4
        0x8181818181818181ull,   // 2 of 8 (lanes 0 and 7 of each group)
        0x0707070707070707ull,   // 3 of 8
        0x0F0F0F0F0F0F0F0Full,   // 4 of 8
        0x00FF00FF00FF00FFull,   // low half of every row
        0x000000000000FFFFull, 0x00000000FFFF0000ull,
        0x000F000F000F000Full,   // 4 lanes of each row
        0x0001000100010001ull,   // 1 lane of each row
        0x0003000300030003ull,   // 2 lanes of each row
    };
    for (int wps = 1; wps <= 2; wps++) {
        const int grid = cu * 4 * wps;
        printf("-- %d wave(s) per SIMD\n", wps);
        for (unsigned long long m : masks) run<0>("fma_f64", grid, iters, m);
        if (wps == 1) {
            for (unsigned long long m : {~0ull, 0x0101010101010101ull, 0x0303030303030303ull, 0xFFFFull}) {
                run<1>("mul_f64", grid, iters, m); run<2>("add_f64", grid, iters, m); run<3>("fma_f32", grid, iters, m);
            }
        }
    }
    return 0;
}
