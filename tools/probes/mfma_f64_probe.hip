// What the fp64 matrix pipe of this box sustains, with nothing else in the way: every wave
// issues `iters` x 16 independent v_mfma_f64_16x16x4_f64 (the accumulator pattern of
// gram_kernel), operands in registers, no memory.  Prints TFLOP/s for 1 and 2 waves per
// SIMD and the shader clock measured inside the kernel (s_memtime over the 100 MHz
// s_memrealtime).  The roofline's "peak" (78.6 TFLOP/s, MI355X_MICROARCH.md) is at the
// boost clock; this is what the chip gives under the load itself.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_f64_probe tools/probes/mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_kernel(int iters, double *out, unsigned long long *clk) {
    f64x4 acc[16];
    for (int q = 0; q < 16; ++q) acc[q] = f64x4{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0.0;
    for (int q = 0; q < 16; ++q) s += acc[q].x + acc[q].y + acc[q].z + acc[q].w;
    if (s == 12345.678) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x < 4096) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    double *out; unsigned long long *clk, h[8192];
    hipMalloc(&out, 8); hipMalloc(&clk, sizeof(h));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        const int grid = 256 * wgs_per_cu, iters = 4000;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            mfma_kernel<<<grid, 256>>>(iters, out, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
            double mhz_min = 1e9, mhz_max = 0;
            for (int g = 0; g < grid; ++g) { double m = (double)h[2 * g] / ((double)h[2 * g + 1] / 100.0); if (m < mhz_min) mhz_min = m; if (m > mhz_max) mhz_max = m; }
            const double flops = (double)grid * 4 * iters * 16 * 2048.0;
            printf("waves/SIMD %d rep %d: %.3f ms  %.1f TFLOP/s  shader clock %.0f-%.0f MHz\n", wgs_per_cu, rep, ms,
                   flops / (ms * 1e-3) / 1e12, mhz_min, mhz_max);
        }
    }
    return 0;
}
