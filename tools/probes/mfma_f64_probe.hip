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
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    // the whole loop in one asm block, accumulators in a[0:127] throughout (left to the compiler,
    // the loop-carried accumulators live in VGPRs and are copied to and from the AGPRs around
    // every trip, which serialises on the MFMAs' completion)
    int n = iters;
    asm volatile(
        "s_mov_b32 s20, %0\n"
        "1:\n"
        "v_mfma_f64_16x16x4_f64 a[0:7], %1, %2, a[0:7]\n"
        "v_mfma_f64_16x16x4_f64 a[8:15], %1, %2, a[8:15]\n"
        "v_mfma_f64_16x16x4_f64 a[16:23], %1, %2, a[16:23]\n"
        "v_mfma_f64_16x16x4_f64 a[24:31], %1, %2, a[24:31]\n"
        "v_mfma_f64_16x16x4_f64 a[32:39], %1, %2, a[32:39]\n"
        "v_mfma_f64_16x16x4_f64 a[40:47], %1, %2, a[40:47]\n"
        "v_mfma_f64_16x16x4_f64 a[48:55], %1, %2, a[48:55]\n"
        "v_mfma_f64_16x16x4_f64 a[56:63], %1, %2, a[56:63]\n"
        "v_mfma_f64_16x16x4_f64 a[64:71], %1, %2, a[64:71]\n"
        "v_mfma_f64_16x16x4_f64 a[72:79], %1, %2, a[72:79]\n"
        "v_mfma_f64_16x16x4_f64 a[80:87], %1, %2, a[80:87]\n"
        "v_mfma_f64_16x16x4_f64 a[88:95], %1, %2, a[88:95]\n"
        "v_mfma_f64_16x16x4_f64 a[96:103], %1, %2, a[96:103]\n"
        "v_mfma_f64_16x16x4_f64 a[104:111], %1, %2, a[104:111]\n"
        "v_mfma_f64_16x16x4_f64 a[112:119], %1, %2, a[112:119]\n"
        "v_mfma_f64_16x16x4_f64 a[120:127], %1, %2, a[120:127]\n"
        "s_sub_u32 s20, s20, 1\n"
        "s_cmp_lg_u32 s20, 0\n"
        "s_cbranch_scc1 1b\n"
        "s_nop 15\n"
        :
        : "s"(n), "v"(a), "v"(b)
        : "s20", "scc", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13",
          "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27",
          "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41",
          "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55",
          "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69",
          "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83",
          "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97",
          "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109",
          "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121",
          "a122", "a123", "a124", "a125", "a126", "a127");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (a == 12345.678) out[0] = a;
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
