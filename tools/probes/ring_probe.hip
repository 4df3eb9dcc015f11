// ring_probe.hip -- how much does the TIMING of the refill loads matter?
// Every wave streams a contiguous chunk of 8-KiB units (like stress_grad_kernel) and
// "computes" for `sleep` x 64 cycles per row pair, in four ways:
//   mode 0  register window of 8 loads, refill AFTER the row's compute   (the product kernel's shape)
//   mode 1  register window of 8 loads, refill BEFORE the row's compute  (what it would like to be)
//   mode 2  LDS ring of D units filled by global_load_lds_dwordx4, hand-counted vmcnt
// Build: hipcc --offload-arch=gfx950 -O3 -o ring_probe tools/probes/ring_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 ntload(const f4 *p) { return __builtin_nontemporal_load(p); }

template <int MODE, int SLEEP>
__global__ __launch_bounds__(256) void sweep_regs(const f4 *__restrict__ units, int units_per_wave, float *sink) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const f4 *base = units + (long)w * units_per_wave * 512 + lane;
    f4 d[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) d[r] = ntload(base + r * 64);
    f4 acc = {0, 0, 0, 0};
    for (int u = 0; u < units_per_wave; ++u) {
        const f4 *next = base + (long)(u + 1 < units_per_wave ? u + 1 : u) * 512;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f4 a = d[2 * r], b = d[2 * r + 1];
            if (MODE == 1) {
                asm volatile("" : "+v"(a), "+v"(b));   // copies: the window registers are free now
                d[2 * r] = ntload(next + (2 * r) * 64);
                d[2 * r + 1] = ntload(next + (2 * r + 1) * 64);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc += a + b;
            if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 0) {
                d[2 * r] = ntload(next + (2 * r) * 64);
                d[2 * r + 1] = ntload(next + (2 * r + 1) * 64);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[w] = acc.x;
}

// LDS ring: slot s of this wave = 8 KiB; row pair r of a unit = 2 KiB (two 1-KiB wave loads).
template <int D, int SLEEP>
__global__ __launch_bounds__(256) void sweep_ring(const f4 *__restrict__ units, int units_per_wave, float *sink) {
    extern __shared__ f4 ring[];   // [4 waves][D][512]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int w = blockIdx.x * 4 + wv;
    const f4 *base = units + (long)w * units_per_wave * 512 + lane;
    f4 *mine = ring + wv * D * 512;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(mine));   // LDS byte address of this wave's ring
    auto dma = [&](const f4 *g, unsigned lds_byte) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt"
                     :: "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_byte)) : "memory");
    };
    // prologue: units 0 .. D-1
    for (int s = 0; s < D; ++s) {
        const f4 *g = base + (long)(s < units_per_wave ? s : units_per_wave - 1) * 512;
#pragma unroll
        for (int r = 0; r < 8; ++r) dma(g + r * 64, lds0 + (s * 512 + r * 64) * 16);
    }
    f4 acc = {0, 0, 0, 0};
    int slot = 0;
    for (int u = 0; u < units_per_wave; ++u) {
        // unit u's 8 loads are complete once at most 8*(D-1) younger ones are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * (D - 1)) : "memory");
        const int un = u + D < units_per_wave ? u + D : units_per_wave - 1;
        const f4 *g = base + (long)un * 512;
        const f4 *src = mine + slot * 512 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f4 a = src[(2 * r) * 64], b = src[(2 * r + 1) * 64];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // data in registers: the slot rows are free
            dma(g + (2 * r) * 64, lds0 + (slot * 512 + (2 * r) * 64) * 16);
            dma(g + (2 * r + 1) * 64, lds0 + (slot * 512 + (2 * r + 1) * 64) * 16);
            acc += a + b;
            if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
            __builtin_amdgcn_sched_barrier(0);
        }
        slot = slot + 1 == D ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[w] = acc.x;
}

int main(int argc, char **argv) {
    const int cus = 256;
    const long total_units = 610352;   // ~5.0 GB, the N=50k matrix
    float *sink;
    f4 *units;
    CK(hipMalloc(&sink, 1 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, int waves_per_cu, auto launch) {
        const int waves = cus * waves_per_cu, upw = (int)(total_units / waves);
        float best = 1e9, sum = 0;
        for (int it = 0; it < 6; ++it) {
            CK(hipEventRecord(e0));
            launch(waves / 4, upw);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it) { sum += ms; if (ms < best) best = ms; }
        }
        const double gb = (double)waves * upw * 8192 / 1e9;
        printf("%-34s waves/CU %2d: avg %.4f ms  %.0f GB/s (best %.0f)\n", name, waves_per_cu, sum / 5,
               gb / (sum / 5) * 1e3, gb / best * 1e3);
        fflush(stdout);
    };
    CK(hipMalloc(&units, (size_t)total_units * 8192));
    CK(hipMemset(units, 0, (size_t)total_units * 8192));
#define REGS(MODE, SL, WPC) run("regs mode " #MODE " sleep " #SL, WPC, [&](int grid, int upw) { \
        hipLaunchKernelGGL((sweep_regs<MODE, SL>), dim3(grid), dim3(256), 0, 0, units, upw, sink); })
#define RING(DD, SL, WPC) CK(hipFuncSetAttribute((const void *)sweep_ring<DD, SL>, \
        hipFuncAttributeMaxDynamicSharedMemorySize, 4 * DD * 8192)); \
    run("ring D=" #DD " sleep " #SL, WPC, [&](int grid, int upw) { \
        hipLaunchKernelGGL((sweep_ring<DD, SL>), dim3(grid), dim3(256), 4 * DD * 8192, 0, units, upw, sink); })
    REGS(0, 0, 8); REGS(0, 0, 4);
    REGS(0, 4, 8); REGS(1, 4, 8); REGS(0, 4, 4); REGS(1, 4, 4);
    REGS(0, 6, 8); REGS(1, 6, 8); REGS(0, 6, 4); REGS(1, 6, 4);
    RING(2, 0, 8); RING(2, 4, 8); RING(2, 6, 8);
    RING(3, 0, 4); RING(3, 4, 4); RING(3, 6, 4);
    RING(4, 0, 4); RING(4, 4, 4); RING(4, 6, 4);
    return 0;
}
