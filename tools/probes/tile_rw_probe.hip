// What normalize's MEMORY PATTERN alone can reach: for every upper tile pair (TJ <= TK) of a
// (d, d) fp64 matrix, read the tile, write it back in place and write the same bytes into the
// mirror tile's place -- row segments of E doubles on all three streams, exactly normalize128's
// traffic (8 B read + 16 B written per upper pair) -- with NO transposition, no LDS and no
// barrier, E = 64 / 128 / 256, and 1 or 2 workgroups of 1024 threads per CU.  If this streams
// at the 5.2 TB/s of the contiguous 1R:2W mix (rw_probe), the tile kernel's 3.6 TB/s is its
// LDS phases; if it does not, it is the 1-KiB segments.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/tile_rw_probe tools/probes/tile_rw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void tile_of(unsigned b, int &TJ, int &TK) {
    int t = (int)((__builtin_sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((unsigned)t * (unsigned)(t + 1) / 2 > b) --t;
    while ((unsigned)(t + 1) * (unsigned)(t + 2) / 2 <= b) ++t;
    TK = t; TJ = (int)(b - (unsigned)t * (unsigned)(t + 1) / 2);
}
template <int E>
__global__ __launch_bounds__(1024) void k(double *m, long d, unsigned n_pairs) {
    constexpr int RPT = 1024 / E;          // rows covered per pass
    const int tx = threadIdx.x % E, ty = threadIdx.x / E;
    for (unsigned b = blockIdx.x; b < n_pairs; b += gridDim.x) {
        int TJ, TK; tile_of(b, TJ, TK);
        const long k0 = (long)TK * E + tx, j0 = (long)TJ * E;
        double v[E / RPT];
#pragma unroll
        for (int q = 0; q < E / RPT; ++q) {
            const long j = j0 + ty + RPT * q;
            v[q] = (j < d && k0 < d) ? __builtin_nontemporal_load(m + j * d + k0) : 0.0;
        }
#pragma unroll
        for (int q = 0; q < E / RPT; ++q) {
            const long j = j0 + ty + RPT * q;
            if (j < d && k0 < d) __builtin_nontemporal_store(v[q] * 1.0000001, m + j * d + k0);
        }
        if (TJ != TK) {
#pragma unroll
            for (int q = 0; q < E / RPT; ++q) {       // mirror tile's place: rows of TK, columns of TJ
                const long kk = (long)TK * E + ty + RPT * q, jj = j0 + tx;
                if (kk < d && jj < d) __builtin_nontemporal_store(v[q], m + kk * d + jj);
            }
        }
    }
}
template <int E>
void run(double *m, long d, int wgs_per_cu) {
    const long nt = (d + E - 1) / E;
    const unsigned n_pairs = (unsigned)(nt * (nt + 1) / 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<E>), dim3(256 * wgs_per_cu), dim3(1024), 0, 0, m, d, n_pairs);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double bytes = (double)d * (d + 1) / 2 * 24;
    printf("segments of %3d doubles (%4d B), %d workgroup(s) of 1024 per CU: %.3f ms  %.0f GB/s on 24 B per upper pair\n",
           E, E * 8, wgs_per_cu, ms, bytes / (ms * 1e-3) / 1e9);
}
int main() {
    const long d = 24927;
    double *m; hipMalloc(&m, (size_t)d * d * 8); hipMemset(m, 0, (size_t)d * d * 8); hipDeviceSynchronize();
    for (int w = 1; w <= 2; ++w) { run<64>(m, d, w); run<128>(m, d, w); run<256>(m, d, w); }
    return 0;
}
