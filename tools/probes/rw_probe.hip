// What a read / write mix can reach on this box: read-only, write-only, copy (1 read : 1
// write) and 1 read : 2 writes (ContactMap.normalize's mix: the upper pair is read, then
// it and its mirror are written), 8 and 16 bytes per lane, over 4.97 GB (a chr1@10kb matrix).
// hipcc --offload-arch=gfx950 -O3 -o rw_probe tools/probes/rw_probe.hip && ./rw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <typename V, int MODE>   // 0 read, 1 write, 2 copy, 3 read + 2 writes
__global__ __launch_bounds__(256) void k(const V *__restrict__ a, V *__restrict__ b, V *__restrict__ c,
                                         size_t n, double *sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    V acc{};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += 4 * stride) {
        V v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t j = i + q * stride;
            if (MODE != 1) v[q] = j < n ? __builtin_nontemporal_load(a + j) : V{};
            else v[q] = V{};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t j = i + q * stride;
            if (MODE == 0) { if constexpr (sizeof(V) == 8) acc += v[q]; else { acc.x += v[q].x; acc.y += v[q].y; } }
            if (MODE >= 1 && j < n) b[j] = v[q];
            if (MODE == 3 && j < n) c[j] = v[q];
        }
    }
    if (MODE == 0) { double s; if constexpr (sizeof(V) == 8) s = acc; else s = acc.x + acc.y; if (s == 12345.678) *sink = s; }
}
typedef double d2 __attribute__((ext_vector_type(2)));
template <typename V, int MODE>
void run(const char *name, void *a, void *b, void *c, size_t bytes, double *sink, int moved) {
    const size_t n = bytes / sizeof(V);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k<V, MODE>), dim3(256 * 16), dim3(256), 0, 0, (const V *)a, (V *)b, (V *)c, n, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-34s %2zu B/lane  %.3f ms  %.0f GB/s moved (%d x %.2f GB)\n", name, sizeof(V), ms, moved * bytes / (ms * 1e-3) / 1e9, moved, bytes / 1e9);
}
int main() {
    const size_t bytes = (size_t)24927 * 24927 * 8 / 16 * 16;
    void *a, *b, *c; double *sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&sink, 8);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes); hipMemset(c, 0, bytes); hipDeviceSynchronize();
    run<double, 0>("read only", a, b, c, bytes, sink, 1);   run<d2, 0>("read only", a, b, c, bytes, sink, 1);
    run<double, 1>("write only", a, b, c, bytes, sink, 1);  run<d2, 1>("write only", a, b, c, bytes, sink, 1);
    run<double, 2>("copy (1 read : 1 write)", a, b, c, bytes, sink, 2); run<d2, 2>("copy (1 read : 1 write)", a, b, c, bytes, sink, 2);
    run<double, 3>("1 read : 2 writes (normalize's mix)", a, b, c, bytes, sink, 3); run<d2, 3>("1 read : 2 writes (normalize's mix)", a, b, c, bytes, sink, 3);
    return 0;
}
