// graph_probe.hip -- does a HIP graph shorten a chain of small dependent kernels?
// Three kernels per "iteration" (like grad / reduce stage 1 / reduce stage 2 at N~1k),
// each a few microseconds of dependent work, 2000 iterations: plain stream launches
// versus one instantiated graph of 16 iterations replayed 125 times.
// Build: hipcc --offload-arch=gfx950 -O3 -o graph_probe tools/probes/graph_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void work(float *buf, int n, int reps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float v = buf[i];
        for (int r = 0; r < reps; ++r) v = v * 1.0001f + 0.5f;
        buf[(i + 1) % n] = v;   // a dependency between consecutive launches
    }
}

int main() {
    float *buf;
    const int n = 1 << 16;
    CK(hipMalloc(&buf, n * sizeof(float)));
    CK(hipMemset(buf, 0, n * sizeof(float)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto iteration = [&]() {
        hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, st, buf, n, 600);   // "grad"
        hipLaunchKernelGGL(work, dim3(9), dim3(256), 0, st, buf, n, 100);     // "reduce 1"
        hipLaunchKernelGGL(work, dim3(2), dim3(256), 0, st, buf, n, 100);     // "reduce 2"
    };
    const int iters = 2000, G = 16;
    for (int rep = 0; rep < 3; ++rep) {
        for (int k = 0; k < 50; ++k) iteration();
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < iters; ++k) iteration();
        CK(hipStreamSynchronize(st));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("stream launches: %.2f us per iteration\n", us / iters);
    }
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < G; ++k) iteration();
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
        for (int k = 0; k < 4; ++k) CK(hipGraphLaunch(exec, st));
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < iters / G; ++k) CK(hipGraphLaunch(exec, st));
        CK(hipStreamSynchronize(st));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("graph of %d iterations: %.2f us per iteration\n", G, us / iters);
    }
    return 0;
}
