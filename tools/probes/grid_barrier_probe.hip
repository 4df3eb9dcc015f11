// grid_barrier_probe.hip -- VERDICT r2 #9: is BASELINE config 2 (N = 963, fp64, the row-owner
// path: one wave group per bin reads ALL coordinates, writes its own) faster as K iterations
// inside ONE launch with a device-scope barrier between them than as K launches?
// A model of row_owner_kernel with the same shape of work: 964 workgroups of 256 threads, 4
// waves per row, every wave reads X (n x 3 doubles) and one row of an n x ld matrix, a
// reduction over the row, the row's owner writes 3 doubles into the other X buffer.
//   (a) one launch per iteration (what the product does)
//   (b) one launch, XCD-sharded sense-reversing barrier: agent-scope release, one atomic per
//       workgroup on its shard's counter, the last of a shard on the top counter, everybody
//       polls the generation word, agent-scope acquire
//   (c) the same launch with the barrier's fences but WITHOUT the wait (what the fences alone
//       cost; results are wrong, timing only)
// Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe tools/probes/grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kShards = 8;

struct Barrier {
    unsigned shard[kShards][32];   // one 128-byte line per shard
    unsigned top[32];
    unsigned gen[32];
};

__device__ __forceinline__ void body(const double *__restrict__ full, int ld, int n, const double *Xin,
                                     double *Xout, double lr) {
    __shared__ double red[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = blockIdx.x;
    double gx = 0, gy = 0, gz = 0;
    const double xi = Xin[3 * i], yi = Xin[3 * i + 1], zi = Xin[3 * i + 2];
    for (int j = wv * 64 + lane; j < ld; j += 256) {
        const double d = full[(long long)i * ld + j];
        const double dx = xi - Xin[3 * j], dy = yi - Xin[3 * j + 1], dz = zi - Xin[3 * j + 2];
        const double d2 = dx * dx + dy * dy + dz * dz + 1e-30;
        const double rinv = rsqrt(d2), res = d > 0 ? d2 * rinv - d : 0.0, c = res * rinv;
        gx += c * dx; gy += c * dy; gz += c * dz;
    }
    for (int o = 32; o > 0; o >>= 1) {
        gx += __shfl_down(gx, o, 64); gy += __shfl_down(gy, o, 64); gz += __shfl_down(gz, o, 64);
    }
    if (lane == 0) { red[wv][0] = gx; red[wv][1] = gy; red[wv][2] = gz; }
    __syncthreads();
    if (tid == 0) {
        for (int p = 1; p < 4; ++p) { gx += red[p][0]; gy += red[p][1]; gz += red[p][2]; }
        Xout[3 * i] = xi - lr * 2 * gx; Xout[3 * i + 1] = yi - lr * 2 * gy; Xout[3 * i + 2] = zi - lr * 2 * gz;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void one_iteration(const double *full, int ld, int n, const double *Xin,
                                                     double *Xout, double lr) {
    if ((int)blockIdx.x < n) body(full, ld, n, Xin, Xout, lr);
}

template <bool WAIT>
__global__ __launch_bounds__(256) void persistent(const double *full, int ld, int n, double *Xa, double *Xb,
                                                  double lr, int iters, Barrier *bar) {
    const int per = (gridDim.x + kShards - 1) / kShards;
    const int shard = blockIdx.x % kShards;                  // blocks of one residue class share an XCD
    const int in_shard = ((int)gridDim.x - shard + kShards - 1) / kShards;
    (void)per;
    for (int k = 0; k < iters; ++k) {
        body(full, ld, n, (k & 1) ? Xb : Xa, (k & 1) ? Xa : Xb, lr);
        if (threadIdx.x == 0) {
            const unsigned want = (unsigned)(k + 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (atomicAdd(&bar->shard[shard][0], 1u) == (unsigned)in_shard - 1) {
                atomicExch(&bar->shard[shard][0], 0u);
                if (atomicAdd(&bar->top[0], 1u) == kShards - 1) {
                    atomicExch(&bar->top[0], 0u);
                    __hip_atomic_store(&bar->gen[0], want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (WAIT)
                while (__hip_atomic_load(&bar->gen[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want)
                    __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

// (d) no barrier at all: three coordinate buffers, and a coordinate word says by itself
// whether it has been written (all bits set = empty).  Whoever has read ALL of X_k knows
// that everybody is done with X_{k-1}: the owner of a row then empties its words of the
// buffer that held X_{k-1} (it will receive X_{k+2}), waits for those stores, and writes
// X_{k+1}.  Every read of X is an agent-scope load (past L2: the words cross XCDs).
__device__ __forceinline__ bool empty_word(double v) { return __double_as_longlong(v) == -1ll; }

__global__ __launch_bounds__(256) void persistent_inband(const double *full, int ld, int n, double *X3,
                                                         long long stride, double lr, int iters) {
    __shared__ double red[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = blockIdx.x;
    const double empty = __longlong_as_double(-1ll);
    for (int k = 0; k < iters; ++k) {
        const double *Xin = X3 + (long long)(k % 3) * stride;
        double *Xout = X3 + (long long)((k + 1) % 3) * stride, *Xold = X3 + (long long)((k + 2) % 3) * stride;
        double xi, yi, zi;
        do {
            xi = __hip_atomic_load(Xin + 3 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            yi = __hip_atomic_load(Xin + 3 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            zi = __hip_atomic_load(Xin + 3 * i + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while (__ballot(empty_word(xi) || empty_word(yi) || empty_word(zi)) != 0);
        double gx = 0, gy = 0, gz = 0;
        for (int j = wv * 64 + lane; j < ld; j += 256) {
            double xj = 0, yj = 0, zj = 0;
            if (j < n) {
                do {
                    xj = __hip_atomic_load(Xin + 3 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    yj = __hip_atomic_load(Xin + 3 * j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    zj = __hip_atomic_load(Xin + 3 * j + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while (empty_word(xj) || empty_word(yj) || empty_word(zj));
            }
            const double d = full[(long long)i * ld + j];
            const double dx = xi - xj, dy = yi - yj, dz = zi - zj;
            const double d2 = dx * dx + dy * dy + dz * dz + 1e-30;
            const double rinv = rsqrt(d2), res = d > 0 ? d2 * rinv - d : 0.0, c = res * rinv;
            gx += c * dx; gy += c * dy; gz += c * dz;
        }
        for (int o = 32; o > 0; o >>= 1) {
            gx += __shfl_down(gx, o, 64); gy += __shfl_down(gy, o, 64); gz += __shfl_down(gz, o, 64);
        }
        if (lane == 0) { red[wv][0] = gx; red[wv][1] = gy; red[wv][2] = gz; }
        __syncthreads();                       // every wave of the row has read all of X_k
        if (tid == 0) {
            for (int p = 1; p < 4; ++p) { gx += red[p][0]; gy += red[p][1]; gz += red[p][2]; }
            for (int c = 0; c < 3; ++c)
                __hip_atomic_store(Xold + 3 * i + c, empty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(Xout + 3 * i, xi - lr * 2 * gx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(Xout + 3 * i + 1, yi - lr * 2 * gy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(Xout + 3 * i + 2, zi - lr * 2 * gz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
}

// (e) the same protocol with X staged ONCE per workgroup: 1024 threads = 16 waves = 4 rows of
// 4 waves, one workgroup per CU; all threads fetch X_k (agent-scope loads, retried while a
// word is empty) into LDS, then the rows are swept from LDS.
__global__ __launch_bounds__(1024) void persistent_staged(const double *full, int ld, int n, double *X3,
                                                          long long stride, double lr, int iters) {
    extern __shared__ double xs[];                 // 3 * ld doubles
    __shared__ double red[16][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = blockIdx.x * 4 + (wv >> 2), part = wv & 3;
    const double empty = __longlong_as_double(-1ll);
    for (int k = 0; k < iters; ++k) {
        const double *Xin = X3 + (long long)(k % 3) * stride;
        double *Xout = X3 + (long long)((k + 1) % 3) * stride, *Xold = X3 + (long long)((k + 2) % 3) * stride;
        for (int w = tid; w < 3 * n; w += 1024) {
            double v;
            do v = __hip_atomic_load(Xin + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (empty_word(v));
            xs[w] = v;
        }
        for (int w = 3 * n + tid; w < 3 * ld; w += 1024) xs[w] = 0.0;
        __syncthreads();
        double gx = 0, gy = 0, gz = 0, xi = 0, yi = 0, zi = 0;
        if (i < n) {
            xi = xs[3 * i]; yi = xs[3 * i + 1]; zi = xs[3 * i + 2];
            for (int j = part * 64 + lane; j < ld; j += 256) {
                const double d = full[(long long)i * ld + j];
                const double dx = xi - xs[3 * j], dy = yi - xs[3 * j + 1], dz = zi - xs[3 * j + 2];
                const double d2 = dx * dx + dy * dy + dz * dz + 1e-30;
                const double rinv = rsqrt(d2), res = d > 0 ? d2 * rinv - d : 0.0, c = res * rinv;
                gx += c * dx; gy += c * dy; gz += c * dz;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            gx += __shfl_down(gx, o, 64); gy += __shfl_down(gy, o, 64); gz += __shfl_down(gz, o, 64);
        }
        if (lane == 0) { red[wv][0] = gx; red[wv][1] = gy; red[wv][2] = gz; }
        __syncthreads();
        if (i < n && part == 0 && lane == 0) {
            for (int p = 1; p < 4; ++p) { gx += red[wv + p][0]; gy += red[wv + p][1]; gz += red[wv + p][2]; }
            for (int c = 0; c < 3; ++c)
                __hip_atomic_store(Xold + 3 * i + c, empty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(Xout + 3 * i, xi - lr * 2 * gx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(Xout + 3 * i + 1, yi - lr * 2 * gy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(Xout + 3 * i + 2, zi - lr * 2 * gz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 963, ld = (n + 127) / 128 * 128, iters = 2000;
    double *full, *Xa, *Xb;
    Barrier *bar;
    CK(hipMalloc(&full, (size_t)n * ld * 8));
    CK(hipMalloc(&Xa, (size_t)ld * 3 * 8));
    CK(hipMalloc(&Xb, (size_t)ld * 3 * 8));
    CK(hipMalloc(&bar, sizeof(Barrier)));
    double *h = (double *)malloc((size_t)n * ld * 8), *hx = (double *)calloc((size_t)ld * 3, 8);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < ld; ++j) h[(size_t)i * ld + j] = (j < n && j != i) ? 1.0 + ((i * 31 + j * 17) % 97) * 0.01 : 0.0;
    for (int i = 0; i < 3 * n; ++i) hx[i] = ((i * 7919) % 1000) * 0.01;
    CK(hipMemcpy(full, h, (size_t)n * ld * 8, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int per_cu = 0, cus = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persistent<true>, 256, 0));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("n = %d: %d workgroups of 256 threads; %d fit the device at once (%d per CU x %d CUs)\n", n, n,
           per_cu * cus, per_cu, cus);
    if (per_cu * cus < n) { printf("the grid is not co-resident: no persistent form\n"); return 0; }
    const double lr = 1.0 / (2 * n);
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemcpy(Xa, hx, (size_t)ld * 3 * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(Xb, hx, (size_t)ld * 3 * 8, hipMemcpyHostToDevice));
        for (int k = 0; k < 100; ++k)
            hipLaunchKernelGGL(one_iteration, dim3(n), dim3(256), 0, st, full, ld, n, (k & 1) ? Xb : Xa, (k & 1) ? Xa : Xb, lr);
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < iters; ++k)
            hipLaunchKernelGGL(one_iteration, dim3(n), dim3(256), 0, st, full, ld, n, (k & 1) ? Xb : Xa, (k & 1) ? Xa : Xb, lr);
        CK(hipStreamSynchronize(st));
        const double a = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
        double xa[3];
        CK(hipMemcpy(xa, Xa, 24, hipMemcpyDeviceToHost));

        CK(hipMemcpy(Xa, hx, (size_t)ld * 3 * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(Xb, hx, (size_t)ld * 3 * 8, hipMemcpyHostToDevice));
        CK(hipMemset(bar, 0, sizeof(Barrier)));
        hipLaunchKernelGGL(persistent<true>, dim3(n), dim3(256), 0, st, full, ld, n, Xa, Xb, lr, 100, bar);
        CK(hipStreamSynchronize(st));
        CK(hipMemset(bar, 0, sizeof(Barrier)));
        t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(persistent<true>, dim3(n), dim3(256), 0, st, full, ld, n, Xa, Xb, lr, iters, bar);
        CK(hipStreamSynchronize(st));
        const double b = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
        double xb[3];
        CK(hipMemcpy(xb, Xa, 24, hipMemcpyDeviceToHost));

        CK(hipMemset(bar, 0, sizeof(Barrier)));
        t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(persistent<false>, dim3(n), dim3(256), 0, st, full, ld, n, Xa, Xb, lr, iters, bar);
        CK(hipStreamSynchronize(st));
        const double c = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
        // (d): X_0 in buffer 0, buffers 1 and 2 empty; 2100 iterations like the others
        double *X3;
        CK(hipMalloc(&X3, (size_t)3 * ld * 3 * 8));
        CK(hipMemset(X3, 0xff, (size_t)3 * ld * 3 * 8));
        CK(hipMemcpy(X3, hx, (size_t)ld * 3 * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(persistent_inband, dim3(n), dim3(256), 0, st, full, ld, n, X3, (long long)ld * 3, lr, 100);
        CK(hipStreamSynchronize(st));
        // the next launch goes on from X_100 in buffer 100 % 3 = 1: rotate so that it is buffer 0
        double *tmp = (double *)malloc((size_t)ld * 3 * 8);
        CK(hipMemcpy(tmp, X3 + (size_t)1 * ld * 3, (size_t)ld * 3 * 8, hipMemcpyDeviceToHost));
        CK(hipMemset(X3, 0xff, (size_t)3 * ld * 3 * 8));
        CK(hipMemcpy(X3, tmp, (size_t)ld * 3 * 8, hipMemcpyHostToDevice));
        t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(persistent_inband, dim3(n), dim3(256), 0, st, full, ld, n, X3, (long long)ld * 3, lr, iters);
        CK(hipStreamSynchronize(st));
        const double dd = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
        double xd[3];
        CK(hipMemcpy(xd, X3 + (size_t)(iters % 3) * ld * 3, 24, hipMemcpyDeviceToHost));
        // (e): the same from X_0 with X staged in LDS, one 1024-thread workgroup per 4 rows
        double ee = 0, xe[3] = {0, 0, 0};
        if ((n + 3) / 4 <= cus) {
            CK(hipFuncSetAttribute((const void *)persistent_staged, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * ld * 8));
            CK(hipMemset(X3, 0xff, (size_t)3 * ld * 3 * 8));
            CK(hipMemcpy(X3, hx, (size_t)ld * 3 * 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(persistent_staged, dim3((n + 3) / 4), dim3(1024), 3 * ld * 8, st, full, ld, n, X3, (long long)ld * 3, lr, 100);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(tmp, X3 + (size_t)1 * ld * 3, (size_t)ld * 3 * 8, hipMemcpyDeviceToHost));
            CK(hipMemset(X3, 0xff, (size_t)3 * ld * 3 * 8));
            CK(hipMemcpy(X3, tmp, (size_t)ld * 3 * 8, hipMemcpyHostToDevice));
            t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(persistent_staged, dim3((n + 3) / 4), dim3(1024), 3 * ld * 8, st, full, ld, n, X3, (long long)ld * 3, lr, iters);
            CK(hipStreamSynchronize(st));
            ee = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
            CK(hipMemcpy(xe, X3 + (size_t)(iters % 3) * ld * 3, 24, hipMemcpyDeviceToHost));
        }
        printf("rep %d: ... the same with X staged in LDS once per 1024-thread workgroup %.2f us (x0 %.6f)\n", rep, ee, xe[0]);
        free(tmp);
        CK(hipFree(X3));
        printf("rep %d: one launch per iteration %.2f us | one launch, grid barrier %.2f us | fences and atomics "
               "without the wait %.2f us | one launch, no barrier, words that say whether they are written %.2f us   "
               "(x0 after 2100 steps: %.6f / %.6f / %.6f)\n",
               rep, a, b, c, dd, xa[0], xb[0], xd[0]);
    }
    return 0;
}
