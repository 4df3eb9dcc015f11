// bb_misc.hip -- the two remaining numeric Cython helpers of blueberry.pyx:
//   benjamini_hochberg   blueberry/blueberry.pyx:40-75   (sequential running max)
//   downsample           blueberry/blueberry.pyx:93-104  (5x5 max-pool, in place)
// Both are exact (max / min only, plus one multiply and one divide per element
// in the reference's operation order), pinned by golden vectors.
#include "bb_common.h"

namespace {

constexpr int kScanBlock = 256;
constexpr int kItems = 4;  // per thread -> 1024 elements per workgroup

// q_i = max_{k<=i} min(p_k * n / (k+1), 1): phase 1 = per-workgroup inclusive max-scan
__global__ __launch_bounds__(kScanBlock) void bh_local_kernel(const double *__restrict__ p,
                                                              int64_t d, double n,
                                                              double *__restrict__ q,
                                                              double *__restrict__ blockmax) {
    __shared__ double sh[kScanBlock];
    const int tid = threadIdx.x;
    const int64_t base = ((int64_t)blockIdx.x * kScanBlock + tid) * kItems;
    double v[kItems];
    double run = 0.0;  // the reference starts from prev_q_value = 0.0
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int64_t i = base + k;
        double t = 0.0;
        if (i < d) {
            t = p[i] * n / (double)(i + 1);
            t = t < 1.0 ? t : 1.0;
        }
        run = t > run ? t : run;
        v[k] = run;
    }
    sh[tid] = run;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {  // Hillis-Steele inclusive max-scan
        double o = tid >= off ? sh[tid - off] : 0.0;
        __syncthreads();
        sh[tid] = o > sh[tid] ? o : sh[tid];
        __syncthreads();
    }
    const double before = tid > 0 ? sh[tid - 1] : 0.0;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int64_t i = base + k;
        if (i < d) q[i] = before > v[k] ? before : v[k];
    }
    if (tid == kScanBlock - 1) blockmax[blockIdx.x] = sh[tid];
}

// phase 2: exclusive max-scan of the workgroup maxima (one workgroup, sequential over chunks)
__global__ __launch_bounds__(kScanBlock) void bh_blockscan_kernel(double *__restrict__ blockmax,
                                                                  int64_t nblocks) {
    __shared__ double sh[kScanBlock];
    __shared__ double carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0.0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < nblocks; c0 += kScanBlock) {
        const int64_t i = c0 + tid;
        const double mine = i < nblocks ? blockmax[i] : 0.0;
        sh[tid] = mine;
        __syncthreads();
        for (int off = 1; off < kScanBlock; off <<= 1) {
            double o = tid >= off ? sh[tid - off] : 0.0;
            __syncthreads();
            sh[tid] = o > sh[tid] ? o : sh[tid];
            __syncthreads();
        }
        const double c = carry;
        const double excl = tid > 0 ? sh[tid - 1] : 0.0;
        if (i < nblocks) blockmax[i] = c > excl ? c : excl;  // max over all earlier workgroups
        __syncthreads();
        if (tid == kScanBlock - 1) carry = c > sh[tid] ? c : sh[tid];
        __syncthreads();
    }
}

// phase 3: fold the prefix of earlier workgroups in
__global__ __launch_bounds__(kScanBlock) void bh_apply_kernel(double *__restrict__ q, int64_t d,
                                                              const double *__restrict__ prefix) {
    const double pre = prefix[blockIdx.x];
    const int64_t base = ((int64_t)blockIdx.x * kScanBlock + threadIdx.x) * kItems;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int64_t i = base + k;
        if (i < d) q[i] = pre > q[i] ? pre : q[i];
    }
}

// yp5i[i][j] = max(yp5i[i][j], max of the 5x5 block of yp1), i, j < n5 - 1
__global__ __launch_bounds__(256) void downsample_kernel(const float *__restrict__ yp1, int64_t n1,
                                                         float *__restrict__ yp5i, int64_t n5) {
    const int64_t j = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
    const int64_t i = (int64_t)blockIdx.y * 16 + (threadIdx.x >> 4);
    if (i >= n5 - 1 || j >= n5 - 1) return;
    float m = yp5i[i * n5 + j];
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b) {
            const float v = yp1[(i * 5 + a) * n1 + (j * 5 + b)];
            m = v > m ? v : m;   // the reference's max(): NaN never replaces m
        }
    yp5i[i * n5 + j] = m;
}

}  // namespace

extern "C" {

int bb_benjamini_hochberg(const double *p_values, int64_t d, int64_t n, double *q_values,
                          int device) {
    BB_REQUIRE(d >= 0, "bb_benjamini_hochberg: d < 0");
    BB_REQUIRE(d == 0 || (p_values != nullptr && q_values != nullptr),
               "bb_benjamini_hochberg: NULL argument");
    int rc = bb::use_device(device);
    if (rc != BB_OK) return rc;
    if (d == 0) return BB_OK;
    const int64_t per_block = (int64_t)kScanBlock * kItems;
    const int64_t nblocks = (d + per_block - 1) / per_block;
    bb::DevBuf p, q, bm;
    hipStream_t st = nullptr;
    hipError_t e = p.alloc((size_t)d * 8);
    if (e == hipSuccess) e = q.alloc((size_t)d * 8);
    if (e == hipSuccess) e = bm.alloc((size_t)nblocks * 8);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("bb_benjamini_hochberg: ") + hipGetErrorString(e));
    e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipMemcpyAsync(p.p, p_values, (size_t)d * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(bh_local_kernel, dim3((unsigned)nblocks), dim3(kScanBlock), 0, st,
                           (const double *)p.p, d, (double)n, (double *)q.p, (double *)bm.p);
        hipLaunchKernelGGL(bh_blockscan_kernel, dim3(1), dim3(kScanBlock), 0, st, (double *)bm.p,
                           nblocks);
        hipLaunchKernelGGL(bh_apply_kernel, dim3((unsigned)nblocks), dim3(kScanBlock), 0, st,
                           (double *)q.p, d, (const double *)bm.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(q_values, q.p, (size_t)d * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (st) (void)hipStreamDestroy(st);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_benjamini_hochberg: ") + hipGetErrorString(e));
    return BB_OK;
}

int bb_downsample(const float *yp1, int64_t n1, float *yp5i, int64_t n5, int device) {
    BB_REQUIRE(yp1 != nullptr && yp5i != nullptr, "bb_downsample: NULL argument");
    BB_REQUIRE(n5 >= 1 && n1 >= 5 * (n5 - 1), "bb_downsample: yp1 smaller than 5 * (n5 - 1)");
    int rc = bb::use_device(device);
    if (rc != BB_OK) return rc;
    if (n5 < 2) return BB_OK;
    bb::DevBuf a, b;
    hipStream_t st = nullptr;
    hipError_t e = a.alloc((size_t)n1 * n1 * 4);
    if (e == hipSuccess) e = b.alloc((size_t)n5 * n5 * 4);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("bb_downsample: ") + hipGetErrorString(e));
    e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipMemcpyAsync(a.p, yp1, (size_t)n1 * n1 * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = hipMemcpyAsync(b.p, yp5i, (size_t)n5 * n5 * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        const unsigned g = (unsigned)((n5 - 1 + 15) / 16);
        hipLaunchKernelGGL(downsample_kernel, dim3(g, g), dim3(256), 0, st, (const float *)a.p, n1,
                           (float *)b.p, n5);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(yp5i, b.p, (size_t)n5 * n5 * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (st) (void)hipStreamDestroy(st);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_downsample: ") + hipGetErrorString(e));
    return BB_OK;
}

}  // extern "C"
