// bb_contactmap.hip -- the ContactMap stage on the device: one (d, d) float64 matrix,
// d = n_bins + 1, that STAYS in HBM from the moment it is built until the solver has
// packed it (bb_cm_* handle), plus the two host-matrix entry points of round 1
// (bb_contactmap_scatter / bb_contactmap_normalize), now thin wrappers over the handle.
//
// What it replaces in the reference -- the lifecycle of `ContactMap.matrix`:
//   scatter     blueberry/datatypes.pyx:97-116   (zeros, then the triple loop)
//   normalize   blueberry/datatypes.pyx:161-171  (KR + observed/expected, nan_to_num)
//   filter      blueberry/datatypes.pyx:140-141  (column sums, boolean gather)
// fp64 throughout and bit-exact against golden vectors from the real class: the divisor is
// formed left to right as (KRnorm[j] * KRnorm[j+i]) * KRexpected[i] and applied with one
// IEEE division; a column's marginal is the sum of its elements in row order, which is
// the order numpy's `matrix.sum(axis=0)` adds a C-contiguous matrix in (checked bit for
// bit in tests/test_oracle.py).
#include <algorithm>

#include "bb_common.h"

struct bb_cm {
    int device = 0;
    int64_t d = 0;            // current edge (shrinks in filter)
    double *m = nullptr;      // (d, d) row-major, resident
    hipStream_t stream = nullptr;
};

namespace {

constexpr int kT = 32;  // tile edge of the normalise kernel

__device__ __forceinline__ double nan_to_num(double v) {
    // numpy.nan_to_num defaults: NaN -> 0, +/-inf -> +/-DBL_MAX
    if (v != v) return 0.0;
    if (v > 1.7976931348623157e308) return 1.7976931348623157e308;
    if (v < -1.7976931348623157e308) return -1.7976931348623157e308;
    return v;
}

// One workgroup per tile pair (TJ <= TK) of the (d,d) matrix, IN PLACE: computes the
// upper tile, writes it back, and writes its mirror through LDS so that every global
// access is row-contiguous.  A workgroup owns its tile pair alone, and the barrier
// separates its reads from its mirror writes, so no second matrix is needed.
// Traffic per upper pair: 8 B read + 16 B written.
__global__ __launch_bounds__(kT * 8) void normalize_kernel(double *m, int64_t d, int64_t n_bins,
                                                           const double *__restrict__ kr,
                                                           const double *__restrict__ krexp) {
    __shared__ double tile[kT][kT + 1];
    const int TJ = blockIdx.y, TK = blockIdx.x;
    if (TJ > TK) return;
    const int tx = threadIdx.x % kT, ty = threadIdx.x / kT;  // 32 x 8
#pragma unroll
    for (int q = 0; q < kT / 8; ++q) {
        const int rr = ty + 8 * q;
        const int64_t j = (int64_t)TJ * kT + rr, k = (int64_t)TK * kT + tx;
        double v = 0.0;
        if (j < d && k < d) {
            v = m[j * d + k];
            if (j < n_bins && k < n_bins && j <= k) v = v / (kr[j] * kr[k] * krexp[k - j]);
            if (j <= k || TJ != TK) m[j * d + k] = nan_to_num(v);
        }
        tile[rr][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kT / 8; ++q) {
        const int rr = ty + 8 * q;
        // mirrored element: row k' = TK*kT + rr, column j' = TJ*kT + tx  (k' > j' region)
        const int64_t kk = (int64_t)TK * kT + rr, jj = (int64_t)TJ * kT + tx;
        if (kk < d && jj < d && kk > jj) {
            // row / column n_bins is outside the reference's loop: such a cell keeps its own
            // value (nobody has written it: phase 1 leaves the lower cells alone)
            const double v = (kk < n_bins && jj < n_bins) ? tile[tx][rr]   // = normalised m[jj][kk]
                                                          : m[kk * d + jj];
            m[kk * d + jj] = nan_to_num(v);
        }
    }
}

// Scatter, pass 1: every cell a triple names records the LAST triple that names it
// (integer atomicMax of t + 1 on the cell's own 8 bytes: the matrix was zeroed first,
// and a count is only stored in pass 2).  No winner array the size of the matrix.
__global__ void scatter_mark_kernel(const double *__restrict__ tr, int64_t n, double resolution,
                                    int64_t d, double *m, int *__restrict__ bad) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int j = (int)(tr[t] / resolution), k = (int)(tr[n + t] / resolution);
    if (j < 0 || k < 0 || j >= d || k >= d) {
        atomicExch(bad, 1);
        return;
    }
    unsigned long long *cells = reinterpret_cast<unsigned long long *>(m);
    atomicMax(&cells[(int64_t)j * d + k], (unsigned long long)(t + 1));
    atomicMax(&cells[(int64_t)k * d + j], (unsigned long long)(t + 1));
}

// Pass 2: the winning triple stores its count (plain stores, as pyx:115-116).
__global__ void scatter_store_kernel(const double *__restrict__ tr, int64_t n, double resolution,
                                     int64_t d, double *m) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int j = (int)(tr[t] / resolution), k = (int)(tr[n + t] / resolution);
    if (j < 0 || k < 0 || j >= d || k >= d) return;
    const double c = tr[2 * n + t];
    unsigned long long *cells = reinterpret_cast<unsigned long long *>(m);
    // a cell still holding t + 1 is ours; j == k names one cell twice, harmlessly
    if (cells[(int64_t)j * d + k] == (unsigned long long)(t + 1)) m[(int64_t)j * d + k] = c;
    if (cells[(int64_t)k * d + j] == (unsigned long long)(t + 1)) m[(int64_t)k * d + j] = c;
}

// Column marginals: one thread per column, rows added IN ORDER (that is numpy's
// sum(axis=0) for a C-contiguous matrix, bit for bit); kUnroll loads are in flight
// before the dependent adds so that the sweep runs near HBM speed all the same.
constexpr int kSumUnroll = 16;
__global__ __launch_bounds__(128) void column_sums_kernel(const double *__restrict__ m, int64_t d,
                                                          double *__restrict__ sums) {
    const int64_t c = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (c >= d) return;
    const double *p = m + c;
    double acc = 0.0;
    int64_t i = 0;
    for (; i + kSumUnroll <= d; i += kSumUnroll) {
        double v[kSumUnroll];
#pragma unroll
        for (int q = 0; q < kSumUnroll; ++q) v[q] = __builtin_nontemporal_load(p + (i + q) * d);
#pragma unroll
        for (int q = 0; q < kSumUnroll; ++q) acc += v[q];
    }
    for (; i < d; ++i) acc += p[i * d];
    sums[c] = acc;
}

// keep[c] = sums[c] > threshold (NaN: false, as numpy), and the exclusive prefix sum of
// keep -> old_of_new[]; one workgroup, sequential over chunks of 1024 columns.
__global__ __launch_bounds__(1024) void keep_scan_kernel(const double *__restrict__ sums, int64_t d,
                                                         double threshold,
                                                         unsigned char *__restrict__ keep,
                                                         int *__restrict__ old_of_new,
                                                         int64_t *__restrict__ n_kept) {
    __shared__ int sh[1024];
    __shared__ int base;
    const int tid = threadIdx.x;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < d; c0 += 1024) {
        const int64_t c = c0 + tid;
        const int k = (c < d && sums[c] > threshold) ? 1 : 0;
        if (c < d) keep[c] = (unsigned char)k;
        sh[tid] = k;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int o = tid >= off ? sh[tid - off] : 0;
            __syncthreads();
            sh[tid] += o;
            __syncthreads();
        }
        if (k) old_of_new[base + sh[tid] - 1] = (int)c;
        __syncthreads();
        if (tid == 1023) base += sh[tid];
        __syncthreads();
    }
    if (tid == 0) *n_kept = base;
}

// out (dn, dn) = in[kept rows][:, kept columns]
__global__ __launch_bounds__(256) void gather_kernel(const double *__restrict__ in, int64_t d,
                                                     const int *__restrict__ old_of_new,
                                                     double *__restrict__ out, int64_t dn) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= dn) return;
    const int64_t oc = old_of_new[c];
    for (int64_t r = blockIdx.y; r < dn; r += gridDim.y)
        out[r * dn + c] = in[(int64_t)old_of_new[r] * d + oc];
}

int cm_check(const bb_cm *cm, const char *who) {
    if (!cm) return bb::fail(BB_ERR_INVALID, std::string(who) + ": contact map is NULL");
    return bb::enter_device(cm->device);
}

}  // namespace

extern "C" {

int bb_cm_create(bb_cm **out, int64_t d, int device) {
    BB_REQUIRE(out != nullptr, "bb_cm_create: out is NULL");
    *out = nullptr;
    BB_REQUIRE(d >= 1 && d <= (int64_t)2000000, "bb_cm_create: bad matrix edge");
    int rc = bb::use_device(device);
    if (rc != BB_OK) return rc;
    bb_cm *cm = new (std::nothrow) bb_cm();
    if (!cm) return bb::fail(BB_ERR_NOMEM, "bb_cm_create: out of host memory");
    cm->device = device;
    cm->d = d;
    hipError_t e = hipMalloc((void **)&cm->m, (size_t)d * d * sizeof(double));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&cm->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemsetAsync(cm->m, 0, (size_t)d * d * sizeof(double), cm->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(cm->stream);
    if (e != hipSuccess) {
        bb_cm_destroy(cm);
        return bb::fail(e == hipErrorOutOfMemory ? BB_ERR_NOMEM : BB_ERR_HIP,
                        std::string("bb_cm_create: ") + hipGetErrorString(e));
    }
    *out = cm;
    return BB_OK;
}

int bb_cm_destroy(bb_cm *cm) {
    if (!cm) return BB_OK;
    (void)hipSetDevice(cm->device);
    if (cm->stream) {
        (void)hipStreamSynchronize(cm->stream);
        (void)hipStreamDestroy(cm->stream);
    }
    (void)hipFree(cm->m);
    delete cm;
    (void)hipGetLastError();   // tear-down is best effort; its errors end here
    return BB_OK;
}

int bb_cm_dim(const bb_cm *cm, int64_t *d) {
    BB_REQUIRE(cm != nullptr && d != nullptr, "bb_cm_dim: NULL argument");
    *d = cm->d;
    return BB_OK;
}

int bb_cm_device_ptr(const bb_cm *cm, const double **dev_matrix, int64_t *d, int *device) {
    BB_REQUIRE(cm != nullptr, "bb_cm_device_ptr: contact map is NULL");
    if (dev_matrix) *dev_matrix = cm->m;
    if (d) *d = cm->d;
    if (device) *device = cm->device;
    return BB_OK;
}

int bb_cm_upload(bb_cm *cm, const double *matrix, int64_t ld) {
    BB_TRY(cm_check(cm, "bb_cm_upload"));
    BB_REQUIRE(matrix != nullptr && ld >= cm->d, "bb_cm_upload: bad host matrix");
    BB_HIP_CHECK(hipMemcpy2DAsync(cm->m, (size_t)cm->d * 8, matrix, (size_t)ld * 8,
                                  (size_t)cm->d * 8, (size_t)cm->d, hipMemcpyHostToDevice,
                                  cm->stream));
    BB_HIP_CHECK(hipStreamSynchronize(cm->stream));
    return BB_OK;
}

int bb_cm_download(bb_cm *cm, double *matrix, int64_t ld) {
    BB_TRY(cm_check(cm, "bb_cm_download"));
    BB_REQUIRE(matrix != nullptr && ld >= cm->d, "bb_cm_download: bad host matrix");
    BB_HIP_CHECK(hipMemcpy2DAsync(matrix, (size_t)ld * 8, cm->m, (size_t)cm->d * 8,
                                  (size_t)cm->d * 8, (size_t)cm->d, hipMemcpyDeviceToHost,
                                  cm->stream));
    BB_HIP_CHECK(hipStreamSynchronize(cm->stream));
    return BB_OK;
}

int bb_cm_scatter(bb_cm *cm, const double *triples, int64_t n, int32_t resolution) {
    BB_TRY(cm_check(cm, "bb_cm_scatter"));
    BB_REQUIRE(n >= 0 && (triples != nullptr || n == 0), "bb_cm_scatter: bad triples");
    BB_REQUIRE(n < (int64_t)0x7fffffff, "bb_cm_scatter: too many triples");
    BB_REQUIRE(resolution != 0, "bb_cm_scatter: resolution is 0");
    const int64_t d = cm->d;
    bb::DevBuf tr, bad;
    hipError_t e = tr.alloc((size_t)n * 3 * sizeof(double));
    if (e == hipSuccess) e = bad.alloc(sizeof(int));
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("bb_cm_scatter: ") + hipGetErrorString(e));
    hipStream_t st = cm->stream;
    int host_bad = 0;
    if (n > 0) e = hipMemcpyAsync(tr.p, triples, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(cm->m, 0, (size_t)d * d * sizeof(double), st);  // pyx:99
    if (e == hipSuccess) e = hipMemsetAsync(bad.p, 0, sizeof(int), st);
    if (e == hipSuccess && n > 0) {
        const unsigned grid = (unsigned)((n + 255) / 256);
        e = bb::launch(scatter_mark_kernel, dim3(grid), dim3(256), 0, st, (const double *)tr.p, n,
                       (double)resolution, d, cm->m, (int *)bad.p);
        if (e == hipSuccess)
            e = bb::launch(scatter_store_kernel, dim3(grid), dim3(256), 0, st,
                           (const double *)tr.p, n, (double)resolution, d, cm->m);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(&host_bad, bad.p, sizeof(int), hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_cm_scatter: ") + hipGetErrorString(e));
    if (host_bad) {
        // marks of valid triples may be left in the matrix: clear it, the map is unusable
        (void)hipMemset(cm->m, 0, (size_t)d * d * sizeof(double));
        return bb::fail(BB_ERR_INVALID,
                        "bb_cm_scatter: a position maps to a bin outside [0, n_bins]");
    }
    return BB_OK;
}

int bb_cm_normalize(bb_cm *cm, int64_t n_bins, const double *KRnorm, const double *KRexpected) {
    BB_TRY(cm_check(cm, "bb_cm_normalize"));
    BB_REQUIRE(KRnorm != nullptr && KRexpected != nullptr, "bb_cm_normalize: NULL argument");
    BB_REQUIRE(n_bins >= 0 && n_bins + 1 == cm->d,
               "bb_cm_normalize: the matrix edge is not n_bins + 1 (filtered already?)");
    const int64_t d = cm->d;
    bb::DevBuf kr, ke;
    hipError_t e = kr.alloc((size_t)n_bins * sizeof(double));
    if (e == hipSuccess) e = ke.alloc((size_t)n_bins * sizeof(double));
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("bb_cm_normalize: ") + hipGetErrorString(e));
    hipStream_t st = cm->stream;
    if (n_bins > 0) {
        e = hipMemcpyAsync(kr.p, KRnorm, (size_t)n_bins * sizeof(double), hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(ke.p, KRexpected, (size_t)n_bins * sizeof(double), hipMemcpyHostToDevice, st);
    }
    if (e == hipSuccess) {
        const unsigned nt = (unsigned)((d + kT - 1) / kT);
        e = bb::launch(normalize_kernel, dim3(nt, nt), dim3(kT * 8), 0, st, cm->m, d, n_bins,
                       (const double *)kr.p, (const double *)ke.p);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_cm_normalize: ") + hipGetErrorString(e));
    return BB_OK;
}

int bb_cm_marginals(bb_cm *cm, double *sums) {
    BB_TRY(cm_check(cm, "bb_cm_marginals"));
    BB_REQUIRE(sums != nullptr, "bb_cm_marginals: sums is NULL");
    bb::DevBuf s;
    hipError_t e = s.alloc((size_t)cm->d * 8);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("bb_cm_marginals: ") + hipGetErrorString(e));
    e = bb::launch(column_sums_kernel, dim3((unsigned)((cm->d + 127) / 128)), dim3(128), 0,
                   cm->stream, (const double *)cm->m, cm->d, (double *)s.p);
    if (e == hipSuccess) e = hipStreamSynchronize(cm->stream);
    if (e == hipSuccess) e = hipMemcpy(sums, s.p, (size_t)cm->d * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_cm_marginals: ") + hipGetErrorString(e));
    return BB_OK;
}

int bb_cm_filter(bb_cm *cm, double threshold, int64_t *d_new, uint8_t *keep_out) {
    BB_TRY(cm_check(cm, "bb_cm_filter"));
    const int64_t d = cm->d;
    bb::DevBuf sums, keep, idx, cnt;
    hipError_t e = sums.alloc((size_t)d * 8);
    if (e == hipSuccess) e = keep.alloc((size_t)d);
    if (e == hipSuccess) e = idx.alloc((size_t)d * 4);
    if (e == hipSuccess) e = cnt.alloc(8);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("bb_cm_filter: ") + hipGetErrorString(e));
    hipStream_t st = cm->stream;
    e = bb::launch(column_sums_kernel, dim3((unsigned)((d + 127) / 128)), dim3(128), 0, st,
                   (const double *)cm->m, d, (double *)sums.p);
    if (e == hipSuccess)
        e = bb::launch(keep_scan_kernel, dim3(1), dim3(1024), 0, st, (const double *)sums.p, d,
                       threshold, (unsigned char *)keep.p, (int *)idx.p, (int64_t *)cnt.p);
    int64_t dn = 0;
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(&dn, cnt.p, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && keep_out) e = hipMemcpy(keep_out, keep.p, (size_t)d, hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_cm_filter: ") + hipGetErrorString(e));
    double *out = nullptr;
    e = hipMalloc((void **)&out, (size_t)std::max<int64_t>(dn * dn, 1) * sizeof(double));
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("bb_cm_filter: ") + hipGetErrorString(e));
    if (dn > 0) {
        e = bb::launch(gather_kernel,
                       dim3((unsigned)((dn + 255) / 256), (unsigned)std::min<int64_t>(dn, 32768)), dim3(256), 0,
                       st, (const double *)cm->m, d, (const int *)idx.p, out, dn);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess) {
        (void)hipFree(out);
        return bb::fail(BB_ERR_HIP, std::string("bb_cm_filter: ") + hipGetErrorString(e));
    }
    (void)hipFree(cm->m);
    cm->m = out;
    cm->d = dn;
    if (d_new) *d_new = dn;
    return BB_OK;
}

// ---- round-1 entry points: the same kernels around a host matrix -------------------

int bb_contactmap_scatter(const double *triples, int64_t n, int32_t resolution, double *matrix,
                          int64_t d, int device) {
    BB_REQUIRE(matrix != nullptr && d >= 1, "bb_contactmap_scatter: bad matrix");
    bb_cm *cm = nullptr;
    int rc = bb_cm_create(&cm, d, device);
    if (rc == BB_OK) rc = bb_cm_scatter(cm, triples, n, resolution);
    if (rc == BB_OK) rc = bb_cm_download(cm, matrix, d);
    const std::string keep = bb_last_error();
    bb_cm_destroy(cm);
    if (rc != BB_OK) bb::set_error(keep);
    return rc;
}

int bb_contactmap_normalize(double *matrix, int64_t n_bins, const double *KRnorm,
                            const double *KRexpected, int device) {
    BB_REQUIRE(matrix != nullptr && KRnorm != nullptr && KRexpected != nullptr,
               "bb_contactmap_normalize: NULL argument");
    BB_REQUIRE(n_bins >= 0, "bb_contactmap_normalize: n_bins < 0");
    bb_cm *cm = nullptr;
    int rc = bb_cm_create(&cm, n_bins + 1, device);
    if (rc == BB_OK) rc = bb_cm_upload(cm, matrix, n_bins + 1);
    if (rc == BB_OK) rc = bb_cm_normalize(cm, n_bins, KRnorm, KRexpected);
    if (rc == BB_OK) rc = bb_cm_download(cm, matrix, n_bins + 1);
    const std::string keep = bb_last_error();
    bb_cm_destroy(cm);
    if (rc != BB_OK) bb::set_error(keep);
    return rc;
}

}  // extern "C"
