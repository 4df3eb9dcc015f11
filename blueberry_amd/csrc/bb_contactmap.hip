// bb_contactmap.hip -- A2/A3: ContactMap build (sparse triples -> dense
// symmetric matrix) and KR + observed/expected normalisation on the GPU.
//
// Replaces the two Cython loops at reference blueberry/datatypes.pyx:110-116
// (scatter) and :166-169 (+ nan_to_num at :171).  fp64 throughout, bit-exact
// against the reference's golden vectors: the divisor is formed left to right
// as (KRnorm[j] * KRnorm[j+i]) * KRexpected[i] and applied with one IEEE
// division, exactly as the C the reference compiles to.
#include "bb_common.h"

namespace {

constexpr int kT = 32;  // tile edge of the normalise kernel

__device__ __forceinline__ double nan_to_num(double v) {
    // numpy.nan_to_num defaults: NaN -> 0, +/-inf -> +/-DBL_MAX
    if (v != v) return 0.0;
    if (v > 1.7976931348623157e308) return 1.7976931348623157e308;
    if (v < -1.7976931348623157e308) return -1.7976931348623157e308;
    return v;
}

// One workgroup per tile pair (TJ <= TK) of the (d,d) matrix: computes the
// upper tile from `in`, writes it, and writes its mirror through LDS so that
// both global accesses are row-contiguous.
__global__ __launch_bounds__(kT * 8) void normalize_kernel(const double *__restrict__ in,
                                                           double *__restrict__ out, int64_t d,
                                                           int64_t n_bins,
                                                           const double *__restrict__ kr,
                                                           const double *__restrict__ krexp) {
    __shared__ double tile[kT][kT + 1];
    const int TJ = blockIdx.y, TK = blockIdx.x;
    if (TJ > TK) return;
    const int tx = threadIdx.x % kT, ty = threadIdx.x / kT;  // 32 x 8
    for (int rr = ty; rr < kT; rr += 8) {
        const int64_t j = (int64_t)TJ * kT + rr, k = (int64_t)TK * kT + tx;
        double v = 0.0;
        if (j < d && k < d) {
            v = in[j * d + k];
            if (j < n_bins && k < n_bins && j <= k) v = v / (kr[j] * kr[k] * krexp[k - j]);
            if (j <= k || TJ != TK) out[j * d + k] = nan_to_num(v);
        }
        tile[rr][tx] = v;
    }
    __syncthreads();
    for (int rr = ty; rr < kT; rr += 8) {
        // mirrored element: row k' = TK*kT + rr, column j' = TJ*kT + tx  (k' >= j' region)
        const int64_t kk = (int64_t)TK * kT + rr, jj = (int64_t)TJ * kT + tx;
        if (kk < d && jj < d && kk > jj) {
            double v;
            if (kk < n_bins && jj < n_bins)
                v = tile[tx][rr];           // = normalised m[jj][kk]
            else
                v = in[kk * d + jj];        // last row: untouched by the loop, only nan_to_num
            out[kk * d + jj] = nan_to_num(v);
        }
    }
}

// Pass 1 of the scatter: record, per matrix cell, the LAST triple that writes it.
__global__ void scatter_mark_kernel(const double *__restrict__ tr, int64_t n, double resolution,
                                    int64_t d, int *__restrict__ winner, int *__restrict__ bad) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int j = (int)(tr[t] / resolution), k = (int)(tr[n + t] / resolution);
    if (j < 0 || k < 0 || j >= d || k >= d) {
        atomicExch(bad, 1);
        return;
    }
    atomicMax(&winner[(int64_t)j * d + k], (int)t);
    atomicMax(&winner[(int64_t)k * d + j], (int)t);
}

// Pass 2: the winning triple stores its count (plain stores, as pyx:115-116).
__global__ void scatter_store_kernel(const double *__restrict__ tr, int64_t n, double resolution,
                                     int64_t d, const int *__restrict__ winner,
                                     double *__restrict__ m) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int j = (int)(tr[t] / resolution), k = (int)(tr[n + t] / resolution);
    if (j < 0 || k < 0 || j >= d || k >= d) return;
    const double c = tr[2 * n + t];
    if (winner[(int64_t)j * d + k] == (int)t) m[(int64_t)j * d + k] = c;
    if (winner[(int64_t)k * d + j] == (int)t) m[(int64_t)k * d + j] = c;
}

}  // namespace

extern "C" {

int bb_contactmap_scatter(const double *triples, int64_t n, int32_t resolution, double *matrix,
                          int64_t d, int device) {
    BB_REQUIRE(matrix != nullptr && d >= 1, "bb_contactmap_scatter: bad matrix");
    BB_REQUIRE(n >= 0 && (triples != nullptr || n == 0), "bb_contactmap_scatter: bad triples");
    BB_REQUIRE(n <= (int64_t)0x7fffffff, "bb_contactmap_scatter: too many triples");
    BB_REQUIRE(resolution != 0, "bb_contactmap_scatter: resolution is 0");
    int rc = bb::use_device(device);
    if (rc != BB_OK) return rc;
    bb::DevBuf tr, win, m, bad;
    hipStream_t st = nullptr;
    hipError_t e = tr.alloc((size_t)n * 3 * sizeof(double));
    if (e == hipSuccess) e = win.alloc((size_t)d * d * sizeof(int));
    if (e == hipSuccess) e = m.alloc((size_t)d * d * sizeof(double));
    if (e == hipSuccess) e = bad.alloc(sizeof(int));
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("bb_contactmap_scatter: ") + hipGetErrorString(e));
    int host_bad = 0;
    e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess && n > 0)
        e = hipMemcpyAsync(tr.p, triples, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(win.p, 0xFF, (size_t)d * d * sizeof(int), st);  // -1
    if (e == hipSuccess) e = hipMemsetAsync(m.p, 0, (size_t)d * d * sizeof(double), st);
    if (e == hipSuccess) e = hipMemsetAsync(bad.p, 0, sizeof(int), st);
    if (e == hipSuccess && n > 0) {
        const unsigned grid = (unsigned)((n + 255) / 256);
        e = bb::launch(scatter_mark_kernel, dim3(grid), dim3(256), 0, st, (const double *)tr.p, n,
                       (double)resolution, d, (int *)win.p, (int *)bad.p);
        if (e == hipSuccess)
            e = bb::launch(scatter_store_kernel, dim3(grid), dim3(256), 0, st,
                           (const double *)tr.p, n, (double)resolution, d, (const int *)win.p,
                           (double *)m.p);
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(matrix, m.p, (size_t)d * d * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(&host_bad, bad.p, sizeof(int), hipMemcpyDeviceToHost);
    if (st) hipStreamDestroy(st);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_contactmap_scatter: ") + hipGetErrorString(e));
    if (host_bad)
        return bb::fail(BB_ERR_INVALID,
                        "bb_contactmap_scatter: a position maps to a bin outside [0, n_bins]");
    return BB_OK;
}

int bb_contactmap_normalize(double *matrix, int64_t n_bins, const double *KRnorm,
                            const double *KRexpected, int device) {
    BB_REQUIRE(matrix != nullptr && KRnorm != nullptr && KRexpected != nullptr,
               "bb_contactmap_normalize: NULL argument");
    BB_REQUIRE(n_bins >= 0, "bb_contactmap_normalize: n_bins < 0");
    int rc = bb::use_device(device);
    if (rc != BB_OK) return rc;
    const int64_t d = n_bins + 1;
    bb::DevBuf in, out, kr, ke;
    hipStream_t st = nullptr;
    hipError_t e = in.alloc((size_t)d * d * sizeof(double));
    if (e == hipSuccess) e = out.alloc((size_t)d * d * sizeof(double));
    if (e == hipSuccess) e = kr.alloc((size_t)n_bins * sizeof(double));
    if (e == hipSuccess) e = ke.alloc((size_t)n_bins * sizeof(double));
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("bb_contactmap_normalize: ") + hipGetErrorString(e));
    e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipMemcpyAsync(in.p, matrix, (size_t)d * d * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && n_bins > 0)
        e = hipMemcpyAsync(kr.p, KRnorm, (size_t)n_bins * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && n_bins > 0)
        e = hipMemcpyAsync(ke.p, KRexpected, (size_t)n_bins * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        const unsigned nt = (unsigned)((d + kT - 1) / kT);
        e = bb::launch(normalize_kernel, dim3(nt, nt), dim3(kT * 8), 0, st, (const double *)in.p,
                       (double *)out.p, d, n_bins, (const double *)kr.p, (const double *)ke.p);
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(matrix, out.p, (size_t)d * d * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (st) hipStreamDestroy(st);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_contactmap_normalize: ") + hipGetErrorString(e));
    return BB_OK;
}

}  // extern "C"
