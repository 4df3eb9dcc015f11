// bb_band.hip -- K1: count_band_regions on the GPU.
//
// Replaces the O(N^2/2) `nogil` double loop at reference
// blueberry/blueberry.pyx:77-91:
//     for i in range(n): for j in range(i):
//         if LOW <= regions[i] - regions[j] <= HIGH: t += 1
// Exact: fp64 subtract, two fp64 compares against the int thresholds promoted
// to double (pyx:82), integer sum.  The input is O(N), so this kernel is
// issue-bound, not HBM-bound; it is the structural template for the pair
// tiling (j-chunk staged in LDS N-body style, one thread per i, wavefront
// shuffle reduction, one integer atomic per workgroup).
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>

#include "bb_common.h"

namespace {

constexpr int kIB = 256;   // rows (i) per workgroup: one per thread
constexpr int kJC = 2048;  // columns (j) staged in LDS per workgroup: 16 KiB

__global__ __launch_bounds__(kIB) void band_count_kernel(const double *__restrict__ r, int64_t n,
                                                         double lo, double hi, int64_t i_begin,
                                                         int64_t i_end,
                                                         unsigned long long *__restrict__ out) {
    __shared__ double sh[kJC];
    const int tid = threadIdx.x;
    const int64_t i0 = i_begin + (int64_t)blockIdx.y * kIB;
    const int64_t j0 = (int64_t)blockIdx.x * kJC;
    const int64_t i_top = i0 + kIB < i_end ? i0 + kIB : i_end;  // one past the last row here
    // columns that can pair with some row of this block: j <= (i_top - 1) - 1
    int64_t jn = (i_top - 1) - j0;
    if (jn > kJC) jn = kJC;
    if (jn <= 0) return;  // workgroup-uniform: every j of this chunk >= every i of this block

    for (int k = tid; k < jn; k += kIB) sh[k] = r[j0 + k];
    __syncthreads();

    const int64_t i = i0 + tid;
    const bool row_ok = i < i_end;
    const double ri = row_ok ? r[i] : 0.0;
    unsigned int cnt = 0;  // per-thread hits (<= kJC)
    if (j0 + jn <= i0) {
        // chunk entirely below the diagonal for every row of the block: j < i holds
#pragma unroll 8
        for (int k = 0; k < (int)jn; ++k) {
            const double d = ri - sh[k];
            cnt += (row_ok && lo <= d && d <= hi) ? 1u : 0u;
        }
    } else {
        const int64_t lim = i - j0;  // j0 + k < i  <=>  k < lim
#pragma unroll 8
        for (int k = 0; k < (int)jn; ++k) {
            const double d = ri - sh[k];
            cnt += (row_ok && k < lim && lo <= d && d <= hi) ? 1u : 0u;
        }
    }
    // wavefront shuffle reduction, then one integer atomic per workgroup
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    __shared__ unsigned int wsum[kIB / 64];
    if ((tid & 63) == 0) wsum[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) {
        unsigned long long tot = 0;
        for (int w = 0; w < kIB / 64; ++w) tot += wsum[w];
        if (tot) atomicAdd(out, tot);
    }
}

// Fast path for SORTED input.  `regions` is sorted whenever it comes from a ContactMap or a
// FithicContactMap (numpy.union1d, reference datatypes.pyx:119-120, 315), and then row i's
// hits are one contiguous run of j: d_j = r[i] - r[j] does not increase with j (fp64
// subtraction is monotone), so  #{j < i : lo <= d_j <= hi} = (#j with d_j >= lo) -
// (#j with d_j > hi), two binary searches with the SAME fp64 subtract and compares as the
// double loop -- exact, O(N log N) instead of O(N^2): N=309,568 in microseconds where the
// brute-force kernel takes 5.8 ms.  The kernel checks what it relies on: out[1] becomes
// non-zero if any r[k] > r[k+1] or a value is not finite (NaN compares false), and the
// caller then runs band_count_kernel, which needs no order.
__global__ __launch_bounds__(kIB) void band_count_sorted_kernel(
    const double *__restrict__ r, int64_t n, double lo, double hi, int64_t i_begin, int64_t i_end,
    unsigned long long *__restrict__ out) {
    const int tid = threadIdx.x;
    const int64_t gid = (int64_t)blockIdx.x * kIB + tid, gsz = (int64_t)gridDim.x * kIB;
    bool bad = false;
    for (int64_t k = gid; k + 1 < n; k += gsz) bad |= !(r[k] <= r[k + 1]);
    if (gid == 0) bad |= !(r[0] > -1.7976931348623157e308 && r[n - 1] < 1.7976931348623157e308);
    const int64_t i = i_begin + gid;
    unsigned long long cnt = 0;
    if (i < i_end && i > 0) {
        const double ri = r[i];
        int64_t a = 0, len = i;            // a = first j in [0, i) with ri - r[j] <= hi
        while (len > 0) {
            const int64_t half = len >> 1;
            const bool in = (ri - r[a + half]) <= hi;
            a = in ? a : a + half + 1;
            len = in ? half : len - half - 1;
        }
        int64_t b = 0;                     // b = first j in [0, i) with not (ri - r[j] >= lo)
        len = i;
        while (len > 0) {
            const int64_t half = len >> 1;
            const bool ge = (ri - r[b + half]) >= lo;
            b = ge ? b + half + 1 : b;
            len = ge ? len - half - 1 : half;
        }
        cnt = b > a ? (unsigned long long)(b - a) : 0ull;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    __shared__ unsigned long long wsum[kIB / 64];
    __shared__ int wbad[kIB / 64];
    const bool any_bad = __ballot(bad) != 0;
    if ((tid & 63) == 0) { wsum[tid >> 6] = cnt; wbad[tid >> 6] = any_bad ? 1 : 0; }
    __syncthreads();
    if (tid == 0) {
        unsigned long long tot = 0;
        int b = 0;
        for (int w = 0; w < kIB / 64; ++w) { tot += wsum[w]; b |= wbad[w]; }
        if (tot) atomicAdd(out, tot);
        if (b) atomicOr(out + 1, 1ull);
    }
}

// Per-device scratch kept between calls (stream, device buffers): creating a
// stream and two allocations per call cost ~4 ms, ten times the reference's CPU
// time for a 1,000-bin chromosome.  Guarded by a mutex: calls on one device
// serialise, which is what one stream would do anyway.
struct BandCtx {
    std::mutex mu;
    hipStream_t stream = nullptr;
    double *d_r = nullptr;
    int64_t cap = 0;
    unsigned long long *d_out = nullptr;
    unsigned long long *h_out = nullptr;  // pinned
};

BandCtx *band_ctx(int device) {
    static std::mutex table_mu;
    static std::map<int, BandCtx *> table;
    std::lock_guard<std::mutex> lock(table_mu);
    auto it = table.find(device);
    if (it != table.end()) return it->second;
    BandCtx *c = new BandCtx();   // lives for the process: freed by the runtime at exit
    table[device] = c;
    return c;
}

}  // namespace

extern "C" {

int bb_band_count_rows(const double *regions, int64_t n, int32_t low, int32_t high,
                       int64_t i_begin, int64_t i_end, int device, int64_t *count) {
    BB_REQUIRE(count != nullptr, "bb_band_count: count is NULL");
    *count = 0;
    BB_REQUIRE(n >= 0, "bb_band_count: n < 0");
    BB_REQUIRE(n <= (int64_t)0x7fffffff, "bb_band_count: n exceeds the reference's C int range");
    BB_REQUIRE(regions != nullptr || n == 0, "bb_band_count: regions is NULL");
    int rc = bb::use_device(device);
    if (rc != BB_OK) return rc;
    if (i_begin < 0) i_begin = 0;
    if (i_end > n) i_end = n;
    if (n < 2 || i_begin >= i_end || i_end < 2) return BB_OK;  // no (i, j < i) pair

    BandCtx *c = band_ctx(device);
    std::lock_guard<std::mutex> lock(c->mu);
    hipError_t e = hipSuccess;
    if (!c->stream) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess && !c->d_out) e = hipMalloc((void **)&c->d_out, 2 * sizeof(unsigned long long));
    if (e == hipSuccess && !c->h_out)
        e = hipHostMalloc((void **)&c->h_out, 2 * sizeof(unsigned long long), hipHostMallocDefault);
    if (e == hipSuccess && c->cap < n) {
        (void)hipFree(c->d_r);
        c->d_r = nullptr;
        c->cap = 0;
        e = hipMalloc((void **)&c->d_r, (size_t)n * sizeof(double));
        if (e == hipSuccess) c->cap = n;
    }
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("bb_band_count: ") + hipGetErrorString(e));
    hipStream_t st = c->stream;
    e = hipMemcpyAsync(c->d_r, regions, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st);
    const int64_t rows = i_end - i_begin;
    // sorted input (every ContactMap's regions): two binary searches per row; the kernel
    // reports in out[1] if the order it relies on does not hold.  BB_BAND_SORTED=0: never.
    const char *env = getenv("BB_BAND_SORTED");
    bool brute = env && atoi(env) == 0;
    if (!brute) {
        if (e == hipSuccess) e = hipMemsetAsync(c->d_out, 0, 2 * sizeof(unsigned long long), st);
        if (e == hipSuccess) {
            const int64_t want = std::max<int64_t>(rows, std::min<int64_t>(n, 1 << 16));
            e = bb::launch(band_count_sorted_kernel, dim3((unsigned)((want + kIB - 1) / kIB)), dim3(kIB),
                           0, st, c->d_r, n, (double)low, (double)high, i_begin, i_end, c->d_out);
        }
        if (e == hipSuccess)
            e = hipMemcpyAsync(c->h_out, c->d_out, 2 * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess)
            return bb::fail(BB_ERR_HIP, std::string("bb_band_count: ") + hipGetErrorString(e));
        if (c->h_out[1] == 0) {
            *count = (int64_t)c->h_out[0];
            return BB_OK;
        }
        brute = true;                      // not sorted (or not finite): the double loop
    }
    if (e == hipSuccess) e = hipMemsetAsync(c->d_out, 0, sizeof(unsigned long long), st);
    if (e == hipSuccess) {
        const dim3 grid((unsigned)((i_end - 1 + kJC - 1) / kJC), (unsigned)((rows + kIB - 1) / kIB));
        if (grid.y > 65535u) {
            e = hipErrorInvalidValue;
        } else {
            e = bb::launch(band_count_kernel, grid, dim3(kIB), 0, st, c->d_r, n, (double)low,
                           (double)high, i_begin, i_end, c->d_out);
        }
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(c->h_out, c->d_out, sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_band_count: ") + hipGetErrorString(e));
    const unsigned long long host_out = *c->h_out;
    *count = (int64_t)host_out;
    return BB_OK;
}

int bb_band_count(const double *regions, int64_t n, int32_t low, int32_t high, int device,
                  int64_t *count) {
    return bb_band_count_rows(regions, n, low, high, 0, n, device, count);
}

}  // extern "C"
