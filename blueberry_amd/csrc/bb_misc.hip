// bb_misc.hip -- the two remaining numeric Cython helpers of blueberry.pyx:
//   benjamini_hochberg   blueberry/blueberry.pyx:40-75   (sequential running max)
//   downsample           blueberry/blueberry.pyx:93-104  (5x5 max-pool, in place)
// Both are exact (max / min only, plus one multiply and one divide per element
// in the reference's operation order, NaN behaviour included), pinned by golden vectors.
#include "bb_common.h"

#include <map>
#include <mutex>

namespace {

// Per-device scratch kept between calls: a stream and ONE grow-only arena.  A fresh
// allocation is not what costs -- the first use of a freshly mapped block is, 0.17-0.35 s on
// this platform (tools/alloc_probe.py), and a stream costs 3-4 ms to make and destroy:
// small calls spent 3 ms around 0.1 ms of work.  Arenas above kKeepBytes are given
// back after the call.  Guarded by a mutex: calls on one device serialise, which is what
// one stream would do anyway.
constexpr size_t kKeepBytes = (size_t)1 << 30;
struct MiscCtx {
    std::mutex mu;
    hipStream_t stream = nullptr;
    char *arena = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        hipError_t e = hipSuccess;
        if (!stream) e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
        if (e == hipSuccess && cap < bytes) {
            (void)hipFree(arena);
            arena = nullptr;
            cap = 0;
            e = hipMalloc((void **)&arena, bytes);
            if (e == hipSuccess) cap = bytes;
        }
        return e;
    }
    void trim() {
        if (cap > kKeepBytes) {
            (void)hipFree(arena);
            arena = nullptr;
            cap = 0;
        }
    }
};
MiscCtx *misc_ctx(int device) {
    static std::mutex table_mu;
    static std::map<int, MiscCtx *> table;
    std::lock_guard<std::mutex> lock(table_mu);
    auto it = table.find(device);
    if (it != table.end()) return it->second;
    MiscCtx *c = new MiscCtx();   // lives for the process: freed by the runtime at exit
    table[device] = c;
    return c;
}
constexpr size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

constexpr int kScanBlock = 256;
constexpr int kItems = 4;  // per thread -> 1024 elements per workgroup

// The reference's loop (pyx:66-71) is
//     q = p[i] * n / (i+1);  q = min(q, 1);  q = max(q, prev);  prev = q
// and Cython lowers min(q, 1) to `1 < q ? 1 : q` and max(q, prev) to `prev > q ? prev : q`.
// For ordinary numbers that is a running maximum; a NaN p-value comes out as NaN AND,
// because `NaN > q` is false, restarts the maximum at the next element (golden cases
// bh_*_4..7).  So the scan runs over pairs (v, cut): v = the running value since the last
// NaN, cut = "a NaN lies in this range, nothing from before it gets through".
struct BhRun {
    double v;
    int cut;
};
__device__ __forceinline__ double ref_max(double prev, double q) { return prev > q ? prev : q; }
// `later` follows `earlier` in the sequence
__device__ __forceinline__ BhRun bh_join(BhRun earlier, BhRun later) {
    BhRun r;
    r.v = later.cut ? later.v : ref_max(earlier.v, later.v);
    r.cut = earlier.cut | later.cut;
    return r;
}
__device__ __forceinline__ BhRun bh_identity() {
    BhRun r;
    r.v = -__builtin_huge_val();  // -inf > q is false for every q: lets q through
    r.cut = 0;
    return r;
}

// inclusive scan of one BhRun per thread over the workgroup (Hillis-Steele)
__device__ __forceinline__ BhRun bh_block_scan(BhRun mine, double *shv, int *shc, int tid) {
    shv[tid] = mine.v;
    shc[tid] = mine.cut;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {
        BhRun o = bh_identity();
        if (tid >= off) { o.v = shv[tid - off]; o.cut = shc[tid - off]; }
        __syncthreads();
        BhRun me;
        me.v = shv[tid];
        me.cut = shc[tid];
        me = bh_join(o, me);
        shv[tid] = me.v;
        shc[tid] = me.cut;
        __syncthreads();
    }
    BhRun r;
    r.v = shv[tid];
    r.cut = shc[tid];
    return r;
}

// phase 1: the scan inside each workgroup's 1024 elements; per workgroup its summary and
// the index of its first NaN (what came before the workgroup still reaches up to there)
__global__ __launch_bounds__(kScanBlock) void bh_local_kernel(const double *__restrict__ p,
                                                              int64_t d, double n,
                                                              double *__restrict__ q,
                                                              double *__restrict__ blockv,
                                                              int *__restrict__ blockcut,
                                                              int *__restrict__ firstcut) {
    __shared__ double shv[kScanBlock];
    __shared__ int shc[kScanBlock];
    __shared__ int first;
    const int tid = threadIdx.x;
    if (tid == 0) first = kScanBlock * kItems;
    __syncthreads();
    const int64_t base = ((int64_t)blockIdx.x * kScanBlock + tid) * kItems;
    BhRun v[kItems];
    BhRun run = bh_identity();
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int64_t i = base + k;
        if (i < d) {
            double t = p[i] * n / (double)(i + 1);
            t = 1.0 < t ? 1.0 : t;                   // min(q, 1) as the reference evaluates it
            BhRun e;
            e.v = t;
            e.cut = t != t;
            run = bh_join(run, e);
            if (e.cut) atomicMin(&first, tid * kItems + k);
        }
        v[k] = run;
    }
    const BhRun incl = bh_block_scan(run, shv, shc, tid);
    BhRun before = bh_identity();
    if (tid > 0) { before.v = shv[tid - 1]; before.cut = shc[tid - 1]; }
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int64_t i = base + k;
        if (i < d) q[i] = bh_join(before, v[k]).v;
    }
    if (tid == kScanBlock - 1) {
        blockv[blockIdx.x] = incl.v;
        blockcut[blockIdx.x] = incl.cut;
        firstcut[blockIdx.x] = first;
    }
}

// phase 2: exclusive scan of the workgroup summaries (one workgroup, sequential over
// chunks), seeded with the reference's prev_q_value = 0.0
__global__ __launch_bounds__(kScanBlock) void bh_blockscan_kernel(double *__restrict__ blockv,
                                                                  int *__restrict__ blockcut,
                                                                  int64_t nblocks) {
    __shared__ double shv[kScanBlock];
    __shared__ int shc[kScanBlock];
    __shared__ double carry_v;
    __shared__ int carry_c;
    const int tid = threadIdx.x;
    if (tid == 0) { carry_v = 0.0; carry_c = 0; }
    __syncthreads();
    for (int64_t c0 = 0; c0 < nblocks; c0 += kScanBlock) {
        const int64_t i = c0 + tid;
        BhRun mine = bh_identity();
        if (i < nblocks) { mine.v = blockv[i]; mine.cut = blockcut[i]; }
        const BhRun incl = bh_block_scan(mine, shv, shc, tid);
        BhRun c;
        c.v = carry_v;
        c.cut = carry_c;
        BhRun excl = bh_identity();
        if (tid > 0) { excl.v = shv[tid - 1]; excl.cut = shc[tid - 1]; }
        if (i < nblocks) blockv[i] = bh_join(c, excl).v;   // what reaches workgroup i from before
        __syncthreads();
        if (tid == kScanBlock - 1) {
            const BhRun nc = bh_join(c, incl);
            carry_v = nc.v;
            carry_c = nc.cut;
        }
        __syncthreads();
    }
}

// phase 3: fold what came before the workgroup into its elements up to its first NaN
__global__ __launch_bounds__(kScanBlock) void bh_apply_kernel(double *__restrict__ q, int64_t d,
                                                              const double *__restrict__ prefix,
                                                              const int *__restrict__ firstcut) {
    const double pre = prefix[blockIdx.x];
    const int first = firstcut[blockIdx.x];
    const int64_t base = ((int64_t)blockIdx.x * kScanBlock + threadIdx.x) * kItems;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int64_t i = base + k;
        if (i < d && (int)threadIdx.x * kItems + k < first) q[i] = ref_max(pre, q[i]);
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
    MiscCtx *c = misc_ctx(device);
    std::lock_guard<std::mutex> lock(c->mu);
    const size_t o_q = align256((size_t)d * 8), o_bm = o_q + align256((size_t)d * 8),
                 o_bc = o_bm + align256((size_t)nblocks * 8), o_fc = o_bc + align256((size_t)nblocks * 4),
                 total = o_fc + align256((size_t)nblocks * 4);
    hipError_t e = c->ensure(total);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("bb_benjamini_hochberg: ") + hipGetErrorString(e));
    hipStream_t st = c->stream;
    double *p = (double *)c->arena, *q = (double *)(c->arena + o_q), *bm = (double *)(c->arena + o_bm);
    int *bc = (int *)(c->arena + o_bc), *fc = (int *)(c->arena + o_fc);
    e = hipMemcpyAsync(p, p_values, (size_t)d * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        e = bb::launch(bh_local_kernel, dim3((unsigned)nblocks), dim3(kScanBlock), 0, st,
                       (const double *)p, d, (double)n, q, bm, bc, fc);
        if (e == hipSuccess)
            e = bb::launch(bh_blockscan_kernel, dim3(1), dim3(kScanBlock), 0, st, bm, bc, nblocks);
        if (e == hipSuccess)
            e = bb::launch(bh_apply_kernel, dim3((unsigned)nblocks), dim3(kScanBlock), 0, st, q, d,
                           (const double *)bm, (const int *)fc);
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(q_values, q, (size_t)d * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    c->trim();
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
    MiscCtx *c = misc_ctx(device);
    std::lock_guard<std::mutex> lock(c->mu);
    const size_t o_b = align256((size_t)n1 * n1 * 4), total = o_b + align256((size_t)n5 * n5 * 4);
    hipError_t e = c->ensure(total);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("bb_downsample: ") + hipGetErrorString(e));
    hipStream_t st = c->stream;
    float *a = (float *)c->arena, *b = (float *)(c->arena + o_b);
    e = hipMemcpyAsync(a, yp1, (size_t)n1 * n1 * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = hipMemcpyAsync(b, yp5i, (size_t)n5 * n5 * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        const unsigned g = (unsigned)((n5 - 1 + 15) / 16);
        e = bb::launch(downsample_kernel, dim3(g, g), dim3(256), 0, st, (const float *)a, n1, b, n5);
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(yp5i, b, (size_t)n5 * n5 * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    c->trim();
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_downsample: ") + hipGetErrorString(e));
    return BB_OK;
}

}  // extern "C"
