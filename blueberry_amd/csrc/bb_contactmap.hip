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
#include <math.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "bb_common.h"
#include <cstdlib>

struct bb_cm {
    int device = 0;
    int64_t d = 0;            // current edge (shrinks in filter)
    double *m = nullptr;      // (d, d) row-major, resident
    hipStream_t stream = nullptr;
    // grow-only scratch of the symmetric matrix-vector product (symv_upper_kernel): the work
    // list and the row / column partial sums; made by the first product, kept with the handle
    void *sv_buf = nullptr;
    size_t sv_bytes = 0;
    int64_t sv_d = -1;        // the edge the work list was built for
    int sv_items = 0;
    // scratch of bb_cm_correlation (centred rows + Gram matrix), kept between calls: the
    // first touch of a fresh matrix-sized allocation costs 0.2-0.35 s on this platform
};

namespace {

#ifndef BB_CM_TILE
#define BB_CM_TILE 32
#endif
constexpr int kT = BB_CM_TILE;  // tile edge of the normalise / finalise kernels (kT x 8 threads)

__device__ __forceinline__ double nan_to_num(double v) {
    // numpy.nan_to_num defaults: NaN -> 0, +/-inf -> +/-DBL_MAX
    if (v != v) return 0.0;
    if (v > 1.7976931348623157e308) return 1.7976931348623157e308;
    if (v < -1.7976931348623157e308) return -1.7976931348623157e308;
    return v;
}

// A workgroup barrier that orders LDS only.  __syncthreads() is also a release of the
// wave's GLOBAL stores: s_waitcnt vmcnt(0) in front of every s_barrier, i.e. the write
// acknowledgements of a whole tile (and the next tile's loads) twice per tile.  The tile
// pair belongs to this workgroup alone and no thread reads a global cell another thread
// of the launch writes, so nothing global needs ordering here.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// One workgroup per tile pair (TJ <= TK) of the (d,d) matrix, IN PLACE: computes the
// upper tile, writes it back, and writes its mirror through LDS so that every global
// access is row-contiguous.  A workgroup owns its tile pair alone, and the barrier
// separates its reads from its mirror writes, so no second matrix is needed.
// The grid is the triangle itself (nt(nt+1)/2 workgroups, none idle), and a thread has
// all of its loads in flight before its first store: `m` is read and written through
// the same pointer, so loads written after a store would wait for it.
// Traffic per upper pair: 8 B read + 16 B written.
__device__ __forceinline__ void tile_of(unsigned b, int &TJ, int &TK) {
    // b = TK (TK + 1) / 2 + TJ with TJ <= TK
    int t = (int)((__builtin_sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((unsigned)t * (unsigned)(t + 1) / 2 > b) --t;
    while ((unsigned)(t + 1) * (unsigned)(t + 2) / 2 <= b) ++t;
    TK = t;
    TJ = (int)(b - (unsigned)t * (unsigned)(t + 1) / 2);
}

__global__ __launch_bounds__(kT * 8) void normalize_kernel(double *m, int64_t d, int64_t n_bins,
                                                           const double *__restrict__ kr,
                                                           const double *__restrict__ krexp) {
    __shared__ double tile[kT][kT + 1];
    int TJ, TK;
    tile_of(blockIdx.x, TJ, TK);
    const int tx = threadIdx.x % kT, ty = threadIdx.x / kT;  // 32 x 8
    constexpr int Q = kT / 8;
    const int64_t k = (int64_t)TK * kT + tx;
    double v[Q], den[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t j = (int64_t)TJ * kT + ty + 8 * q;
        const bool in = j < d && k < d;
        const bool scaled = j < n_bins && k < n_bins && j <= k;
        v[q] = in ? m[j * d + k] : 0.0;
        den[q] = scaled ? kr[j] * kr[k] * krexp[k - j] : 1.0;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int rr = ty + 8 * q;
        const int64_t j = (int64_t)TJ * kT + rr;
        const bool scaled = j < n_bins && k < n_bins && j <= k;
        if (scaled) v[q] = v[q] / den[q];
        if (j < d && k < d && (j <= k || TJ != TK)) m[j * d + k] = nan_to_num(v[q]);
        tile[rr][tx] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int rr = ty + 8 * q;
        // mirrored element: row k' = TK*kT + rr, column j' = TJ*kT + tx  (k' > j' region)
        const int64_t kk = (int64_t)TK * kT + rr, jj = (int64_t)TJ * kT + tx;
        if (kk < d && jj < d && kk > jj) {
            // row / column n_bins is outside the reference's loop: such a cell keeps its own
            // value (nobody has written it: phase 1 leaves the lower cells alone)
            const double w = (kk < n_bins && jj < n_bins) ? tile[tx][rr]   // = normalised m[jj][kk]
                                                          : m[kk * d + jj];
            m[kk * d + jj] = nan_to_num(w);
        }
    }
}

// The same, 128 x 128 tiles through 132 KB of LDS, persistent workgroups (round 3).  With
// 32 x 32 tiles a tile row is 256 B of a 200-KB matrix row: on one of the two sides (the
// tile or its mirror) consecutive accesses of a workgroup open a new DRAM page each, and
// the kernel ran at 3.0-3.5 TB/s of its 24 B per upper pair (39 % of the 8 TB/s peak).
// Here every global access of a wave is 512 contiguous bytes of a 1-KiB row segment on
// BOTH sides; one workgroup of 1024 threads per CU walks a list of tile pairs, and the
// loads of its next pair are in flight while the mirror of the current one goes out
// (one workgroup per CU has nobody else to cover its load latency).  Same arithmetic,
// same order, same bits as normalize_kernel.  Dynamic LDS: tile[128][129] doubles.
#ifdef BB_CM_PLAIN_STORES
__device__ __forceinline__ void nt_store(double *p, double v) { *p = v; }
#else
__device__ __forceinline__ void nt_store(double *p, double v) { __builtin_nontemporal_store(v, p); }
#endif
// kNT = tile edge (128: one 1024-thread workgroup per CU; 64: 512 threads, up to four per CU)
template <int kNT>
__global__ __launch_bounds__(kNT * 8) void normalize128_kernel(double *m, int64_t d, int64_t n_bins,
                                                               const double *__restrict__ kr,
                                                               const double *__restrict__ krexp,
                                                               unsigned n_pairs) {
    constexpr int kNQ = kNT / 8;          // rows per thread: kNT columns x 8 threads
    extern __shared__ __attribute__((aligned(16))) double ntile[];   // [kNT][kNT + 1]
    const int tx = threadIdx.x % kNT, ty = threadIdx.x / kNT;
    double nv[kNQ];
    auto fetch = [&](unsigned b) __attribute__((always_inline)) {
        int TJ, TK;
        tile_of(b, TJ, TK);
        const int64_t k = (int64_t)TK * kNT + tx;
#pragma unroll
        for (int q = 0; q < kNQ; ++q) {
            const int64_t j = (int64_t)TJ * kNT + ty + 8 * q;
            nv[q] = (j < d && k < d) ? __builtin_nontemporal_load(m + j * d + k) : 0.0;
        }
    };
    unsigned b = blockIdx.x;
    if (b < n_pairs) fetch(b);
    while (b < n_pairs) {
        int TJ, TK;
        tile_of(b, TJ, TK);
        const int64_t k = (int64_t)TK * kNT + tx;
        const double krk = k < n_bins ? kr[k] : 1.0;
        double v[kNQ];
#pragma unroll
        for (int q = 0; q < kNQ; ++q) {
            const int rr = ty + 8 * q;
            const int64_t j = (int64_t)TJ * kNT + rr;
            const bool scaled = j < n_bins && k < n_bins && j <= k;
            v[q] = nv[q];
            if (scaled) v[q] = v[q] / (kr[j] * krk * krexp[k - j]);
            if (j < d && k < d && (j <= k || TJ != TK)) nt_store(m + j * d + k, nan_to_num(v[q]));
            ntile[rr * (kNT + 1) + tx] = v[q];
        }
        lds_barrier();
        const unsigned bn = b + gridDim.x;
        if (bn < n_pairs) fetch(bn);          // in flight while the mirror goes out
#pragma unroll
        for (int q = 0; q < kNQ; ++q) {
            const int rr = ty + 8 * q;
            // mirrored element: row k' = TK*kNT + rr, column j' = TJ*kNT + tx  (k' > j' region)
            const int64_t kk = (int64_t)TK * kNT + rr, jj = (int64_t)TJ * kNT + tx;
            if (kk < d && jj < d && kk > jj) {
                // row / column n_bins is outside the reference's loop: such a cell keeps its
                // own value (nobody has written it: phase 1 leaves the lower cells alone)
                const double w = (kk < n_bins && jj < n_bins) ? ntile[tx * (kNT + 1) + rr]
                                                              : m[kk * d + jj];
                nt_store(m + kk * d + jj, nan_to_num(w));
            }
        }
        lds_barrier();                        // the tile is free again
        b = bn;
    }
}

// Scatter, pass 1: every cell a triple names records the LAST triple that names it
// (integer atomicMax of t + 1 on the cell's own 8 bytes: the matrix was zeroed first,
// and a count is only stored in pass 2).  No winner array the size of the matrix.
// Element c of triple t is tr[t * st + c * sc]: (1, n) for the reference's column-major
// array (pyx:111-113), (3, 1) for C-ordered (n, 3) rows.  `present` (d bytes, may be NULL)
// gets a 1 for every bin a position falls in, `offgrid` a 1 if some position is not exactly
// bin * resolution -- what the caller needs to write down `regions` without sorting.
// numpy.nan_to_num with its defaults (pyx:102), applied to a triple's values as they are read:
// the 240 MB of a chr1@10kb file need no pass over them on the host
__device__ __forceinline__ double scatter_value(double v) {
    return v != v ? 0.0 : (v > 1.7976931348623157e308 ? 1.7976931348623157e308
                                                      : (v < -1.7976931348623157e308 ? -1.7976931348623157e308 : v));
}

__global__ void scatter_mark_kernel(const double *__restrict__ tr, int64_t n, int64_t st, int64_t sc,
                                    double resolution, int64_t d, double *m, int *__restrict__ bad,
                                    unsigned char *__restrict__ present, int *__restrict__ offgrid) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double pj = scatter_value(tr[t * st]), pk = scatter_value(tr[t * st + sc]);
    // (a position beyond the int range -- an infinity turned into 1.8e308 -- is out of range
    // whatever the cast would make of it)
    const double qj = pj / resolution, qk = pk / resolution;
    const bool wild = !(qj > -2147483648.0 && qj < 2147483648.0 && qk > -2147483648.0 && qk < 2147483648.0);
    const int j = wild ? -1 : (int)qj, k = wild ? -1 : (int)qk;
    if (j < 0 || k < 0 || j >= d || k >= d) {
        atomicExch(bad, 1);
        return;
    }
    if (present != nullptr) {
        present[j] = 1;
        present[k] = 1;
        if ((double)j * resolution != pj || (double)k * resolution != pk) *offgrid = 1;
    }
    unsigned long long *cells = reinterpret_cast<unsigned long long *>(m);
    atomicMax(&cells[(int64_t)j * d + k], (unsigned long long)(t + 1));
    atomicMax(&cells[(int64_t)k * d + j], (unsigned long long)(t + 1));
}

// Pass 2: the winning triple stores its count (plain stores, as pyx:115-116).
__global__ void scatter_store_kernel(const double *__restrict__ tr, int64_t n, int64_t st, int64_t sc,
                                     double resolution, int64_t d, double *m) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double qj = scatter_value(tr[t * st]) / resolution, qk = scatter_value(tr[t * st + sc]) / resolution;
    const bool wild = !(qj > -2147483648.0 && qj < 2147483648.0 && qk > -2147483648.0 && qk < 2147483648.0);
    const int j = wild ? -1 : (int)qj, k = wild ? -1 : (int)qk;
    if (j < 0 || k < 0 || j >= d || k >= d) return;
    const double c = scatter_value(tr[t * st + 2 * sc]);
    unsigned long long *cells = reinterpret_cast<unsigned long long *>(m);
    // a cell still holding t + 1 is ours; j == k names one cell twice, harmlessly
    if (cells[(int64_t)j * d + k] == (unsigned long long)(t + 1)) m[(int64_t)j * d + k] = c;
    if (cells[(int64_t)k * d + j] == (unsigned long long)(t + 1)) m[(int64_t)k * d + j] = c;
}

// Column marginals: one thread per column, rows added IN ORDER (that is numpy's
// sum(axis=0) for a C-contiguous matrix, bit for bit); kUnroll loads are in flight
// before the dependent adds so that the sweep runs near HBM speed all the same.
#ifndef BB_CM_SUM_UNROLL
#define BB_CM_SUM_UNROLL 64
#endif
constexpr int kSumUnroll = BB_CM_SUM_UNROLL;
#ifndef BB_CM_SUM_WG
#define BB_CM_SUM_WG 128
#endif
constexpr int kSumWG = BB_CM_SUM_WG;
__global__ __launch_bounds__(kSumWG) void column_sums_kernel(const double *__restrict__ m, int64_t d,
                                                          double *__restrict__ sums) {
    const int64_t c = (int64_t)blockIdx.x * kSumWG + threadIdx.x;
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

constexpr int64_t kFilterBounceElems = 8 << 20;   // 64 MiB of doubles between gather and copy-back

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

// tmp (r1 - r0, dn) = in[kept rows r0..r1)[:, kept columns]
__global__ __launch_bounds__(256) void gather_kernel(const double *__restrict__ in, int64_t d,
                                                     const int *__restrict__ old_of_new,
                                                     double *__restrict__ tmp, int64_t dn,
                                                     int64_t r0, int64_t r1) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= dn) return;
    const int64_t oc = old_of_new[c];
    for (int64_t r = r0 + blockIdx.y; r < r1; r += gridDim.y)
        tmp[(r - r0) * dn + c] = in[(int64_t)old_of_new[r] * d + oc];
}

// ---- eigenvector (datatypes.pyx:216-235): Lanczos on the resident matrix -----------
// y = M x.  One wave takes kSymvRows consecutive rows (a workgroup 4 waves), so x is
// loaded once per kSymvRows row loads, and reads them as 16-byte elements: a row starts on
// a 16-byte boundary or 8 bytes past one (odd d: every other row), so each row gets its
// own one-element head and its pairs are {x[2c+o], x[2c+o+1]} with o = 0 or 1.  kSymvUnroll
// wave loads of 1 KiB per row in flight; fixed summation order; HBM-bound, 8 B per element.
constexpr int kSymvRows = 4, kSymvUnroll = 4;
__global__ __launch_bounds__(256) void symv_kernel(const double *__restrict__ m, int64_t d,
                                                   const double *__restrict__ x,
                                                   double *__restrict__ y) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * kSymvRows;
    if (row0 >= d) return;
    const d2 *p2[kSymvRows];
    int off[kSymvRows];
    double acc[kSymvRows];
#pragma unroll
    for (int r = 0; r < kSymvRows; ++r) {
        const int64_t row = row0 + r < d ? row0 + r : d - 1;      // a ragged last group re-reads row d-1
        const double *p = m + row * d;
        off[r] = (int)((row * d) & 1);
        p2[r] = reinterpret_cast<const d2 *>(p + off[r]);
        acc[r] = (off[r] && lane == 0) ? p[0] * x[0] : 0.0;        // the head element
    }
    const int64_t n2 = (d - 1) / 2;      // pairs every row has after its head, whatever its o
    int64_t c = lane;
    for (; c + 64 * (kSymvUnroll - 1) < n2; c += 64 * kSymvUnroll) {
        d2 a[kSymvRows][kSymvUnroll];
        double x0[kSymvUnroll], x1[kSymvUnroll], x2[kSymvUnroll];
#pragma unroll
        for (int q = 0; q < kSymvUnroll; ++q) {
#pragma unroll
            for (int r = 0; r < kSymvRows; ++r)
                a[r][q] = __builtin_nontemporal_load(p2[r] + c + 64 * q);
            const int64_t e = 2 * (c + 64 * q);
            x0[q] = x[e]; x1[q] = x[e + 1]; x2[q] = x[e + 2];      // e + 2 <= 2 n2 <= d - 1
        }
#pragma unroll
        for (int q = 0; q < kSymvUnroll; ++q)
#pragma unroll
            for (int r = 0; r < kSymvRows; ++r) {
                acc[r] = fma(a[r][q].x, off[r] ? x1[q] : x0[q], acc[r]);
                acc[r] = fma(a[r][q].y, off[r] ? x2[q] : x1[q], acc[r]);
            }
    }
    for (; c < n2; c += 64) {
        const int64_t e = 2 * c;
#pragma unroll
        for (int r = 0; r < kSymvRows; ++r) {
            const d2 v = p2[r][c];
            acc[r] = fma(v.x, x[e + off[r]], acc[r]);
            acc[r] = fma(v.y, x[e + off[r] + 1], acc[r]);
        }
    }
    // what is left of a row behind its n2 pairs: elements off + 2 n2 .. d - 1 (at most 2)
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < kSymvRows; ++r) {
            const int64_t row = row0 + r < d ? row0 + r : d - 1;
            for (int64_t e = off[r] + 2 * n2; e < d; ++e) acc[r] = fma(m[row * d + e], x[e], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < kSymvRows; ++r) {
        double v = acc[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0 && row0 + r < d) y[row0 + r] = v;
    }
}

// ---- the same product from the UPPER triangle alone (round 3) ------------------------
// symv_kernel reads both triangles of a symmetric matrix: 8 B per element where 8 B per
// PAIR would do (4.25 TB/s, 53 % of peak, on its own accounting).  Here element (i, j),
// j >= i, is read once and serves both ends, y_i += m_ij x_j and y_j += m_ij x_i -- the
// pattern of the solver's sweep (kOpMatvec2).  The matrix is taken to be symmetric, as the
// reference's eigsh call takes it (datatypes.pyx:234) and as every ContactMap is built.
//   work item   64 rows (4 waves x 16) x up to 4096 columns of the upper triangle; the list
//               is cut by rows AND columns so that no item is long (a row block alone would
//               be 0.2 to 12.8 MB at d = 25k and the launch as slow as its longest)
//   row side    16 per-lane accumulators per wave, reduced across the lanes once per item
//               -> rowpart[segment][row]
//   column side a lane owns one column of a 64-column chunk; the 16 rows of the wave add
//               into one register; after 8 chunks the 4 waves' sums meet in LDS and leave
//               as one value per column -> colpart[row block][column] (1.5 % of the bytes read)
//   symv_reduce_kernel adds, per element of y, its row partials (<= d / 4096 + 1) and its
//               column partials (<= d / 64 + 1) in a fixed order, 8 slices in parallel.
// All loads are 8 bytes per lane, 512 contiguous bytes per wave: rows of an odd-d matrix
// start 8 bytes off every other time, and 16 loads of a wave are in flight per chunk.
constexpr int kSvRows = 64, kSvSeg = 4096, kSvGroup = 8;
// plain loads: the 512-byte segments of a wave are not line-aligned (odd d), neighbouring
// chunks share their end lines, and a non-temporal load does not leave them in L2 for the
// neighbour -- 481 against 505 us per product at d = 24,927 (-DBB_CM_SYMV_NT for the A/B)
#ifdef BB_CM_SYMV_NT
__device__ __forceinline__ double sv_load(const double *p) { return __builtin_nontemporal_load(p); }
#else
__device__ __forceinline__ double sv_load(const double *p) { return *p; }
#endif
__global__ __launch_bounds__(256, 2) void symv_upper_kernel(const double *__restrict__ m, int64_t d,
                                                            const double *__restrict__ x,
                                                            const int2 *__restrict__ items,
                                                            double *__restrict__ rowpart,
                                                            double *__restrict__ colpart) {
    __shared__ double meet[4][kSvGroup][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int2 it = items[blockIdx.x];
    const int64_t I = it.x, S = it.y;
    const int64_t row0 = I * kSvRows + wave * 16;
    const int64_t c_begin = std::max<int64_t>(I * kSvRows, S * kSvSeg);
    const int64_t c_end = std::min<int64_t>(d, (S + 1) * (int64_t)kSvSeg);
    double xr[16], racc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        xr[r] = row0 + r < d ? x[row0 + r] : 0.0;
        racc[r] = 0.0;
    }
    for (int64_t cg = c_begin; cg < c_end; cg += 64 * kSvGroup) {
        double cacc[kSvGroup];
#pragma unroll
        for (int g = 0; g < kSvGroup; ++g) {
            cacc[g] = 0.0;
            const int64_t c0 = cg + 64 * g;              // chunk start (uniform)
            if (c0 >= c_end) continue;
            const int64_t c = c0 + lane;
            const bool in_c = c < c_end;
            const double xc = in_c ? x[c] : 0.0;
            double a[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                a[r] = (in_c && row0 + r < d) ? sv_load(m + (row0 + r) * d + c) : 0.0;
            if (c0 < row0 + 16) {
                // the chunk crosses this wave's rows: below the diagonal nothing counts, on it
                // only the row side (the column side would count m_ii x_i twice)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    racc[r] = fma(c >= row0 + r ? a[r] : 0.0, xc, racc[r]);
                    cacc[g] = fma(c > row0 + r ? a[r] : 0.0, xr[r], cacc[g]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    racc[r] = fma(a[r], xc, racc[r]);
                    cacc[g] = fma(a[r], xr[r], cacc[g]);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < kSvGroup; ++g) meet[wave][g][lane] = cacc[g];
        lds_barrier();   // (LDS only: nobody reads the global cells stored here)
        for (int j = threadIdx.x; j < 64 * kSvGroup; j += 256) {
            const int g = j >> 6, l = j & 63;
            const int64_t c = cg + j;
            if (c < c_end)
                colpart[I * d + c] = ((meet[0][g][l] + meet[1][g][l]) + meet[2][g][l]) + meet[3][g][l];
        }
        lds_barrier();   // (LDS only: nobody reads the global cells stored here)
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        double v = racc[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0 && row0 + r < d) rowpart[S * d + row0 + r] = v;
    }
}

// y[c] = sum of c's row partials (segments c / 4096 ..) + its column partials (row blocks
// 0 .. c / 64), each list in order, cut into 8 slices that are added in slice order.
__global__ __launch_bounds__(1024) void symv_reduce_kernel(const double *__restrict__ rowpart,
                                                           const double *__restrict__ colpart,
                                                           int64_t d, int nseg, double *__restrict__ y) {
    __shared__ double meet[8][128];
    const int el = threadIdx.x & 127, sl = threadIdx.x >> 7;
    const int64_t c = (int64_t)blockIdx.x * 128 + el;
    double acc = 0.0;
    if (c < d) {
        const int64_t s0 = c / kSvSeg, nrow = nseg - s0, ncol = c / kSvRows + 1, n = nrow + ncol;
        const int64_t per = (n + 7) / 8, k0 = sl * per, k1 = std::min<int64_t>(n, k0 + per);
        for (int64_t k = k0; k < k1; k += 16) {
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t kk = k + q;
                v[q] = kk >= k1 ? 0.0
                                : (kk < nrow ? rowpart[(s0 + kk) * d + c] : colpart[(kk - nrow) * d + c]);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) acc += v[q];
        }
    }
    meet[sl][el] = acc;
    __syncthreads();
    if (sl == 0 && c < d) {
        double t = meet[0][el];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += meet[q][el];
        y[c] = t;
    }
}

// out[j] = V[j] . w for the basis vectors j < k (one workgroup each); fixed-order sums
__global__ __launch_bounds__(256) void basis_dots_kernel(const double *__restrict__ V, int64_t d,
                                                         const double *__restrict__ w,
                                                         double *__restrict__ out) {
    __shared__ double sh[256];
    const double *v = V + (int64_t)blockIdx.x * d;
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < d; i += 256) a = fma(v[i], w[i], a);
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

// The same in two stages for long vectors: segment s of kDotSegs of vector j -> part[j *
// kDotSegs + s]; lanczos_finish_kernel adds a vector's segments in order.  One workgroup
// per vector walked d = 25k doubles in 19-30 us on 1 to 48 CUs, three times per Lanczos
// step: more than a tenth of the step.
constexpr int kDotSegs = 16;
__global__ __launch_bounds__(256) void basis_dots_part_kernel(const double *__restrict__ V, int64_t d,
                                                              const double *__restrict__ w,
                                                              double *__restrict__ part) {
    __shared__ double sh[256];
    const double *v = V + (int64_t)blockIdx.x * d;
    const int64_t per = (d + kDotSegs - 1) / kDotSegs;
    const int64_t i0 = (int64_t)blockIdx.y * per, i1 = std::min<int64_t>(d, i0 + per);
    double a = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) a = fma(v[i], w[i], a);
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(int64_t)blockIdx.x * kDotSegs + blockIdx.y] = sh[0];
}

// w -= sum_{j<k} c[j] V[j]
__global__ __launch_bounds__(256) void basis_subtract_kernel(const double *__restrict__ V, int64_t d,
                                                             int k, const double *__restrict__ c,
                                                             double *__restrict__ w) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= d) return;
    double a = w[i];
    for (int j = 0; j < k; ++j) a = fma(-c[j], V[(int64_t)j * d + i], a);
    w[i] = a;
}

// out = scale * in
__global__ __launch_bounds__(256) void scale_kernel(const double *__restrict__ in, double scale,
                                                    double *__restrict__ out, int64_t d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < d) out[i] = scale * in[i];
}

// u = sum_{j<k} y[j] V[j]   (the Ritz vector)
__global__ __launch_bounds__(256) void basis_combine_kernel(const double *__restrict__ V, int64_t d,
                                                            int k, const double *__restrict__ y,
                                                            double *__restrict__ u) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= d) return;
    double a = 0.0;
    for (int j = 0; j < k; ++j) a = fma(y[j], V[(int64_t)j * d + i], a);
    u[i] = a;
}

// Eigen-decomposition of a small symmetric matrix on the host (cyclic Jacobi): a (n x n,
// row-major) -> eigenvalues on its diagonal, eigenvectors in the columns of z.
void jacobi_eigh(std::vector<double> &a, std::vector<double> &z, int n) {
    z.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) z[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) off += a[(size_t)p * n + q] * a[(size_t)p * n + q];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double app = a[(size_t)p * n + p], aqq = a[(size_t)q * n + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
                    a[(size_t)k * n + p] = c * akp - sn * akq;
                    a[(size_t)k * n + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
                    a[(size_t)p * n + k] = c * apk - sn * aqk;
                    a[(size_t)q * n + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double zkp = z[(size_t)k * n + p], zkq = z[(size_t)k * n + q];
                    z[(size_t)k * n + p] = c * zkp - sn * zkq;
                    z[(size_t)k * n + q] = sn * zkp + c * zkq;
                }
            }
    }
}

// ---- correlation (datatypes.pyx:173-188: numpy.corrcoef(matrix)) ---------------------
// corrcoef = centre every row, Gram matrix of the centred rows, scale by the diagonal.
// The Gram matrix is the one dense contraction of this library (2 d^3 / 2 flops: 1.6e13 at
// d = 25k) and the one place the matrix cores belong: v_mfma_f64_16x16x4_f64, fp64 in and
// out, so the result is numpy's to rounding (1e-10 asserted; BLAS adds in another order).
constexpr int kGT = 128;  // output tile edge per workgroup (4 waves x 64 x 64)
constexpr int kGK = 16;   // K per LDS tile
constexpr int kGS = 18;   // LDS row stride in doubles: makes the 16 x 4 operand reads of a
                          // wave (lane = row + 16 k) hit 64 distinct banks (ds_read_b64)
typedef double f64x4 __attribute__((ext_vector_type(4)));

// xc (dp rows of ldx doubles, zero-padded) = m - row mean; one wave per row, two passes
__global__ __launch_bounds__(256) void center_rows_kernel(const double *__restrict__ m, int64_t d,
                                                          double *__restrict__ xc, int64_t ldx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d) return;
    const double *p = m + row * d;
    double acc = 0.0;
    for (int64_t c = lane; c < d; c += 64) acc += p[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    const double mean = acc / (double)d;
    double *q = xc + row * ldx;
    for (int64_t c = lane; c < d; c += 64) q[c] = p[c] - mean;
}

// The same in ONE pass (round 4): a workgroup of 1024 threads holds a whole row in registers
// (CH values per thread, d <= 1024 * CH), adds it up, subtracts the mean and writes it out --
// 8 B read + 8 B written per element, every load of the row in flight at once.  The two-pass
// kernel above re-read the 200-KB row for the subtraction (counter traffic x1.50 at d = 24,927)
// behind one wave's worth of dependent 8-byte loads: 3.15 TB/s.  Also writes the zero padding
// of the row (columns d .. ldx), so the caller clears only the padding rows.
// (amdgpu_waves_per_eu(4, 4): 128 registers per lane.  Left to itself the compiler aims at 8
// waves per SIMD, caps the kernel at 64 VGPRs and spills the row -- 208 bytes per lane at
// CH = 32, which the counters showed as twice the write traffic.)
template <int CH>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void center_rows_reg_kernel(const double *__restrict__ m, int64_t d,
                                                               double *__restrict__ xc, int64_t ldx) {
    __shared__ double part[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t row = blockIdx.x;
    // a wave owns 64 * CH consecutive columns and walks them in 512-byte steps.  The row is
    // addressed as a BUFFER of exactly d doubles: a lane beyond the row's end reads 0 and its
    // store is dropped by the range check, so there is no predicate, no branch and no 64-bit
    // column index per load to keep (all of which the first version of this kernel had: 4
    // registers per element, a spill from CH = 24 on).
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(m) + row * d, 0, (int)(d * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        xc + row * ldx, 0, (int)(ldx * 8), 0x00020000);
    const int c0 = wv * (64 * CH) + lane;
    const unsigned voff = (unsigned)c0 * 8u;
    double v[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q) {
        const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, voff + 512u * q, 0, 2);   // nt
        v[q] = __hiloint2double((int)w.y, (int)w.x);
    }
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < CH; ++q) acc += v[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) part[wv] = acc;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += part[k];      // every thread, the same order
    const double mean = tot / (double)d;
#pragma unroll
    for (int q = 0; q < CH; ++q) {
        const double o = c0 + 64 * q < (int)d ? v[q] - mean : 0.0;      // d .. ldx: the padding
        const u32x2_t w = {(unsigned)__double2loint(o), (unsigned)__double2hiint(o)};
        __builtin_amdgcn_raw_buffer_store_b64(w, out_rsrc, voff + 512u * q, 0, 2);
    }
}

// g (dp x dp) upper tiles = xc xc^T.  Workgroup = one 128 x 128 tile, TI <= TJ; wave =
// 64 x 64 = 4 x 4 MFMA tiles (128 accumulator VGPRs).  Both operands are rows of xc
// (K contiguous).  K-tiles of 16 go through LDS as [row][18], double-buffered: while
// tile t is multiplied out of one buffer, tile t + 1 is on its way from HBM into
// registers and is stored into the OTHER buffer afterwards -- one barrier per K-tile
// (the second wave on each SIMD covers the operand reads' LDS latency; reading a step
// ahead in registers spilled).
// The tiles are walked in 8 x 8 patches (blockIdx -> patch, then row, column inside it), so
// that the workgroups in flight at any time share both their row panels and their
// column panels in L2 / the Infinity Cache instead of streaming a full panel per tile.
constexpr int kGP = 8;   // patch edge, in tiles
__global__ __launch_bounds__(256, 2) void gram_kernel(const double *__restrict__ xc, int64_t ldx,
                                                      double *__restrict__ g, int64_t dp, int nt) {
    const int np = (nt + kGP - 1) / kGP;
    const int patch = blockIdx.x / (kGP * kGP), inner = blockIdx.x % (kGP * kGP);
    const int TI = (patch / np) * kGP + inner / kGP, TJ = (patch % np) * kGP + inner % kGP;
    if (TI > TJ || TJ >= nt) return;
    __shared__ __attribute__((aligned(16))) double As[2][kGT * kGS];
    __shared__ __attribute__((aligned(16))) double Bs[2][kGT * kGS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    const int64_t i0 = (int64_t)TI * kGT, j0 = (int64_t)TJ * kGT;
    f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int srow = tid >> 3, skc = (tid & 7) * 2;   // staging: row srow + 32 q, doubles skc, skc+1
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    f64x2 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;       // the K-tile in flight, 128 B per thread
    const double *ga = xc + (i0 + srow) * ldx + skc, *gb = xc + (j0 + srow) * ldx + skc;
#define BB_GLOAD(k0)                                                           \
    do {                                                                       \
        ra0 = *reinterpret_cast<const f64x2 *>(ga + (k0));                     \
        ra1 = *reinterpret_cast<const f64x2 *>(ga + 32 * ldx + (k0));          \
        ra2 = *reinterpret_cast<const f64x2 *>(ga + 64 * ldx + (k0));          \
        ra3 = *reinterpret_cast<const f64x2 *>(ga + 96 * ldx + (k0));          \
        rb0 = *reinterpret_cast<const f64x2 *>(gb + (k0));                     \
        rb1 = *reinterpret_cast<const f64x2 *>(gb + 32 * ldx + (k0));          \
        rb2 = *reinterpret_cast<const f64x2 *>(gb + 64 * ldx + (k0));          \
        rb3 = *reinterpret_cast<const f64x2 *>(gb + 96 * ldx + (k0));          \
    } while (0)
#define BB_LSTORE(buf)                                                         \
    do {                                                                       \
        double *sa = &As[buf][srow * kGS + skc], *sb = &Bs[buf][srow * kGS + skc]; \
        *reinterpret_cast<f64x2 *>(sa) = ra0;                                  \
        *reinterpret_cast<f64x2 *>(sa + 32 * kGS) = ra1;                       \
        *reinterpret_cast<f64x2 *>(sa + 64 * kGS) = ra2;                       \
        *reinterpret_cast<f64x2 *>(sa + 96 * kGS) = ra3;                       \
        *reinterpret_cast<f64x2 *>(sb) = rb0;                                  \
        *reinterpret_cast<f64x2 *>(sb + 32 * kGS) = rb1;                       \
        *reinterpret_cast<f64x2 *>(sb + 64 * kGS) = rb2;                       \
        *reinterpret_cast<f64x2 *>(sb + 96 * kGS) = rb3;                       \
    } while (0)
    const int orow = lane & 15, ok = lane >> 4;       // operand element of this lane
    const double *pa = &As[0][(wr * 64 + orow) * kGS + ok];
    const double *pb = &Bs[0][(wc * 64 + orow) * kGS + ok];
    BB_GLOAD(0);
    BB_LSTORE(0);
    __syncthreads();
    const int ntile = (int)(ldx / kGK);
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        // (the last trip re-reads its own tile into the idle buffer: no branch in the loop)
        BB_GLOAD((int64_t)(t + 1 < ntile ? t + 1 : t) * kGK);
        const double *qa = pa + buf * (kGT * kGS), *qb = pb + buf * (kGT * kGS);
#pragma unroll
        for (int kk = 0; kk < kGK / 4; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = qa[16 * u * kGS + kk * 4];
                b[u] = qb[16 * u * kGS + kk * 4];
            }
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
        // The K-tile in flight goes to LDS BEHIND this tile's 64 MFMAs (4096 cycles of the
        // matrix pipe: the loads have long landed).  Left alone the scheduler hoists the
        // stores -- and with them `s_waitcnt vmcnt(0)` -- to four MFMAs below the loads, and
        // every K-tile then waits out the memory latency with the pipe idle: 58.7 TFLOP/s,
        // 75 % of the peak (round 3); -DBB_GRAM_NO_SCHED restores that for an A/B.  (Weaving the
        // eight stores between the LAST sixteen MFMAs with sched_group_barrier instead: 66.5
        // against 67.4 TFLOP/s -- the operand reads of the last K-step then issue late.)
#ifndef BB_GRAM_NO_SCHED
        __builtin_amdgcn_sched_barrier(0);
#endif
        BB_LSTORE(buf ^ 1);
        __syncthreads();
    }
#undef BB_GLOAD
#undef BB_LSTORE
    // v_mfma_f64_16x16x4_f64 leaves D[g + 4 r][j] in register r of lane 16 g + j (the fp64
    // form interleaves the rows over the lane groups; the fp32 forms hold D[4 g + r][j])
    const int gq = lane >> 4, jj = lane & 15;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            double *o = g + (i0 + wr * 64 + 16 * ta + gq) * dp + j0 + wc * 64 + 16 * tb + jj;
            o[0] = acc[ta][tb].x;
            o[4 * dp] = acc[ta][tb].y;
            o[8 * dp] = acc[ta][tb].z;
            o[12 * dp] = acc[ta][tb].w;
        }
}

__global__ void gram_diag_kernel(const double *__restrict__ g, int64_t dp, int64_t d, double fact_inv,
                                 double *__restrict__ stddev) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < d) stddev[i] = sqrt(g[i * dp + i] * fact_inv);
}

// m (d x d) = clip(((g * 1/(d-1)) / std_i) / std_j, -1, 1), upper tile pairs + mirror
// (numpy.corrcoef's own order of operations)
__global__ __launch_bounds__(kT * 8) void corr_finalize_kernel(const double *__restrict__ g, int64_t dp,
                                                              const double *__restrict__ stddev,
                                                              double fact_inv, double *__restrict__ m,
                                                              int64_t d) {
    __shared__ double tile[kT][kT + 1];
    const int TJ = blockIdx.y, TK = blockIdx.x;
    if (TJ > TK) return;
    const int tx = threadIdx.x % kT, ty = threadIdx.x / kT;
#pragma unroll
    for (int q = 0; q < kT / 8; ++q) {
        const int rr = ty + 8 * q;
        const int64_t j = (int64_t)TJ * kT + rr, k = (int64_t)TK * kT + tx;
        double v = 0.0;
        if (j < d && k < d) {
            // lower cells of a diagonal tile: take the symmetric upper one (only the upper
            // 128-tiles of g were computed, and within them every cell)
            const int64_t a = j <= k ? j : k, b = j <= k ? k : j;
            v = g[a * dp + b] * fact_inv;
            v = v / stddev[a];
            v = v / stddev[b];
            v = v > 1.0 ? 1.0 : (v < -1.0 ? -1.0 : v);     // NaN stays NaN, as numpy.clip
            m[j * d + k] = v;
        }
        tile[rr][tx] = v;
    }
    __syncthreads();
    if (TJ == TK) return;
#pragma unroll
    for (int q = 0; q < kT / 8; ++q) {
        const int rr = ty + 8 * q;
        const int64_t kk = (int64_t)TK * kT + rr, jj = (int64_t)TJ * kT + tx;
        if (kk < d && jj < d) m[kk * d + jj] = tile[tx][rr];
    }
}

int cm_check(const bb_cm *cm, const char *who) {
    if (!cm) return bb::fail(BB_ERR_INVALID, std::string(who) + ": contact map is NULL");
    return bb::enter_device(cm->device);
}

// y = M x from the upper triangle (symv_upper_kernel + symv_reduce_kernel), x and y on the
// device, enqueued on the handle's stream.  The work list and the partial-sum buffers are
// made on first use and whenever the edge has changed (filter), and kept with the handle.
hipError_t symv_enqueue(bb_cm *cm, const double *dx, double *dy) {
    const int64_t d = cm->d;
    const int64_t nrb = (d + kSvRows - 1) / kSvRows, nseg = (d + kSvSeg - 1) / kSvSeg;
    if (cm->sv_d != d) {
        std::vector<int2> items;
        // longest items first: the dispatcher hands them out in order
        for (int64_t S = nseg - 1; S >= 0; --S)
            for (int64_t I = 0; I < nrb && I * kSvRows < (S + 1) * (int64_t)kSvSeg; ++I)
                if (std::max<int64_t>(I * kSvRows, S * kSvSeg) < std::min<int64_t>(d, (S + 1) * (int64_t)kSvSeg))
                    items.push_back(make_int2((int)I, (int)S));
        std::stable_sort(items.begin(), items.end(), [&](const int2 &a, const int2 &b) {
            auto len = [&](const int2 &t) {
                return std::min<int64_t>(d, (t.y + 1) * (int64_t)kSvSeg) -
                       std::max<int64_t>((int64_t)t.x * kSvRows, (int64_t)t.y * kSvSeg);
            };
            return len(a) > len(b);
        });
        const size_t item_bytes = (items.size() * sizeof(int2) + 255) & ~(size_t)255;
        const size_t need = item_bytes + (size_t)(nseg + nrb) * (size_t)d * 8;
        if (need > cm->sv_bytes) {
            (void)hipFree(cm->sv_buf);
            cm->sv_buf = nullptr;
            cm->sv_bytes = 0;
            hipError_t e = hipMalloc(&cm->sv_buf, need);
            if (e != hipSuccess) return e;
            cm->sv_bytes = need;
        }
        hipError_t e = hipMemcpyAsync(cm->sv_buf, items.data(), items.size() * sizeof(int2),
                                      hipMemcpyHostToDevice, cm->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(cm->stream);   // `items` dies with this scope
        if (e != hipSuccess) return e;
        cm->sv_items = (int)items.size();
        cm->sv_d = d;
    }
    const size_t item_bytes = ((size_t)cm->sv_items * sizeof(int2) + 255) & ~(size_t)255;
    double *rowpart = (double *)((char *)cm->sv_buf + item_bytes);
    double *colpart = rowpart + nseg * d;
    hipError_t e = bb::launch(symv_upper_kernel, dim3((unsigned)cm->sv_items), dim3(256), 0, cm->stream,
                              (const double *)cm->m, d, dx, (const int2 *)cm->sv_buf, rowpart, colpart);
    if (e == hipSuccess)
        e = bb::launch(symv_reduce_kernel, dim3((unsigned)((d + 127) / 128)), dim3(1024), 0, cm->stream,
                       (const double *)rowpart, (const double *)colpart, d, (int)nseg, dy);
    return e;
}

// One Lanczos step's scalars, on the device (no host round trip inside a cycle):
//   mode 0  alpha[k] = dc[k]                   after the first Gram-Schmidt pass
//   mode 1  alpha[k] += dc[k]                  after the second
//   mode 2  beta[k] = sqrt(dc[0]); inv[0] = 1 / beta[k], or 0 when the new direction has
//           vanished against |alpha_k| + beta_{k-1} (an invariant subspace: every later
//           basis vector of the cycle is then 0 and drops out of the tridiagonal matrix)
//   part != NULL: first dc[j] = the sum of vector j's kDotSegs segment dots, j < n_vec
__global__ void lanczos_scalars_kernel(double *__restrict__ dc, int k, int mode,
                                       double *__restrict__ alpha, double *__restrict__ beta,
                                       double *__restrict__ inv, const double *__restrict__ part,
                                       int n_vec) {
    if (part != nullptr) {
        if ((int)threadIdx.x < n_vec) {
            double t = 0.0;
            for (int q = 0; q < kDotSegs; ++q) t += part[(int)threadIdx.x * kDotSegs + q];
            dc[threadIdx.x] = t;
        }
        __syncthreads();
    }
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (mode == 0) {
        alpha[k] = dc[k];
    } else if (mode == 1) {
        alpha[k] += dc[k];
    } else {
        const double b2 = dc[0];
        const double b = sqrt(b2 > 0.0 ? b2 : 0.0);
        const double ref = fabs(alpha[k]) + (k > 0 ? beta[k - 1] : 0.0);
        const bool alive = b > 1e-14 * (ref > 0.0 ? ref : 1.0);
        beta[k] = alive ? b : 0.0;
        inv[0] = alive ? 1.0 / b : 0.0;
    }
}

// out = scale[0] * in, the scale read on the device
__global__ __launch_bounds__(256) void scale_by_kernel(const double *__restrict__ in,
                                                       const double *__restrict__ scale,
                                                       double *__restrict__ out, int64_t d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < d) out[i] = scale[0] * in[i];
}

// correlation's two matrix-sized temporaries (centred rows, Gram matrix): ONE grow-only
// allocation per DEVICE, shared by every map on it and guarded by a mutex (calls on one device
// serialise, as bb_band.hip's context does).  Round 3 kept one per HANDLE for the map's
// lifetime: a genome-wide loop holding one ContactMap per chromosome then held 3x the
// matrices' memory.  bb_cm_release_scratch gives it back.
struct CorrScratch {
    std::mutex mu;
    void *buf = nullptr;
    size_t bytes = 0;
};
CorrScratch *corr_scratch(int device) {
    static std::mutex table_mu;
    static std::map<int, CorrScratch *> table;
    std::lock_guard<std::mutex> lock(table_mu);
    auto it = table.find(device);
    if (it != table.end()) return it->second;
    CorrScratch *c = new CorrScratch();   // lives for the process
    table[device] = c;
    return c;
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
    if (e == hipSuccess) e = bb::acquire_stream(device, &cm->stream);
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
    bb::release_stream(cm->device, cm->stream);     // synchronises it
    (void)hipFree(cm->m);
    (void)hipFree(cm->sv_buf);
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

int bb_cm_scatter_ex(bb_cm *cm, const double *triples, int64_t n, int32_t resolution,
                     int32_t row_major, uint8_t *present, int32_t *on_grid) {
    BB_TRY(cm_check(cm, "bb_cm_scatter"));
    BB_REQUIRE(n >= 0 && (triples != nullptr || n == 0), "bb_cm_scatter: bad triples");
    BB_REQUIRE(n < (int64_t)0x7fffffff, "bb_cm_scatter: too many triples");
    BB_REQUIRE(resolution != 0, "bb_cm_scatter: resolution is 0");
    BB_REQUIRE((present == nullptr) == (on_grid == nullptr),
               "bb_cm_scatter_ex: present and on_grid go together");
    const int64_t d = cm->d;
    const bool want = present != nullptr;
    bb::DevBuf tr, bad, pres;
    hipError_t e = tr.alloc((size_t)n * 3 * sizeof(double));
    if (e == hipSuccess) e = bad.alloc(2 * sizeof(int));          // {bad, off the grid}
    if (e == hipSuccess && want) e = pres.alloc((size_t)d);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("bb_cm_scatter: ") + hipGetErrorString(e));
    hipStream_t st = cm->stream;
    int host_flags[2] = {0, 0};
    if (n > 0) e = hipMemcpyAsync(tr.p, triples, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(cm->m, 0, (size_t)d * d * sizeof(double), st);  // pyx:99
    if (e == hipSuccess) e = hipMemsetAsync(bad.p, 0, 2 * sizeof(int), st);
    if (e == hipSuccess && want) e = hipMemsetAsync(pres.p, 0, (size_t)d, st);
    if (e == hipSuccess && n > 0) {
        const unsigned grid = (unsigned)((n + 255) / 256);
        const int64_t ts = row_major ? 3 : 1, cs = row_major ? 1 : n;
        e = bb::launch(scatter_mark_kernel, dim3(grid), dim3(256), 0, st, (const double *)tr.p, n, ts,
                       cs, (double)resolution, d, cm->m, (int *)bad.p,
                       (unsigned char *)(want ? pres.p : nullptr), (int *)bad.p + 1);
        if (e == hipSuccess)
            e = bb::launch(scatter_store_kernel, dim3(grid), dim3(256), 0, st,
                           (const double *)tr.p, n, ts, cs, (double)resolution, d, cm->m);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpy(host_flags, bad.p, 2 * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess && want) e = hipMemcpy(present, pres.p, (size_t)d, hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_cm_scatter: ") + hipGetErrorString(e));
    if (host_flags[0]) {
        // marks of valid triples may be left in the matrix: clear it, the map is unusable
        (void)hipMemset(cm->m, 0, (size_t)d * d * sizeof(double));
        return bb::fail(BB_ERR_INVALID,
                        "bb_cm_scatter: a position maps to a bin outside [0, n_bins]");
    }
    if (want) *on_grid = host_flags[1] ? 0 : 1;
    return BB_OK;
}

int bb_cm_scatter(bb_cm *cm, const double *triples, int64_t n, int32_t resolution) {
    return bb_cm_scatter_ex(cm, triples, n, resolution, 0, nullptr, nullptr);
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
    const char *env = getenv("BB_CM_NORMALIZE_TILE");
    const int tile = env ? atoi(env) : 128;
    const bool big = tile != 32 && d >= 256;              // small maps: more, smaller tiles
    if (e == hipSuccess && big) {
        const int nt_edge = tile == 64 ? 64 : 128;
        const int lds = nt_edge * (nt_edge + 1) * 8;
        const uint64_t nt = (uint64_t)((d + nt_edge - 1) / nt_edge), pairs = nt * (nt + 1) / 2;
        static bool attr_done[2] = {false, false};
        const void *fn = nt_edge == 64 ? (const void *)normalize128_kernel<64>
                                       : (const void *)normalize128_kernel<128>;
        if (!attr_done[nt_edge == 64]) {
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr_done[nt_edge == 64] = e == hipSuccess;
        }
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cm->device);
        const char *envw = getenv("BB_CM_NORMALIZE_WGS");
        const uint64_t per_cu = envw ? (uint64_t)atoi(envw) : (nt_edge == 64 ? 4 : 1);
        const unsigned grid = (unsigned)std::min<uint64_t>(pairs, (uint64_t)cus * per_cu);
        if (e == hipSuccess && nt_edge == 64)
            e = bb::launch(normalize128_kernel<64>, dim3(grid), dim3(512), (size_t)lds, st, cm->m, d, n_bins,
                           (const double *)kr.p, (const double *)ke.p, (unsigned)pairs);
        else if (e == hipSuccess)
            e = bb::launch(normalize128_kernel<128>, dim3(grid), dim3(1024), (size_t)lds, st, cm->m, d,
                           n_bins, (const double *)kr.p, (const double *)ke.p, (unsigned)pairs);
    } else if (e == hipSuccess) {
        const uint64_t nt = (uint64_t)((d + kT - 1) / kT);
        if (nt * (nt + 1) / 2 > 0x7fffffffull)
            return bb::fail(BB_ERR_INVALID, "bb_cm_normalize: matrix too large for one launch");
        e = bb::launch(normalize_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(kT * 8), 0, st, cm->m, d, n_bins,
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
    e = bb::launch(column_sums_kernel, dim3((unsigned)((cm->d + kSumWG - 1) / kSumWG)), dim3(kSumWG), 0,
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
    e = bb::launch(column_sums_kernel, dim3((unsigned)((d + kSumWG - 1) / kSumWG)), dim3(kSumWG), 0, st,
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
    // Compaction IN PLACE, a band of rows at a time through a small bounce buffer: the
    // new matrix is the old one read in order with elements left out, so new row r lands at
    // [r dn, (r+1) dn), in front of every element a later row still needs (those start at
    // old_row(r+1) d >= (r+1) dn).  No matrix-sized allocation: the first touch of a fresh
    // 1-5 GB block costs 170-350 ms on this platform (tools/alloc_probe.py), the whole filter 3 ms.
    // The buffer keeps its size; only d shrinks.
    if (dn > 0 && dn < d) {
        // BB_CM_FILTER_BOUNCE (elements): tests shrink the bounce buffer so that small
        // matrices go through many bands and a ragged last one
        int64_t bounce = kFilterBounceElems;
        if (const char *v = getenv("BB_CM_FILTER_BOUNCE")) bounce = std::max<int64_t>(1, atoll(v));
        const int64_t band = std::max<int64_t>(1, std::min<int64_t>(dn, bounce / dn));
        bb::DevBuf tmp;
        e = tmp.alloc((size_t)(band * dn) * sizeof(double));
        if (e != hipSuccess)
            return bb::fail(BB_ERR_NOMEM, std::string("bb_cm_filter: ") + hipGetErrorString(e));
        for (int64_t r0 = 0; r0 < dn && e == hipSuccess; r0 += band) {
            const int64_t r1 = std::min(dn, r0 + band);
            e = bb::launch(gather_kernel,
                           dim3((unsigned)((dn + 255) / 256), (unsigned)std::min<int64_t>(r1 - r0, 32768)),
                           dim3(256), 0, st, (const double *)cm->m, d, (const int *)idx.p,
                           (double *)tmp.p, dn, r0, r1);
            if (e == hipSuccess)
                e = hipMemcpyAsync(cm->m + r0 * dn, tmp.p, (size_t)((r1 - r0) * dn) * sizeof(double),
                                   hipMemcpyDeviceToDevice, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess)
            return bb::fail(BB_ERR_HIP, std::string("bb_cm_filter: ") + hipGetErrorString(e));
    }
    cm->d = dn;
    if (d_new) *d_new = dn;
    return BB_OK;
}

int bb_cm_symv(bb_cm *cm, const double *x, double *y) {
    BB_TRY(cm_check(cm, "bb_cm_symv"));
    BB_REQUIRE(x != nullptr && y != nullptr, "bb_cm_symv: NULL argument");
    const int64_t d = cm->d;
    bb::DevBuf dx, dy;
    hipError_t e = dx.alloc((size_t)d * 8);
    if (e == hipSuccess) e = dy.alloc((size_t)d * 8);
    if (e != hipSuccess) return bb::fail(BB_ERR_NOMEM, std::string("bb_cm_symv: ") + hipGetErrorString(e));
    e = hipMemcpyAsync(dx.p, x, (size_t)d * 8, hipMemcpyHostToDevice, cm->stream);
    const char *env = getenv("BB_CM_SYMV_FULL");          // 1: read both triangles (round 2)
    if (e == hipSuccess && env && atoi(env) != 0)
        e = bb::launch(symv_kernel, dim3((unsigned)((d + 4 * kSymvRows - 1) / (4 * kSymvRows))), dim3(256), 0, cm->stream,
                       (const double *)cm->m, d, (const double *)dx.p, (double *)dy.p);
    else if (e == hipSuccess)
        e = symv_enqueue(cm, (const double *)dx.p, (double *)dy.p);
    if (e == hipSuccess) e = hipStreamSynchronize(cm->stream);
    if (e == hipSuccess) e = hipMemcpy(y, dy.p, (size_t)d * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return bb::fail(BB_ERR_HIP, std::string("bb_cm_symv: ") + hipGetErrorString(e));
    return BB_OK;
}

// Restarted Lanczos with full re-orthogonalisation for the eigenpair of largest
// magnitude -- the pair scipy.sparse.linalg.eigsh(matrix, k=1) returns (ARPACK's
// default which='LM'), reference blueberry/datatypes.pyx:234.  Every matrix-vector
// product is one sweep of the UPPER TRIANGLE of the resident matrix (symv_upper_kernel; the
// matrix is taken to be symmetric, as eigsh takes it); the basis (<= kBasis vectors of d
// doubles) stays on the device and a whole cycle of up to kBasis steps is enqueued
// without a host round trip; only the scalars of the tridiagonal matrix cross PCIe, once
// per cycle.  The sign of the vector is fixed: its largest-magnitude component is > 0.
int bb_cm_eigenvector(bb_cm *cm, double *vec, double *eigenvalue, double tol, int64_t max_matvecs,
                      int64_t *matvecs_used, double *residual) {
    BB_TRY(cm_check(cm, "bb_cm_eigenvector"));
    BB_REQUIRE(vec != nullptr, "bb_cm_eigenvector: vec is NULL");
    BB_REQUIRE(tol >= 0.0 && max_matvecs >= 1, "bb_cm_eigenvector: bad tol / max_matvecs");
    const int64_t d = cm->d;
    constexpr int kBasis = 48;
    const int m = (int)std::min<int64_t>(kBasis, d);
    bb::DevBuf V, w, coef, yv;
    hipError_t e = V.alloc((size_t)(m + 1) * d * 8);
    if (e == hipSuccess) e = w.alloc((size_t)d * 8);
    if (e == hipSuccess) e = coef.alloc((size_t)(m + 1) * 8);
    if (e == hipSuccess) e = yv.alloc((size_t)(m + 1) * 8);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("bb_cm_eigenvector: ") + hipGetErrorString(e));
    hipStream_t st = cm->stream;
    double *dV = (double *)V.p, *dw = (double *)w.p, *dc = (double *)coef.p, *dy = (double *)yv.p;
    const dim3 gvec((unsigned)((d + 255) / 256)), b256(256);
    // start vector: a fixed pseudo-random direction (no global RNG state is touched)
    std::vector<double> host((size_t)d);
    {
        uint64_t sdt = 0x9E3779B97F4A7C15ull;
        double nrm = 0.0;
        for (int64_t i = 0; i < d; ++i) {
            sdt = sdt * 6364136223846793005ull + 1442695040888963407ull;
            host[(size_t)i] = (double)((sdt >> 11) & 0xFFFFFFFFull) / 4294967296.0 + 0.25;
            nrm += host[(size_t)i] * host[(size_t)i];
        }
        nrm = 1.0 / sqrt(nrm);
        for (auto &h : host) h *= nrm;
    }
    e = hipMemcpy(dV, host.data(), (size_t)d * 8, hipMemcpyHostToDevice);
    int64_t used = 0;
    double theta = 0.0, resid = 0.0;
    std::vector<double> alpha, beta, T, Z;
    // alpha[m] | beta[m] | 1 / beta of the current step: written by lanczos_scalars_kernel,
    // read back once per cycle -- a cycle of up to 48 steps is enqueued without a host
    // round trip (round 2 synchronised three times per step: 13 of the 56 ms of a call at
    // d = 24,927)
    bb::DevBuf sc;
    if (e == hipSuccess) e = sc.alloc((size_t)(2 * m + 1 + (m + 1) * kDotSegs) * 8);
    double *d_alpha = (double *)sc.p, *d_beta = d_alpha + m, *d_inv = d_beta + m, *d_part = d_inv + 1;
    const bool two_stage = d >= 4096;
    const char *env_full = getenv("BB_CM_SYMV_FULL");
    const bool full = env_full && atoi(env_full) != 0;
    bool done = false;
    while (e == hipSuccess && !done) {
        const int steps = (int)std::min<int64_t>(m, std::max<int64_t>(1, max_matvecs - used));
        e = hipMemsetAsync(sc.p, 0, (size_t)(2 * m + 1) * 8, st);
        for (int k = 0; k < steps && e == hipSuccess; ++k) {
            // w = M v_k
            if (full)
                e = bb::launch(symv_kernel, dim3((unsigned)((d + 4 * kSymvRows - 1) / (4 * kSymvRows))), b256, 0, st,
                               (const double *)cm->m, d, (const double *)(dV + (int64_t)k * d), dw);
            else
                e = symv_enqueue(cm, (const double *)(dV + (int64_t)k * d), dw);
            ++used;
            // coefficients against the whole basis (alpha_k is the last one), subtract, and
            // once more for the rounding the first pass leaves (classical Gram-Schmidt x 2)
            for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {
                // the dots (in two stages for long vectors) and alpha_k, THEN the subtraction
                // (which overwrites w; dc is not written again before it has been read)
                if (two_stage)
                    e = bb::launch(basis_dots_part_kernel, dim3((unsigned)(k + 1), kDotSegs), b256, 0, st,
                                   (const double *)dV, d, (const double *)dw, d_part);
                else
                    e = bb::launch(basis_dots_kernel, dim3((unsigned)(k + 1)), b256, 0, st,
                                   (const double *)dV, d, (const double *)dw, dc);
                if (e == hipSuccess)
                    e = bb::launch(lanczos_scalars_kernel, dim3(1), dim3(64), 0, st, dc, k, pass, d_alpha,
                                   d_beta, d_inv, (const double *)(two_stage ? d_part : nullptr), k + 1);
                if (e == hipSuccess)
                    e = bb::launch(basis_subtract_kernel, gvec, b256, 0, st, (const double *)dV, d,
                                   k + 1, (const double *)dc, dw);
            }
            // beta_k = |w|, v_{k+1} = w / beta_k (0 once the direction has vanished)
            if (e == hipSuccess) {
                if (two_stage)
                    e = bb::launch(basis_dots_part_kernel, dim3(1, kDotSegs), b256, 0, st,
                                   (const double *)dw, d, (const double *)dw, d_part);
                else
                    e = bb::launch(basis_dots_kernel, dim3(1), b256, 0, st, (const double *)dw, d,
                                   (const double *)dw, dc);
            }
            if (e == hipSuccess)
                e = bb::launch(lanczos_scalars_kernel, dim3(1), dim3(64), 0, st, dc, k, 2, d_alpha,
                               d_beta, d_inv, (const double *)(two_stage ? d_part : nullptr), 1);
            if (k + 1 <= m && e == hipSuccess)
                e = bb::launch(scale_by_kernel, gvec, b256, 0, st, (const double *)dw,
                               (const double *)d_inv, dV + (int64_t)(k + 1) * d, d);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        std::vector<double> ab((size_t)2 * m);
        if (e == hipSuccess) e = hipMemcpy(ab.data(), sc.p, (size_t)2 * m * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) break;
        // the cycle ends where the new direction vanished (an invariant subspace), else at `steps`
        int n_steps = steps;
        for (int k = 0; k < steps; ++k)
            if (ab[(size_t)m + k] == 0.0) { n_steps = k + 1; break; }
        used -= steps - n_steps;               // products of vanished directions do not count
        alpha.assign(ab.begin(), ab.begin() + n_steps);
        beta.assign(ab.begin() + m, ab.begin() + m + n_steps);
        const int n = (int)alpha.size();
        T.assign((size_t)n * n, 0.0);
        for (int i = 0; i < n; ++i) {
            T[(size_t)i * n + i] = alpha[(size_t)i];
            if (i + 1 < n) T[(size_t)i * n + i + 1] = T[(size_t)(i + 1) * n + i] = beta[(size_t)i];
        }
        jacobi_eigh(T, Z, n);
        int best = 0;
        for (int i = 1; i < n; ++i)
            if (fabs(T[(size_t)i * n + i]) > fabs(T[(size_t)best * n + best])) best = i;
        theta = T[(size_t)best * n + best];
        std::vector<double> y((size_t)n);
        for (int i = 0; i < n; ++i) y[(size_t)i] = Z[(size_t)i * n + best];
        resid = fabs(beta[(size_t)n - 1] * y[(size_t)n - 1]);   // |M u - theta u| of the Ritz pair
        // the Ritz vector becomes basis vector 0: the answer, or the restart vector
        e = hipMemcpy(dy, y.data(), (size_t)n * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess)
            e = bb::launch(basis_combine_kernel, gvec, b256, 0, st, (const double *)dV, d, n,
                           (const double *)dy, dw);
        if (e == hipSuccess)
            e = bb::launch(basis_dots_kernel, dim3(1), b256, 0, st, (const double *)dw, d,
                           (const double *)dw, dc);
        double n2 = 1.0;
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipMemcpy(&n2, dc, 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            e = bb::launch(scale_kernel, gvec, b256, 0, st, (const double *)dw,
                           n2 > 0.0 ? 1.0 / sqrt(n2) : 1.0, dV, d);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        done = resid <= tol * fabs(theta) || resid == 0.0 || used >= max_matvecs || n == (int)d;
    }
    if (e == hipSuccess) e = hipMemcpy(host.data(), dV, (size_t)d * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_cm_eigenvector: ") + hipGetErrorString(e));
    // sign: the component of largest magnitude is positive (first one on ties)
    int64_t big = 0;
    for (int64_t i = 1; i < d; ++i)
        if (fabs(host[(size_t)i]) > fabs(host[(size_t)big])) big = i;
    const double sgn = host[(size_t)big] < 0.0 ? -1.0 : 1.0;
    for (int64_t i = 0; i < d; ++i) vec[i] = sgn * host[(size_t)i];
    if (eigenvalue) *eigenvalue = theta;
    if (matvecs_used) *matvecs_used = used;
    if (residual) *residual = resid;
    return BB_OK;
}

int bb_cm_release_scratch(int device) {
    BB_TRY(bb::enter_device(device));
    CorrScratch *sc = corr_scratch(device);
    std::lock_guard<std::mutex> lock(sc->mu);
    (void)hipFree(sc->buf);
    sc->buf = nullptr;
    sc->bytes = 0;
    return BB_OK;
}

int bb_cm_correlation(bb_cm *cm, double *tflops) {
    BB_TRY(cm_check(cm, "bb_cm_correlation"));
    const int64_t d = cm->d;
    const int64_t dp = bb::round_up(d, kGT), ldx = bb::round_up(d, kGK);
    CorrScratch *scr = corr_scratch(cm->device);
    std::lock_guard<std::mutex> scratch_lock(scr->mu);
    // centred rows | Gram matrix | standard deviations: ONE grow-only allocation kept with
    // the handle.  Round 2 allocated the two matrix-sized temporaries per call, and on this
    // platform the first touch of a fresh block of that size costs 0.17-0.35 s
    // (tools/alloc_probe.py): the whole call took 1.5x its Gram kernel.
    struct View { void *p; } xc, g, sd;
    const size_t xc_bytes = ((size_t)dp * ldx * 8 + 255) & ~(size_t)255;
    const size_t g_bytes = ((size_t)dp * dp * 8 + 255) & ~(size_t)255;
    const size_t need = xc_bytes + g_bytes + (size_t)d * 8;
    hipError_t e = hipSuccess;
    if (need > scr->bytes) {
        (void)hipFree(scr->buf);
        scr->buf = nullptr;
        scr->bytes = 0;
        e = hipMalloc(&scr->buf, need);
        if (e == hipSuccess) scr->bytes = need;
    }
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("bb_cm_correlation: ") + hipGetErrorString(e));
    xc.p = scr->buf;
    g.p = (char *)scr->buf + xc_bytes;
    sd.p = (char *)scr->buf + xc_bytes + g_bytes;
    hipStream_t st = cm->stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const double fact_inv = 1.0 / (double)(d - 1);   // numpy: true_divide(1, N - ddof); d = 1 -> inf
    // a whole row in the registers of one workgroup where it fits (d <= 32,768); else two passes
    const int ch = d >= 2048 && d <= 32768 && !getenv("BB_CM_CENTER_OLD") ? (int)((d + 1023) / 1024) : 0;
    if (ch == 0) {
        e = hipMemsetAsync(xc.p, 0, (size_t)dp * ldx * 8, st);
        if (e == hipSuccess)
            e = bb::launch(center_rows_kernel, dim3((unsigned)((d + 3) / 4)), dim3(256), 0, st,
                           (const double *)cm->m, d, (double *)xc.p, ldx);
    } else {
        // the kernel writes each row's own padding; only the padding ROWS are cleared here
        if (dp > d) e = hipMemsetAsync((double *)xc.p + d * ldx, 0, (size_t)(dp - d) * ldx * 8, st);
#define BB_CENTER(CHV)                                                                              \
    bb::launch(center_rows_reg_kernel<CHV>, dim3((unsigned)d), dim3(1024), 0, st, (const double *)cm->m, \
               d, (double *)xc.p, ldx)
        if (e == hipSuccess)
            e = ch <= 4 ? BB_CENTER(4) : ch <= 8 ? BB_CENTER(8) : ch <= 16 ? BB_CENTER(16)
              : ch <= 24 ? BB_CENTER(24) : BB_CENTER(32);
#undef BB_CENTER
    }
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, st);
    const unsigned nt = (unsigned)(dp / kGT);
    const unsigned np = (nt + kGP - 1) / kGP;
    if (e == hipSuccess)
        e = bb::launch(gram_kernel, dim3(np * np * kGP * kGP), dim3(256), 0, st,
                       (const double *)xc.p, ldx, (double *)g.p, dp, (int)nt);
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess)
        e = bb::launch(gram_diag_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, st,
                       (const double *)g.p, dp, d, fact_inv, (double *)sd.p);
    if (e == hipSuccess) {
        const unsigned ntf = (unsigned)((d + kT - 1) / kT);
        e = bb::launch(corr_finalize_kernel, dim3(ntf, ntf), dim3(kT * 8), 0, st, (const double *)g.p,
                       dp, (const double *)sd.p, fact_inv, cm->m, d);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && tflops) {
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, e0, e1);
        // flops of the tiles that were computed: nt (nt + 1) / 2 tiles of 2 * 128 * 128 * ldx
        *tflops = ms > 0.f ? (double)nt * (nt + 1) / 2 * 2.0 * kGT * kGT * (double)ldx / (ms * 1e-3) / 1e12
                           : 0.0;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_cm_correlation: ") + hipGetErrorString(e));
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
