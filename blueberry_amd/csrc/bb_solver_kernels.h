// bb_solver_kernels.h -- device code of the 3D-structure solver (gfx950 only): the
// stress + gradient sweep, the deterministic partial reduce (+ update / exchange /
// peer push), the update kernels, the pack kernels and the read-only sweep used for
// measurement.  Included by bb_solver.hip (the host side and the C-ABI) and by nothing
// else: everything here lives in that translation unit's anonymous namespace.
//
// Specification: docs/SPEC.md (build-authored; the reference has no solver,
// SURVEY.md section 0).  Data layout and kernel design: DESIGN.md 3-4.
//
// Layout recap (SPEC 3).  The matrix is cut into vw x vw tiles (vw = 512 columns;
// 128 for small fp64 problems, see Lay<> below), upper-triangular tiles only, ordered
// column-strip major (J, then I).  A tile is vw/rpu "units"; a unit is rpu matrix rows
// x vw columns = 8 KiB, row-major (fp32: 4 rows of 2 KiB; fp64: 2 rows of 4 KiB, or 8
// rows of 1 KiB).
// One wave reads a matrix row of a unit with LPR 16-B-per-lane loads and lane
// l always owns the same LPR*VPL columns of the strip.  That makes the column
// side of the symmetric update register-resident for a whole strip sweep (no
// cross-lane traffic), and only the row side needs one DPP wave reduction per
// matrix row -- amortised over 8 pairs per lane in fp32.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "bb_ablate.h"
#include "blueberry_hip.h"

namespace {


// --------------------------------------------------------------------------
// type traits
// --------------------------------------------------------------------------
template <typename T>
struct Traits;
template <>
struct Traits<float> {
    using Vec = float4;
    static constexpr int VPL = 4;  // elements per lane per 16-B load
    static __device__ __forceinline__ float eps2() { return 1e-30f; }
};
template <>
struct Traits<double> {
    using Vec = double2;
    static constexpr int VPL = 2;
    static __device__ __forceinline__ double eps2() { return 1e-300; }
};

// Wish distances below this are stored as 0 = "no constraint" by every pack kernel: far
// below the distance floor eps of SPEC 2.2 (1e-15 / 1e-150), and what lets the kernels
// turn "delta > 0" into a 0/1 weight with one clamped multiply (delta * 2^100 resp.
// 2^1000 saturates to 1 for every delta that survives the flush).
template <typename T>
__host__ __device__ constexpr double wish_floor() { return sizeof(T) == 4 ? 1e-30 : 1e-290; }

// Shape of a unit (8 KiB = 8 wave-loads) and of a strip.  LPR 16-byte loads per lane
// and matrix row, VW = 64 * VPL * LPR columns per strip, RPU = 8 / LPR matrix rows.
//   fp32          4 rows x 512 columns: 8 pairs per lane and row, packed math
//   fp64 wide     2 rows x 512 columns: also 8 pairs per lane and row, so the DPP
//                 reduction, the row-coordinate fetch and the row-sum stores are
//                 amortised over 4x more pairs than in the narrow shape (+40 % at
//                 N=20k); 96 VGPRs of column state, so at most 2 waves per SIMD
//   fp64 narrow   8 rows x 128 columns: 4x smaller column partials and 4x more
//                 blocks for the reduce -- what small problems want (N=963: 14.5 us
//                 per iteration against 26 us in the wide shape); bb_common.h picks
//                 it for n_bins <= kF64WideFrom
template <typename T, bool W>
struct Lay;
template <bool W>
struct Lay<float, W> {
    static constexpr int LPR = 2, VW = 512, RPU = 4;
    static constexpr int MIN_WG = abl::kDppRowSum ? 4 : 2;   // __launch_bounds__: <= 128 VGPRs (the
                                                // -DBB_MFMA_ROWSUM experiment keeps 12 selectors and
                                                // an accumulator more: <= 256)
    static constexpr bool SCALAR_XROW = true;   // 12 row coordinates through the scalar cache
};
template <>
struct Lay<double, true> {
    static constexpr int LPR = 4, VW = 512, RPU = 2;
    static constexpr int MIN_WG = 2;            // <= 256 VGPRs
    static constexpr bool SCALAR_XROW = true;   // 6 doubles = 12 SGPRs, double-buffered
};
template <>
struct Lay<double, false> {
    static constexpr int LPR = 1, VW = 128, RPU = 8;
    // <= 256 VGPRs.  This shape only ever runs one workgroup of 4 waves per CU (at most 4096
    // bins: fewer than 65k units, waves_per_cu() = 4), so a cap of 128 registers bought no
    // occupancy and, from round 3 on, cost a 20-byte scratch spill in the stress sweep
    // (tests/test_host.py::test_no_product_kernel_spills guards against the next one).
    static constexpr int MIN_WG = 2;
    static constexpr bool SCALAR_XROW = false;  // 24 doubles x 2 would not fit the SGPR file:
                                                // one per-lane load + v_readlane instead
};

template <int C>
__device__ __forceinline__ float elem(const float4 &v) {
    if constexpr (C == 0) return v.x;
    if constexpr (C == 1) return v.y;
    if constexpr (C == 2) return v.z;
    return v.w;
}
template <int C>
__device__ __forceinline__ double elem(const double2 &v) {
    if constexpr (C == 0) return v.x;
    return v.y;
}

// --------------------------------------------------------------------------
// cross-lane helpers (wave64, DPP; no LDS)
// --------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get(double v) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes; the total is valid in lanes 48..63 (we read lane 63).
// Fixed tree => bitwise deterministic.
template <typename T>
__device__ __forceinline__ T wave_sum_hi(T v) {
    v += dpp_get<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_get<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_get<0x141, 0xF>(v);  // row_half_mirror
    v += dpp_get<0x140, 0xF>(v);  // row_mirror
    v += dpp_get<0x142, 0xA>(v);  // row_bcast15 -> rows 1,3
    v += dpp_get<0x143, 0xC>(v);  // row_bcast31 -> rows 2,3
    return v;
}

// The same tree for three fp32 values at once, as one asm block.  Interleaving
// the three chains puts two independent VALU ops between every write of a
// register and its next DPP read, which is exactly the 2 wait states that
// hazard needs (the compiler serialises the chains and pads each step with
// s_nop), and the two row_bcast steps become single v_add_f32_dpp with a
// partial row_mask (disabled rows keep their value) instead of mov+mov+add.
// Only the leading s_nop is needed: the inputs were just written by VALU code.
__device__ __forceinline__ void wave_sum_hi3(float &a, float &b, float &c) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(a), "+v"(b), "+v"(c));
}
__device__ __forceinline__ void wave_sum_hi3(double &a, double &b, double &c) {
    a = wave_sum_hi(a);
    b = wave_sum_hi(b);
    c = wave_sum_hi(c);
}

// Value of `v` in lane `l` (compile-time l) as a wave-uniform scalar.
__device__ __forceinline__ float lane_value(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ double lane_value(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// Branch-free "lane 63 only" store of a row's three sums through a raw buffer
// descriptor: every other lane carries an out-of-range offset and the
// hardware range check drops its store.  (An `if (lane == 63)` around a plain
// store splits the unit into basic blocks, and LLVM then sinks all column-side
// accumulation below the last of them, spilling 32 pairs of forces.)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kDropOffset = 0x80000000u;

__device__ __forceinline__ void store_row3(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, double a,
                                           double b, double c) {
    u32x4 v = {(unsigned)__double2loint(a), (unsigned)__double2hiint(a),
               (unsigned)__double2loint(b), (unsigned)__double2hiint(b)};
    u32x2 w = {(unsigned)__double2loint(c), (unsigned)__double2hiint(c)};
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, voff + 16, 0, 0);
}

// --------------------------------------------------------------------------
// stress + gradient kernel
// --------------------------------------------------------------------------
// What the sweep computes per pair (i, j), template parameter OP:
//   kOpStress : residual force of SPEC 2.3 -> gradient (row and column side) + stress
//   kOpMatvec2: y_i += delta_ij^2 * x_j and y_j += delta_ij^2 * x_i for three
//               right-hand sides held in the coordinate slots: Y = (D o D) X, the
//               kernel of classical-MDS / spectral initialisation
enum { kOpStress = 0, kOpMatvec2 = 1 };

// 1/sqrt(d2) and sqrt(d2) in fp64 from the v_rsq_f64 seed (relative error <= 2^-23) by ONE
// third-order step: with e = 1 - d2 r0^2, r = r0 (1 - e)^(-1/2) = r0 (1 + e/2 + 3 e^2/8 + ...);
// the term dropped is 5 e^3 / 16 < 1e-20, so both results are good to the last bit or two.
// 6 instructions behind the seed (two Newton steps on r plus one on the root took 10).
__device__ __forceinline__ void rsqrt_sqrt_f64(double d2, double &rinv, double &dist) {
    const double r0 = __builtin_amdgcn_rsq(d2);
    const double t = d2 * r0;                    // ~ sqrt(d2)
    const double e = fma(-t, r0, 1.0);
    const double q = e * fma(e, 0.375, 0.5);
    rinv = fma(r0, q, r0);
    dist = fma(t, q, t);
}
// 1.0 where delta > 0, else 0.0, in one instruction (see wish_floor): the compiler folds
// min(max(x, 0), 1) into the clamp output modifier (v_ldexp_f64 ... clamp).  Not inline asm:
// asm statements are convergent in HIP device code, and a loop that holds one is not
// unrolled at run time (row_owner_kernel).
__device__ __forceinline__ double weight01_f64(double delta) {
    return __builtin_fmin(__builtin_fmax(delta * 0x1p1000, 0.0), 1.0);
}

// Pair math for one matrix row of a unit: VPL pairs per lane.
template <typename T, int C, int OP>
__device__ __forceinline__ void pair_step(const typename Traits<T>::Vec &drow, T xi, T yi, T zi,
                                          const T (&xj)[Traits<T>::VPL][3],
                                          T (&gc)[Traits<T>::VPL][3], T &gx, T &gy, T &gz, T &s) {
    const T delta = elem<C>(drow);
    if constexpr (OP == kOpMatvec2) {
        const T a = delta * delta;
        gx += a * xj[C][0]; gy += a * xj[C][1]; gz += a * xj[C][2];
        gc[C][0] += a * xi; gc[C][1] += a * yi; gc[C][2] += a * zi;
        return;
    }
    const T dx = xi - xj[C][0], dy = yi - xj[C][1], dz = zi - xj[C][2];
    const T d2 = fma(dx, dx, fma(dy, dy, fma(dz, dz, Traits<T>::eps2())));  // SPEC 2.2
    if constexpr (sizeof(T) == 8 && !abl::kF64Libm) {
        // fp64 has no packed forms and its pair math alone is worth the HBM time of a
        // unit, so every instruction counts: 25 + v_rsq per pair
        T rinv, dist;
        rsqrt_sqrt_f64(d2, rinv, dist);
        // (dist - delta) or 0 in ONE instruction: w is 1, or 0 exactly where delta is 0, so
        // dist * w - delta is dist - delta (the product is exact) or 0 - 0
        const T res = fma(dist, weight01_f64(delta), -delta);
        s = fma(res, res, s);
        const T coef = res * rinv;  // (d - delta) / d ; the factor 2 is applied in the reduce
        gx = fma(coef, dx, gx); gy = fma(coef, dy, gy); gz = fma(coef, dz, gz);
        gc[C][0] = fma(-coef, dx, gc[C][0]);
        gc[C][1] = fma(-coef, dy, gc[C][1]);
        gc[C][2] = fma(-coef, dz, gc[C][2]);
        return;
    }
    T rinv, dist;
    if constexpr (sizeof(T) == 4) {
        rinv = __builtin_amdgcn_rsqf(d2);
        dist = d2 * rinv;
    } else {
        dist = sqrt(d2);            // timing experiment (BB_ABL_F64_LIBM): ~55 fp64 ops
        rinv = 1.0 / dist;
    }
    const T res = delta > T(0) ? dist - delta : T(0);
    s = fma(res, res, s);
    const T coef = res * rinv;  // (d - delta) / d ; the factor 2 is applied in the reduce
    const T fx = coef * dx, fy = coef * dy, fz = coef * dz;
    gx += fx; gy += fy; gz += fz;
    gc[C][0] -= fx; gc[C][1] -= fy; gc[C][2] -= fz;
}

// Matrix rows are read exactly once per launch: NT = true marks the loads
// non-temporal so the stream does not evict X and the partials from L2 / MALL.
typedef float f32x4_raw __attribute__((ext_vector_type(4)));
typedef double f64x2_raw __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ float4 stream_load(const float4 *p) {
    if constexpr (NT) {
        const f32x4_raw v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_raw *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}
template <bool NT>
__device__ __forceinline__ double2 stream_load(const double2 *p) {
    if constexpr (NT) {
        const f64x2_raw v = __builtin_nontemporal_load(reinterpret_cast<const f64x2_raw *>(p));
        return make_double2(v.x, v.y);
    } else {
        return *p;
    }
}

// Where a wave's rolling window is refilled from: the k-th wave-load (1 KiB) of the NEXT
// unit.  WinPtr (product): plain pointers.  WinBuf (-DBB_BUFFER_WINDOW, A/B only): buffer
// loads through a descriptor that covers exactly the wave's chunk, so that the refills of
// a wave's last unit fall out of range and cost no traffic -- measured 1.5-2.5 us SLOWER per
// launch at N=12,000-17,700 (the scheduler clusters the refills differently:
// profiles/r03_window_ab.txt); the pointer form gets the same saving by pointing all lanes
// of those last refills at one line (see window_of in the kernel).
template <typename Vec>
struct WinPtr {
    const Vec *p;
    template <bool NT>
    __device__ __forceinline__ Vec load(int k) const { return stream_load<NT>(p + k * 64); }
};
template <typename Vec>
struct WinBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned off;      // byte offset of this lane's 16 bytes of wave-load 0 of the unit
    template <bool NT>
    __device__ __forceinline__ Vec load(int k) const {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + (unsigned)k * 1024u, 0,
                                                              NT ? 2 : 0);   // aux bit 1 = nt
        if constexpr (sizeof(Vec) == sizeof(float4) && std::is_same<Vec, float4>::value)
            return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z),
                               __uint_as_float(v.w));
        else
            return make_double2(__hiloint2double((int)v.y, (int)v.x),
                                __hiloint2double((int)v.w, (int)v.z));
    }
};

// One unit, generic (fp64) form: RPU matrix rows of LPR wave-loads each.  Load k of
// row r of the CURRENT unit is consumed from d[r*LPR + k], which is then refilled at
// once with the same load of the NEXT unit, so 8 KiB per wave stay in flight with a
// single register window.  xrow.get(q): the unit's q-th row coordinate, wave-uniform.
template <typename T, bool W, bool NT, int OP, typename XR, typename WIN>
__device__ __forceinline__ void process_unit(typename Traits<T>::Vec (&d)[8], const XR &xrow,
                                             const WIN &next,
                                             const T (&xj)[(Lay<T, W>::LPR)][Traits<T>::VPL][3],
                                             T (&gc)[(Lay<T, W>::LPR)][Traits<T>::VPL][3],
                                             double &stress, __amdgpu_buffer_rsrc_t row_rsrc,
                                             unsigned row_voff) {
    constexpr int VPL = Traits<T>::VPL, LPR = Lay<T, W>::LPR;
    static_assert(Lay<T, W>::RPU * LPR == 8, "a unit is 8 wave-loads");
    T s = T(0);
#pragma unroll
    for (int r = 0; r < Lay<T, W>::RPU; ++r) {
        const T xi = xrow.get(3 * r), yi = xrow.get(3 * r + 1), zi = xrow.get(3 * r + 2);
        T gx = T(0), gy = T(0), gz = T(0);
#pragma unroll
        for (int k = 0; k < LPR; ++k) {
            pair_step<T, 0, OP>(d[r * LPR + k], xi, yi, zi, xj[k], gc[k], gx, gy, gz, s);
            pair_step<T, 1, OP>(d[r * LPR + k], xi, yi, zi, xj[k], gc[k], gx, gy, gz, s);
            if constexpr (VPL == 4) {
                pair_step<T, 2, OP>(d[r * LPR + k], xi, yi, zi, xj[k], gc[k], gx, gy, gz, s);
                pair_step<T, 3, OP>(d[r * LPR + k], xi, yi, zi, xj[k], gc[k], gx, gy, gz, s);
            }
            d[r * LPR + k] = next.template load<NT>(r * LPR + k);
        }
        wave_sum_hi3(gx, gy, gz);
        // one 3-element store per matrix row, from the lane holding the sums
        store_row3(row_rsrc, row_voff + r * 3 * (unsigned)sizeof(T), gx, gy, gz);
        // keep the rows in program order: otherwise the scheduler interleaves the
        // rows for ILP and spills
        __builtin_amdgcn_sched_barrier(0);
    }
    stress += (double)s;
}

// ---- fp64, 2 x 512 units: the row sums go through the matrix pipe ---------------
// In fp64 the pair math alone (~35 ops per pair, no packed forms) is worth the whole HBM
// time of a unit, so everything else the VALU does is paid in full.  The largest such
// item was the cross-lane reduction of the 6 row sums of a unit: v_add_f64 has no DPP
// form, so each of the 6 tree steps of each sum is 2 v_mov_dpp + 1 add -- 108
// instructions per unit, a sixth of the unit.  v_mfma_f64_16x16x4_f64 adds across lanes
// for free: with A = one partial sum per lane (A[i][k] = lane i + 16 k) and B = a 0/1
// selector with ones in column v, D[i][v] += sum_k A[i][k], and six of them accumulate the
// unit's six sums side by side in the columns 0..5 of ONE 16 x 16 accumulator.  What is
// left for the VALU: 3 adds over the lane's 4 accumulator rows and 2 lane-preserving
// shuffles over the 4 lane groups.  The matrix pipe is idle otherwise and runs beside the
// partner wave's VALU work; the order of the additions is fixed (bitwise reproducible).
typedef double f64x4 __attribute__((ext_vector_type(4)));

// t + (t of the lane 16 / 32 further on or back): x = {r0, r0, r2, r2}, y = {r1, r1, r3, r3}
__device__ __forceinline__ double swap_sum16(double t) {
    const unsigned lo = (unsigned)__double2loint(t), hi = (unsigned)__double2hiint(t);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double swap_sum32(double t) {
    const unsigned lo = (unsigned)__double2loint(t), hi = (unsigned)__double2hiint(t);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

__device__ __forceinline__ float swap_sum16(float t) {
    const unsigned b = __float_as_uint(t);
    const auto a = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float swap_sum32(float t) {
    const unsigned b = __float_as_uint(t);
    const auto a = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
typedef float f32x4v __attribute__((ext_vector_type(4)));

// The three sums of a matrix row (x, y, z components) over the 64 lanes in 10 VALU
// instructions instead of 18 (round 3; the sweep is issue-bound at the clock the chip runs it
// at, see the kernel's comment).  gfx950's lane swaps exchange half of one register with the
// other half of a second one, so ONE swap + ONE add both adds across the halves and sorts two
// sums apart: after the 32-lane step the lower half holds x, the upper y (z is folded with
// itself); after the 16-lane step the even 16-lane rows hold x | y, the odd ones z; four DPP
// steps then add up each row.  Result: the row's x total in every lane 0..15, z in 16..31
// and 48..63, y in 32..47.  Fixed tree: bitwise reproducible.
__device__ __forceinline__ float row_sum3_swap(float gx, float gy, float gz) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(gx), __float_as_uint(gy), false, false);
    const float xy = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(gz), __float_as_uint(gz), false, false);
    const float zz = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    const auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(xy), __float_as_uint(zz), false, false);
    float w = __uint_as_float(c[0]) + __uint_as_float(c[1]);
    w += dpp_get<0xB1, 0xF>(w);   // quad_perm [1,0,3,2]
    w += dpp_get<0x4E, 0xF>(w);   // quad_perm [2,3,0,1]
    w += dpp_get<0x141, 0xF>(w);  // row_half_mirror
    w += dpp_get<0x140, 0xF>(w);  // row_mirror
    return w;
}
// which of a unit's 12 row sums a lane keeps under row_sum3_swap: row r = lane % 16 (< 4) of
// component x (lanes 0..3), z (16..19) or y (32..35); -1: none
__device__ __forceinline__ int swap_rowsum_slot(int lane) {
    const int r = lane & 15;
    if (r >= 4 || lane >= 48) return -1;
    return 3 * r + (lane < 16 ? 0 : (lane < 32 ? 2 : 1));
}

template <bool NT, int OP, bool DEFER, typename XR, typename WIN>
__device__ __forceinline__ void process_unit_f64w(double2 (&d)[8], const XR &xrow,
                                                  const WIN &next,
                                                  const double (&xj)[4][2][3], double (&gc)[4][2][3],
                                                  const double (&sel)[6], double &stress,
                                                  __amdgpu_buffer_rsrc_t row_rsrc, unsigned row_voff,
                                                  int stage_idx) {
    extern __shared__ __attribute__((aligned(16))) float row_lds[];
    double s = 0.0;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const double xi = xrow.get(3 * r), yi = xrow.get(3 * r + 1), zi = xrow.get(3 * r + 2);
        double gx = 0.0, gy = 0.0, gz = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pair_step<double, 0, OP>(d[r * 4 + k], xi, yi, zi, xj[k], gc[k], gx, gy, gz, s);
            pair_step<double, 1, OP>(d[r * 4 + k], xi, yi, zi, xj[k], gc[k], gx, gy, gz, s);
            d[r * 4 + k] = next.template load<NT>(r * 4 + k);
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(gx, sel[3 * r + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(gy, sel[3 * r + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(gz, sel[3 * r + 2], acc, 0, 0, 0);
    }
    // lane (g = lane / 16, j = lane % 16) holds four of the 16 rows of column j: add them,
    // then the 4 lane groups with the gfx950 lane swaps (VALU, no LDS round trip):
    // permlane16_swap exchanges the odd 16-lane rows of one register with the even rows
    // of the other, permlane32_swap the upper half with the lower half
    double t = (acc.x + acc.y) + (acc.z + acc.w);
    t = swap_sum16(t);
    t = swap_sum32(t);                   // every lane with lane % 16 == v: the unit's sum v
    // lanes 48..53 keep one sum each; parked in LDS or stored, exactly one of the two lands
    if constexpr (DEFER) *reinterpret_cast<double *>(row_lds + stage_idx) = t;
    const u32x2 bits = {(unsigned)__double2loint(t), (unsigned)__double2hiint(t)};
    __builtin_amdgcn_raw_buffer_store_b64(bits, row_rsrc, row_voff, 0, 0);
    stress += s;
}

// ---- fp32: the same unit with explicit 2-wide packed math (v_pk_*_f32) -------
// A lane owns 8 columns of the 512-wide strip: load k (k = 0,1) brings columns
// k*256 + 4*lane .. +3, and within a load the pairs {0,1} and {2,3} share every
// instruction that has a packed form.  Column-side state is [load][half]
// [component] so that the two pairs of a half sit in adjacent registers.
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct StripF32 {
    f32x2 x[2][2][3];  // coordinates of this lane's columns
    f32x2 g[2][2][3];  // column-side gradient accumulators, same shape
};

// w = 1 where delta > 0, else 0, for both halves in ONE instruction:
// clamp(delta * 2^100) saturates any delta >= 2^-100 to 1 and leaves 0 at 0
// (delta is never negative or NaN: the pack kernels store 0 for "no
// constraint" and flush anything below 1e-30).  Replaces 2 v_cmp + 2 v_cndmask.
__device__ __forceinline__ f32x2 weight01(f32x2 delta) {
    f32x2 w;
    const f32x2 big = {0x1p100f, 0x1p100f};
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(w) : "v"(delta), "v"(big));
    return w;
}

template <int K, int H, bool FIRST, int OP>
__device__ __forceinline__ void pair_step2(f32x2 delta, f32x2 xi, f32x2 yi, f32x2 zi, StripF32 &st,
                                           f32x2 &rx, f32x2 &ry, f32x2 &rz, f32x2 &s2) {
    if constexpr (OP == kOpMatvec2) {
        const f32x2 a = delta * delta;
        if constexpr (FIRST) {
            rx = a * st.x[K][H][0]; ry = a * st.x[K][H][1]; rz = a * st.x[K][H][2];
        } else {
            rx += a * st.x[K][H][0]; ry += a * st.x[K][H][1]; rz += a * st.x[K][H][2];
        }
        st.g[K][H][0] += a * xi;
        st.g[K][H][1] += a * yi;
        st.g[K][H][2] += a * zi;
        return;
    }
    const f32x2 eps2 = {1e-30f, 1e-30f};
    const f32x2 dx = xi - st.x[K][H][0], dy = yi - st.x[K][H][1], dz = zi - st.x[K][H][2];
    const f32x2 d2 = dx * dx + (dy * dy + (dz * dz + eps2));  // SPEC 2.2: |d|^2 + eps^2
    f32x2 rinv;
    if constexpr (abl::kNoRsq) {
        rinv = d2 * eps2;
    } else {
        rinv.x = __builtin_amdgcn_rsqf(d2.x);
        rinv.y = __builtin_amdgcn_rsqf(d2.y);
    }
    f32x2 res = d2 * rinv - delta;
    if constexpr (!abl::kNoMask) res *= weight01(delta);  // (dist - delta) or 0
    s2 += res * res;
    const f32x2 coef = res * rinv;
    if constexpr (FIRST) {
        rx = coef * dx; ry = coef * dy; rz = coef * dz;
    } else {
        rx += coef * dx; ry += coef * dy; rz += coef * dz;
    }
    if constexpr (!abl::kNoCol) {
        st.g[K][H][0] -= coef * dx;
        st.g[K][H][1] -= coef * dy;
        st.g[K][H][2] -= coef * dz;
    }
}

// d[] holds the unit's 8 wave-loads in row order: d[2*r + k] = row r, load k.
// Row sums: a fixed-tree 64-lane DPP reduction per matrix row (wave_sum_hi3).  Round 3 tried
// them through the matrix pipe, as the fp64 unit does (-DBB_MFMA_ROWSUM: v_mfma_f32_16x16x4_f32
// with a 0/1 selector per sum, the unit's 12 sums side by side in one accumulator, 3 adds and two
// lane swaps at the end): 72 DPP adds fewer per unit, but 164 VGPRs, a different clustering of
// the refills and the wait for the accumulator at the end of every unit -- 2.7-6 % slower at
// every size (profiles/r03_mfma_rowsum_ab.txt).  Why it looked promising: at two waves per SIMD
// the chip runs this kernel at 1.75-1.83 GHz (power) and the SIMD's VALU is then busy most of a
// unit's time (418 VALU instructions per unit and wave, 32 of them quarter-rate v_rsq_f32);
// tools/unit_trace.py prints each XCD's clock.
template <bool NT, int OP, bool DEFER, bool REFILL = true, typename WIN>
__device__ __forceinline__ void process_unit_f32(float4 (&d)[8], const float (&xrow)[12],
                                                 const WIN &next, StripF32 &st,
                                                 double &stress, __amdgpu_buffer_rsrc_t row_rsrc,
                                                 unsigned row_voff, int stage_idx,
                                                 const float (&sel)[12]) {
    extern __shared__ __attribute__((aligned(16))) float row_lds[];
    f32x2 s2 = {0.f, 0.f};
    // The 12 row sums of the unit are collected into lanes 48..59 of one register
    // (after the reduction every lane >= 48 holds the wave total) and leave with
    // ONE 48-byte store per unit: a 12-byte store per row costs as much VMEM issue
    // as a 1-KiB load and measured 5.6 % of the kernel.
    const int slot = (int)(threadIdx.x & 63) - 48;  // value index this lane keeps, if 0..11
    const int comp = slot - 3 * (slot / 3);     // 0,1,2 = x,y,z
    float keep = 0.f;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    // (row_sum3_swap) the matrix row of the unit whose sum this lane keeps, or -1
    const int swap_role = (((threadIdx.x & 15) < 4) && ((threadIdx.x & 63) < 48)) ? (int)(threadIdx.x & 15) : -1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float xs = xrow[3 * r], ys = xrow[3 * r + 1], zs = xrow[3 * r + 2];
        const f32x2 xi = {xs, xs}, yi = {ys, ys}, zi = {zs, zs};
        f32x2 rx, ry, rz, qx, qy, qz;  // row-side sums of load 0 / load 1
        pair_step2<0, 0, true, OP>(f32x2{d[2 * r].x, d[2 * r].y}, xi, yi, zi, st, rx, ry, rz, s2);
        pair_step2<0, 1, false, OP>(f32x2{d[2 * r].z, d[2 * r].w}, xi, yi, zi, st, rx, ry, rz, s2);
        if constexpr (REFILL) d[2 * r] = next.template load<NT>(2 * r);
        // (no sched_barrier here: letting the scheduler mix the rows of a unit measured
        // 1 % faster at N=50k and 6 % faster at 1/8 size; it stays within 125 VGPRs)
        pair_step2<1, 0, true, OP>(f32x2{d[2 * r + 1].x, d[2 * r + 1].y}, xi, yi, zi, st, qx, qy, qz, s2);
        pair_step2<1, 1, false, OP>(f32x2{d[2 * r + 1].z, d[2 * r + 1].w}, xi, yi, zi, st, qx, qy, qz, s2);
        if constexpr (REFILL) d[2 * r + 1] = next.template load<NT>(2 * r + 1);
        rx += qx; ry += qy; rz += qz;
        float gx = rx.x + rx.y, gy = ry.x + ry.y, gz = rz.x + rz.y;
        if constexpr (abl::kSwapRowSum) {
            const float w = row_sum3_swap(gx, gy, gz);
            keep = (swap_role == r) ? w : keep;
        } else if constexpr (abl::kDppRowSum) {
            if constexpr (!abl::kNoDpp) wave_sum_hi3(gx, gy, gz);
            const float mine = comp == 0 ? gx : (comp == 1 ? gy : gz);
            keep = (slot >= 3 * r && slot < 3 * r + 3) ? mine : keep;
        } else {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(gx, sel[3 * r + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(gy, sel[3 * r + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(gz, sel[3 * r + 2], acc, 0, 0, 0);
        }
    }
    if constexpr (!abl::kDppRowSum && !abl::kSwapRowSum) {
        // lane (g = lane / 16, j = lane % 16) holds rows 4g..4g+3 of column j: add them, then
        // the four lane groups (gfx950 lane swaps: VALU, no LDS); every lane with
        // lane % 16 == v then holds the unit's sum v -- lanes 48..59 keep theirs
        float t = (acc.x + acc.y) + (acc.z + acc.w);
        t = swap_sum16(t);
        keep = swap_sum32(t);
    }
    if constexpr (abl::kNoStore) {
        asm volatile("" ::"v"(keep));
    } else {
        // DEFER: both are issued for every unit and exactly one of them lands -- the LDS
        // slot is a dummy word while the unit is stored directly, the store's lanes are
        // all out of range (free) while the unit is parked (see the kernel)
        if constexpr (DEFER) row_lds[stage_idx] = keep;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(keep), row_rsrc, row_voff, 0, 0);
    }
    stress += (double)(s2.x + s2.y);
    if constexpr (abl::kSchedRefill > 0) {
        // experiment (-DBB_SCHED_REFILL=n): ask the scheduler for a refill after every n VALU
        // instructions, i.e. right behind the half row it replaces, instead of where it
        // puts them by itself (two in the middle of the unit, six at its end)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x002, abl::kSchedRefill, 0);   // VALU
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                   // one VMEM read
        }
    }
}

__device__ __forceinline__ void load_strip_f32(StripF32 &st, const float *__restrict__ X, int j0,
                                               int lane) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float4 *p = reinterpret_cast<const float4 *>(
            X + ((int64_t)j0 + k * 256 + (int64_t)lane * 4) * 3);
        const float4 a = p[0], b = p[1], c = p[2];  // x0 y0 z0 x1 | y1 z1 x2 y2 | z2 x3 y3 z3
        st.x[k][0][0] = f32x2{a.x, a.w}; st.x[k][0][1] = f32x2{a.y, b.x};
        st.x[k][0][2] = f32x2{a.z, b.y};
        st.x[k][1][0] = f32x2{b.z, c.y}; st.x[k][1][1] = f32x2{b.w, c.z};
        st.x[k][1][2] = f32x2{c.x, c.w};
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) st.g[k][h][c3] = f32x2{0.f, 0.f};
    }
}

__device__ __forceinline__ void store_strip_f32(const StripF32 &st, float *__restrict__ slot,
                                                int lane) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float4 *p = reinterpret_cast<float4 *>(slot + (k * 256 + (int64_t)lane * 4) * 3);
        const auto &g = st.g[k];
        p[0] = make_float4(g[0][0].x, g[0][1].x, g[0][2].x, g[0][0].y);
        p[1] = make_float4(g[0][1].y, g[0][2].y, g[1][0].x, g[1][1].x);
        p[2] = make_float4(g[1][2].x, g[1][0].y, g[1][1].y, g[1][2].y);
    }
}

template <typename T>
__device__ __forceinline__ const typename Traits<T>::Vec *unit_ptr(const T *__restrict__ units,
                                                                   int64_t ul, int lane) {
    using Vec = typename Traits<T>::Vec;
    return reinterpret_cast<const Vec *>(units) + ul * 512 + lane;  // 8 KiB = 512 x 16 B
}

// coordinates of a unit's rows: 3*RPU consecutive elements, one per lane
template <typename T, bool W>
__device__ __forceinline__ T load_xrow(const T *__restrict__ X, int i0, int lane) {
    return X[(int64_t)i0 * 3 + (lane < 3 * Lay<T, W>::RPU ? lane : 0)];
}

// Generic (fp64) column-strip state: load k of a row brings columns
// k*64*VPL + lane*VPL .. +VPL-1, so lane l owns 3*VPL consecutive coordinates per k.
template <typename T, bool W>
__device__ __forceinline__ void load_strip(T (&xj)[(Lay<T, W>::LPR)][Traits<T>::VPL][3],
                                           const T *__restrict__ X, int j0, int lane) {
    using Vec = typename Traits<T>::Vec;
    constexpr int VPL = Traits<T>::VPL;
#pragma unroll
    for (int k = 0; k < Lay<T, W>::LPR; ++k) {
        const Vec *p = reinterpret_cast<const Vec *>(
            X + ((int64_t)j0 + k * 64 * VPL + (int64_t)lane * VPL) * 3);
        Vec a = p[0], b = p[1], c = p[2];
        if constexpr (VPL == 4) {
            xj[k][0][0] = a.x; xj[k][0][1] = a.y; xj[k][0][2] = a.z;
            xj[k][1][0] = a.w; xj[k][1][1] = b.x; xj[k][1][2] = b.y;
            xj[k][2][0] = b.z; xj[k][2][1] = b.w; xj[k][2][2] = c.x;
            xj[k][3][0] = c.y; xj[k][3][1] = c.z; xj[k][3][2] = c.w;
        } else {
            xj[k][0][0] = a.x; xj[k][0][1] = a.y; xj[k][0][2] = b.x;
            xj[k][1][0] = b.y; xj[k][1][1] = c.x; xj[k][1][2] = c.y;
        }
    }
}

template <typename T, bool W>
__device__ __forceinline__ void store_strip(const T (&gc)[(Lay<T, W>::LPR)][Traits<T>::VPL][3],
                                            T *__restrict__ slot, int lane) {
    using Vec = typename Traits<T>::Vec;
    constexpr int VPL = Traits<T>::VPL;
#pragma unroll
    for (int k = 0; k < Lay<T, W>::LPR; ++k) {
        Vec *p = reinterpret_cast<Vec *>(slot + ((int64_t)k * 64 * VPL + (int64_t)lane * VPL) * 3);
        const auto &g = gc[k];
        if constexpr (VPL == 4) {
            p[0] = make_float4(g[0][0], g[0][1], g[0][2], g[1][0]);
            p[1] = make_float4(g[1][1], g[1][2], g[2][0], g[2][1]);
            p[2] = make_float4(g[2][2], g[3][0], g[3][1], g[3][2]);
        } else {
            p[0] = make_double2(g[0][0], g[0][1]);
            p[1] = make_double2(g[0][2], g[1][0]);
            p[2] = make_double2(g[1][1], g[1][2]);
        }
    }
}

// Diagnostic build only (-DBB_WAVE_TRACE): time stamp k (0..7) of wave w, 10-ns ticks of the
// constant-rate clock, into a region of its own behind the per-wave stress partials
// (nothing reads it but bb_solver_debug_wave_trace).  Folds away in the product build.
__device__ __forceinline__ void wave_stamp(double *stresspart, int n_waves, int w, int k) {
    if constexpr (abl::kWaveTrace || abl::kUnitTrace) {
        if ((threadIdx.x & 63) == 0) {
            unsigned long long *t = reinterpret_cast<unsigned long long *>(stresspart + n_waves);
            unsigned long long v = (unsigned long long)wall_clock64();
            if (k == 4) {     // where the wave ran: XCC id | HW_ID
                unsigned xcc, hw;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                v = ((unsigned long long)xcc << 32) | hw;
            }
            t[(long long)w * 8 + k] = v;
        }
    }
}

// One wave = one contiguous chunk of units; 4 independent waves per workgroup.
// No LDS, no barriers, no atomics: results are bitwise reproducible.
//
// Arguments are separate __restrict__ pointers (not a struct) so that the
// read-only index arrays are provably unclobbered and load through the scalar
// cache.  Unit indices are 32-bit (a rank holds < 2^31 units = 16 TiB).
//   units      this rank's units, 8 KiB (RPU rows x VW columns) each
//   X          (n_pad, 3) coordinates
//   udesc      per local unit {i0, j0}
//   chunk_q/_r units per wave: n_local = n_waves * q + r, the first r waves take q + 1
//   wave_slots per wave {first private column-partial slot, the workgroup's shared slot of
//              the strip the wave ENDS in (-1: the wave has no units)}; see the epilogue
//   rowpart    3*RPU elements per unit, base shifted to the rank's first tile
//   colpart    3*VW elements per slot
//   stresspart one double per wave: the stress of the strip the wave ENDS in
//   stress_slot one double per private column slot: the stress of a strip the wave left
//
// DEFER (fp32 only): the row sums do not leave the wave unit by unit.  Writing 48
// bytes to a fresh line per 8 KiB read costs 7.5 % of the kernel at N=50k, and not in
// issue: a store whose lanes are all out of range is free, so is one that keeps
// hitting the same line, and grouping 16 units into 768-byte bursts changes nothing
// -- the write-back cache decides when dirty lines go to HBM, and it sends them into
// a saturated read stream.  The column partials of the same size cost almost nothing
// because they are written when a wave is done.  So, when a wave's whole chunk of
// row sums fits in LDS (cap_units * 48 B per wave: N=50k on one GPU, every multi-GPU
// share), they are parked there and written out as one contiguous burst when the
// wave has finished reading: -5.8 % kernel time at N=50k.  A longer chunk parks its
// LAST cap_units units and stores the ones before them directly.
//
// WPB = waves per workgroup.  4: one wave per SIMD and workgroup.  8 (used when a CU
// holds 8 waves): the two waves that share a SIMD -- wave k and k + 4 -- sit in ONE
// workgroup and keep each other's pace.  Left alone, the SIMD's issue arbitration favours
// the older of two co-resident waves on every conflict, and with static, equal chunks
// that adds up: per-wave time stamps (tools/wave_trace.py) have the favoured partner
// finish its chunk 15-25 % before the other, which then runs the tail alone with half the
// bytes in flight.  Each wave posts the number of units it has done in LDS and reads its
// partner's; whoever is behind raises its priority (s_setprio) for the next unit.  The
// results do not depend on any of it: chunks, slots and summation order stay static.
template <typename T, bool W, bool NT, int OP, bool DEFER, int WPB>
__global__ __launch_bounds__(64 * WPB, (Lay<T, W>::MIN_WG)) void stress_grad_kernel(
    const T *__restrict__ units, const T *__restrict__ X, const int2 *__restrict__ udesc,
    int chunk_q, int chunk_r, const int2 *__restrict__ wave_slots, T *__restrict__ rowpart,
    T *__restrict__ colpart, double *__restrict__ stresspart, int cap_units, int lds_wave_floats,
    int wg_map, int dense_u0, double *__restrict__ stress_slot) {
    using Vec = typename Traits<T>::Vec;
    constexpr int VPL = Traits<T>::VPL;
    constexpr int VW = Lay<T, W>::VW;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave in workgroup
    // Which run of WPB chunks this workgroup sweeps: a bijection of the block index (the
    // host checks it is one).  0: the identity.  m > 0: (b * m) mod n.  -1: block b -- on XCD
    // b mod 8 under round-robin placement -- takes chunk (b mod 8) * n / 8 + b / 8, i.e.
    // every XCD sweeps one contiguous eighth of the units.  Speed only: every chunk is swept
    // exactly once whatever the placement.
    const int wg = wg_map == 0 ? (int)blockIdx.x
                 : wg_map > 0 ? (int)(((unsigned)blockIdx.x * (unsigned)wg_map) % gridDim.x)
                              : (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    const int w = wg * WPB + wib;
    // wave w owns units [w*q + min(w, r), +q (+1 if w < r)): arithmetic, not a table --
    // one dependent memory round trip less before the wave's first matrix load
    const int ua = w * chunk_q + (w < chunk_r ? w : chunk_r);
    const int ub = ua + chunk_q + (w < chunk_r ? 1 : 0);
    const int n_waves_all = gridDim.x * WPB;
    wave_stamp(stresspart, n_waves_all, w, 0);      // (diagnostic build only)
    wave_stamp(stresspart, n_waves_all, w, 4);
    double stress = 0.0;
    // the shared column slots of this workgroup's waves (epilogue), read NOW: behind the
    // fences further down the compiler no longer takes them through the scalar cache, and a
    // vector load there waits -- vmcnt is in order -- for every store the wave has in flight
    int ws_shared[WPB];
#pragma unroll
    for (int k = 0; k < WPB; ++k) ws_shared[k] = wave_slots[(int64_t)wg * WPB + k].y;
    // DEFER: this wave's parking space, cap_units * 12 floats + 4 dummy words
    extern __shared__ __attribute__((aligned(16))) float row_lds[];
    // this wave's LDS region: row-sum parking while it sweeps, its last column partial at
    // the end (lds_wave_floats >= cap_units * 12 + 4 and >= 3 * VW elements of T)
    const int stage0 = wib * lds_wave_floats;
    // WPB = 8: progress words of the 8 waves, behind the regions
    int *progress = reinterpret_cast<int *>(row_lds + WPB * lds_wave_floats);
    int partner_done = 0;
    const int park_from = (ub - ua) > cap_units ? (ub - ua) - cap_units : 0;
    // fp64 2 x 512 units: column selectors of the MFMA row reduction (process_unit_f64w)
    double sel[6];
#pragma unroll
    for (int v = 0; v < 6; ++v) sel[v] = (lane & 15) == v ? 1.0 : 0.0;
    float sel32[12];   // the same for the 12 row sums of an fp32 unit (process_unit_f32)
#pragma unroll
    for (int v = 0; v < 12; ++v) sel32[v] = (lane & 15) == v ? 1.f : 0.f;

    if (ua < ub) {
        int slot = wave_slots[w].x;
        Vec d[8];  // the unit's 8 wave-loads (8 KiB), in memory order
        // column-strip state: coordinates + gradient accumulators of this lane's columns
        struct Generic { T xj[(Lay<T, W>::LPR)][VPL][3], gc[(Lay<T, W>::LPR)][VPL][3]; };
        using Strip = typename std::conditional<sizeof(T) == 4, StripF32, Generic>::type;
        Strip st;
        auto strip_load = [&](int j0) __attribute__((always_inline)) {
            if constexpr (sizeof(T) == 4) {
                load_strip_f32(st, X, j0, lane);
            } else {
                load_strip<T, W>(st.xj, X, j0, lane);
#pragma unroll
                for (int k = 0; k < Lay<T, W>::LPR; ++k)
#pragma unroll
                    for (int c = 0; c < VPL; ++c)
                        st.gc[k][c][0] = st.gc[k][c][1] = st.gc[k][c][2] = T(0);
            }
        };
        auto strip_store = [&](int sl) __attribute__((always_inline)) {
            if constexpr (sizeof(T) == 4)
                store_strip_f32(st, colpart + (int64_t)sl * (3 * VW), lane);
            else
                store_strip<T, W>(st.gc, colpart + (int64_t)sl * (3 * VW), lane);
        };

        // The wave's FIRST unit stands between the kernel's arguments and its first
        // coordinate loads: a cold descriptor load there is one more dependent memory round
        // trip (0.8 us of every launch).  A dense layout (dense_u0 >= 0: this rank's first
        // global unit) has tile t = J (J + 1) / 2 + I in strip-major order (SPEC 3.1,
        // bb_layout_dense_tiles), so that descriptor is arithmetic; the later ones come from
        // the table as before, one unit ahead of their use.
        int2 dc;
        if (dense_u0 >= 0) {
            constexpr int UPT = VW / Lay<T, W>::RPU;
            const unsigned g = (unsigned)dense_u0 + (unsigned)ua;
            const unsigned t = g / UPT, sub = g % UPT;
            unsigned J = (unsigned)((__builtin_sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while (J * (J + 1) / 2 > t) --J;
            while ((J + 1) * (J + 2) / 2 <= t) ++J;
            const unsigned I = t - J * (J + 1) / 2;
            dc = make_int2((int)(I * VW + sub * Lay<T, W>::RPU), (int)(J * VW));
        } else {
            dc = udesc[ua];                                    // current unit
        }
        int2 dn = udesc[ua + 1 < ub ? ua + 1 : ua];            // next unit
        // Prologue: the x rows of the first unit, then its 8 matrix rows.
        // Row coordinates of a unit, one unit ahead.  XRowS: 3*RPU wave-uniform scalars
        // fetched through the scalar cache (X is read-only in this kernel) -- no VMEM
        // slot, no v_readlane (12 floats or 6 doubles: 12 SGPRs, double-buffered).
        // XRowV (fp64 narrow: 24 doubles): one per-lane load, v_readlane per use.
        struct XRowS {
            T v[3 * Lay<T, W>::RPU];
            __device__ __forceinline__ T get(int q) const { return v[q]; }
        };
        struct XRowV {
            T v;
            __device__ __forceinline__ T get(int q) const { return lane_value(v, q); }
        };
        using XRow = typename std::conditional<Lay<T, W>::SCALAR_XROW && !abl::kXrowVector, XRowS,
                                               XRowV>::type;
        auto xrow_load = [&](int i0) __attribute__((always_inline)) {
            XRow x;
            if constexpr (std::is_same<XRow, XRowS>::value) {
                const T *px = X + (int64_t)i0 * 3;
#pragma unroll
                for (int q = 0; q < 3 * Lay<T, W>::RPU; ++q) x.v[q] = px[q];
            } else {
                x.v = load_xrow<T, W>(X, i0, lane);
            }
            return x;
        };
        XRow xr = xrow_load(dc.x);
        using Win = typename std::conditional<abl::kGlobalWindow, WinPtr<Vec>, WinBuf<Vec>>::type;
        // the wave's chunk as a buffer: (ub - ua) units of 8 KiB (< 4 GiB: the host checks)
        const __amdgpu_buffer_rsrc_t win_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T *>(units) + (int64_t)ua * (8192 / (int)sizeof(T)), 0,
            (int)((unsigned)(ub - ua) * 8192u), 0x00020000);
        auto window_of = [&](int u_next) __attribute__((always_inline)) {
            if constexpr (abl::kGlobalWindow) {
                // The refills of the wave's LAST unit are never consumed.  They stay in the
                // loop (a branch around them would cost every unit its exact wait counts),
                // but all lanes ask for the same 16 bytes of the chunk's last unit: 8 lines
                // instead of 8 KiB per wave and launch (1.3-2.7 % of the bytes of a 1/8 share
                // of N=50k), back at once, so the epilogue gets the window's registers early.
                const bool real = u_next < ub;
                return Win{unit_ptr<T>(units, real ? u_next : ub - 1, real ? lane : 0)};
            } else {
                return Win{win_rsrc, (unsigned)(u_next - ua) * 8192u + (unsigned)lane * 16u};
            }
        };
        {
            const Win first = window_of(ua);
#pragma unroll
            for (int r = 0; r < 8; ++r) d[r] = first.template load<NT>(r);
        }
        if constexpr (abl::kWaveTrace) {     // diagnostic build: when does the first data land?
            asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            wave_stamp(stresspart, n_waves_all, w, 5);           // first 1 KiB of the window
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_stamp(stresspart, n_waves_all, w, 6);           // all 8 KiB of it
        }
        // this wave's row partials: 3*RPU elements per unit of its group's chunk
        constexpr unsigned kRowBytes = 3 * Lay<T, W>::RPU * sizeof(T);
        const __amdgpu_buffer_rsrc_t row_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            rowpart + (int64_t)ua * (3 * Lay<T, W>::RPU), 0, (int)((unsigned)(ub - ua) * kRowBytes),
            0x00020000);

        // `first`: the peeled first unit of a strip (a timing experiment may treat it apart)
        // -DBB_UNIT_TRACE: the time at the top of unit k goes to LDS slot k of the wave one
        // unit later (by then it has certainly arrived: no wait of its own), branch-free
        unsigned long long t_prev = 0;
        unsigned long long *ut_lds = reinterpret_cast<unsigned long long *>(
            row_lds + WPB * lds_wave_floats + 8 + 2 * abl::kUnitTraceSlots * wib);
        auto unit_stamp = [&](int k) __attribute__((always_inline)) {
            if constexpr (abl::kUnitTrace) {
                const unsigned long long t_now = (unsigned long long)wall_clock64();
                const int sl = (lane == 0 && k >= 1 && k <= abl::kUnitTraceSlots - 2)
                                   ? k - 1 : abl::kUnitTraceSlots - 1;   // last slot: dummy
                ut_lds[sl] = t_prev;
                t_prev = t_now;
            }
        };
        // fp32: which of a unit's 12 row sums this lane holds -- lanes 0..3 / 16..19 / 32..35
        // after the lane swaps (row_sum3_swap), lanes 48..59 after the DPP tree -- and what
        // follows from it for the unit's store offset and parking slot
        const int f32_sl12 = abl::kSwapRowSum ? swap_rowsum_slot(lane)
                                              : (lane >= 48 && lane < 60 ? lane - 48 : -1);
        const unsigned f32_voff_lane = f32_sl12 >= 0 ? (unsigned)f32_sl12 * 4u : 0x40000000u;
        const int f32_lds_dummy = stage0 + cap_units * 12 + (lane & 3);
        const int f32_lds_real = f32_sl12 >= 0 ? stage0 + f32_sl12 : f32_lds_dummy;
        const int f32_m12 = f32_sl12 >= 0 ? 12 : 0;
        auto unit_step = [&](int u, auto first) __attribute__((always_inline)) {
            unit_stamp(u - ua);
            if constexpr (WPB == 8) {
                // pace keeping (see the kernel's comment): the partner's count was read
                // one unit ago, so nothing here waits on LDS
                const int mine = u - ua;
                if (__builtin_amdgcn_readfirstlane(partner_done) > mine)
                    __builtin_amdgcn_s_setprio(1);
                else
                    __builtin_amdgcn_s_setprio(0);
                __hip_atomic_store(progress + wib, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                partner_done = __hip_atomic_load(progress + (wib ^ 4), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // (the descriptors of the units behind the wave's last one are never used)
            const int un = u + 1 < ub ? u + 1 : u;
            const XRow xrn = xrow_load(dn.x);
            const int2 dnn = udesc[un + 1 < ub ? un + 1 : un];
            unsigned row_voff;
            int stage_slot = 0;
            if constexpr (sizeof(T) == 4) {
                // fp32: lanes 48..59 hold one of the unit's 12 sums each.  Units before
                // park_from are stored directly; the later ones are parked in LDS slot
                // (k - park_from) and their store is dropped (every lane out of range).
                // per-lane parts are loop constants (f32_voff_lane, f32_lds_real / _dummy,
                // f32_m12: below the lambda's captures), per-unit parts scalar: one v_add, one
                // v_cndmask, one v_mad per unit instead of a chain of selects.  A store is out
                // of range (dropped) unless the lane holds a sum AND the unit is not parked:
                // 0x40000000 from either side puts the offset beyond any chunk.
                const int k = u - ua;
                const bool parked = DEFER && k >= park_from;
                row_voff = f32_voff_lane + (parked ? 0x40000000u : (unsigned)k * kRowBytes);
                stage_slot = (parked ? f32_lds_real : f32_lds_dummy) +
                             (parked ? k - park_from : 0) * f32_m12;
            } else if constexpr (W && !abl::kF64Generic) {
                // fp64, 2 x 512 units: lanes 48..53 hold one of the unit's 6 sums each
                // (parking as in fp32; an LDS slot is 4 bytes, a sum takes two)
                const int k = u - ua;
                const bool parked = DEFER && k >= park_from;
                const bool mine = lane >= 48 && lane < 54;
                row_voff = (mine && !parked) ? (unsigned)k * kRowBytes + (unsigned)(lane - 48) * 8u
                                             : kDropOffset;
                stage_slot = stage0 + ((mine && parked) ? (k - park_from) * 12 + (lane - 48) * 2
                                                        : cap_units * 12 + 2 * (lane & 1));
            } else {                        // fp64, 8 x 128: lane 63 stores each row's three sums
                row_voff = lane == 63 ? (unsigned)(u - ua) * kRowBytes : kDropOffset;
            }
            if constexpr (sizeof(T) == 4) {
                float xs12[12];             // scalar registers in the product build
#pragma unroll
                for (int q = 0; q < 12; ++q) xs12[q] = xr.get(q);
                process_unit_f32<NT, OP, DEFER, !(decltype(first)::value && abl::kNoRefillFirst)>(
                    d, xs12, window_of(u + 1), st, stress, row_rsrc, row_voff, stage_slot, sel32);
            }
            else if constexpr (W && !abl::kF64Generic)
                process_unit_f64w<NT, OP, DEFER>(d, xr, window_of(u + 1), st.xj, st.gc,
                                                 sel, stress, row_rsrc, row_voff, stage_slot);
            else
                process_unit<T, W, NT, OP>(d, xr, window_of(u + 1), st.xj, st.gc, stress,
                                       row_rsrc, row_voff);
            xr = xrn;
            dc = dn;
            dn = dnn;
        };
        // Outer loop: one trip per column strip the wave's sweep crosses (rare).
        // Inner loop: the units of that strip, with NO branch in the body.  Its
        // first unit is peeled, so the inner loop is only ever entered from a
        // state with the loop body's own pattern of outstanding loads and
        // stores: hipcc's s_waitcnt counts are static and are merged over every
        // entry of a loop header, and a prologue- or strip-change-shaped entry
        // drains most of the 8-row prefetch window on every iteration.
        if constexpr (abl::kUnitTrace) {
            // slots 5 / 7 of the wave: shader clock (s_memtime) and 100-MHz clock when its
            // loop begins (stored now: nothing is carried through the loop); slot 6 at its end
            if (lane == 0) {
                unsigned long long *t8 = reinterpret_cast<unsigned long long *>(stresspart + n_waves_all);
                t8[(long long)w * 8 + 5] = (unsigned long long)__builtin_amdgcn_s_memtime();
                t8[(long long)w * 8 + 7] = (unsigned long long)wall_clock64();
            }
        }
        int u = ua;
        for (;;) {
            const int curj = dc.y;
            strip_load(curj);
            if constexpr (abl::kWarmup > 0) {
                if (u == ua) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 a0 = {1.f, 1.f}, a1 = a0, a2 = a0, a3 = a0;
                    const f2 m = {0.999f, 1.001f};
                    for (int i = 0; i < abl::kWarmup; ++i)
                        asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n\tv_pk_fma_f32 %1, %1, %4, %4\n\t"
                                     "v_pk_fma_f32 %2, %2, %4, %4\n\tv_pk_fma_f32 %3, %3, %4, %4"
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                                     : "v"(m));
                    asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3));
                }
            }
            if constexpr (abl::kWaveTrace) {
                if (u == ua) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    wave_stamp(stresspart, n_waves_all, w, 7);   // column + row coordinates here
                }
            }
            if constexpr (abl::kPeelFirstUnit || abl::kWaveTrace) {   // (the trace stamps unit 1)
                unit_step(u, std::true_type{});
                if (u == ua) wave_stamp(stresspart, n_waves_all, w, 1);   // first unit done
                ++u;
                while (u < ub && dc.y == curj) {
                    unit_step(u, std::false_type{});
                    ++u;
                }
            } else {
                // ONE copy of the unit body.  The loop is entered with nothing in flight:
                // the strip's coordinates have to be here anyway, and they were asked
                // for after the window -- so the compiler's wait counts inside the loop
                // are those of the back edge alone, which is what round 1 peeled the first
                // iteration for.  Same speed as the peeled form at every size
                // (profiles/archive/r02_peel_ab.txt), a third less code.
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
                do {
                    unit_step(u, std::false_type{});
                    ++u;
                } while (u < ub && dc.y == curj);
            }
            if (u >= ub) break;      // the wave's last strip: its column partial goes out below
            strip_store(slot);
            // the stress of the strip left behind goes with its slot (rare: once per strip a
            // wave crosses), so that every stress partial belongs to ONE strip -- and with
            // several maps in one solver (bb_solver_set_maps) to one map
            {
                const double sv = wave_sum_hi(stress);
                if (lane == 63) stress_slot[slot] = sv;
                stress = 0.0;
            }
            ++slot;
        }
        if constexpr (WPB == 8) {
            // done: the partner stops yielding
            __hip_atomic_store(progress + wib, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_s_setprio(0);
        }
        wave_stamp(stresspart, n_waves_all, w, 2);                    // last unit consumed
        if constexpr (abl::kUnitTrace) {
            if (lane == 0) {
                unsigned long long *t8 = reinterpret_cast<unsigned long long *>(stresspart + n_waves_all);
                t8[(long long)w * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_memtime();
            }
            // slot k = top of unit k, slot n = end of the last unit; out to HBM behind the stamps
            unit_stamp(ub - ua);
            unit_stamp(ub - ua + 1);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            unsigned long long *out = reinterpret_cast<unsigned long long *>(stresspart + n_waves_all * 9) +
                                      (long long)w * abl::kUnitTraceSlots;
            for (int q = lane; q < abl::kUnitTraceSlots - 1; q += 64)
                out[q] = q <= ub - ua && q <= abl::kUnitTraceSlots - 3 ? ut_lds[q] : 0ull;
        }
        if constexpr (DEFER) {
            // the chunk's row sums, 48 bytes per unit in either precision (12 floats or
            // 6 doubles), in one contiguous burst.
            // Lanes read what other lanes of this wave parked: LDS operations of one
            // wave execute in program order; the fence is for the compiler.
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int n4 = ((ub - ua) - park_from) * 3;   // float4 count
            float4 *dst = reinterpret_cast<float4 *>(rowpart + ((int64_t)ua + park_from) *
                                                                   (3 * Lay<T, W>::RPU));
            for (int q = lane; q < n4; q += 64) {
                const float *src = row_lds + stage0 + 4 * q;
                dst[q] = make_float4(src[0], src[1], src[2], src[3]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the region is free again
            __builtin_amdgcn_wave_barrier();
        }
        // the column partial of the wave's LAST strip: into its LDS region (same layout as a
        // slot in HBM), to be added to its neighbours' below
        {
            T *mine = reinterpret_cast<T *>(row_lds + stage0);
            if constexpr (sizeof(T) == 4)
                store_strip_f32(st, mine, lane);
            else
                store_strip<T, W>(st.gc, mine, lane);
        }
    }
    // One column partial per WORKGROUP and strip, not per wave: consecutive waves sweep
    // consecutive chunks, almost always of the same strip, so the 4 or 8 partials of a
    // workgroup are added here, in wave order (fixed), and leave as one slot -- an eighth
    // of the bytes for the sweep to write and for the reduce to read back.  wave_slots[].y
    // names the shared slot; waves of one workgroup that end in the same strip carry the
    // same number (the host deals them, and with BB_WG_COLSUM=0 deals every wave its own).
    // The barrier orders LDS only: __syncthreads() is also a release of the wave's global
    // stores, and waiting here for the acknowledgement of the row-sum burst (s_waitcnt vmcnt
    // in front of s_barrier) kept every wave 1-2 us at the end of every launch.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    {
        constexpr int CH = 3 * VW, NTH = 64 * WPB;
        int k = 0;
        while (k < WPB) {
            const int sl = ws_shared[k];       // wave-uniform (scalar registers)
            if (sl < 0) { ++k; continue; }
            int k2 = k + 1;
            while (k2 < WPB && ws_shared[k2] == sl) ++k2;
            T *dst = colpart + (int64_t)sl * CH;
#pragma unroll
            for (int j = 0; j < (CH + NTH - 1) / NTH; ++j) {
                const int e = (int)threadIdx.x + NTH * j;
                if (CH % NTH == 0 || e < CH) {
                    T acc = T(0);
                    for (int q = k; q < k2; ++q)
                        acc += reinterpret_cast<const T *>(row_lds + q * lds_wave_floats)[e];
                    dst[e] = acc;
                }
            }
            k = k2;
        }
    }

    // per-wave stress: fixed DPP tree, total in lane 63 (no LDS round trips at the very end
    // of the launch: six __shfl_down steps of a double are twelve ds_bpermute)
    stress = wave_sum_hi(stress);
    if (lane == 63) stresspart[w] = stress;
    wave_stamp(stresspart, n_waves_all, w, 3);                        // partials issued
}

// --------------------------------------------------------------------------
// reduce (+ update) kernel: one workgroup per vw-bin block
// --------------------------------------------------------------------------
#ifndef BB_REDUCE_BATCH
#define BB_REDUCE_BATCH 16
#endif
constexpr int kReduceSlice = BB_REDUCE_BATCH;  // chunks per stage-1 slice = chunks loaded per round trip

enum ReduceMode {
    kReduceApply = 0,      // X -= lr * 2 * sum
    kReduceExchange = 1,   // exch = 2 * sum (+ stress hi/lo)
    kReduceStressOnly = 2, // stress only
    kReducePartial = 3,    // stage 1: raw sum of one slice of a block's chunk list
    kReducePeer = 4        // 2 * sum (+ stress hi/lo) stored into this rank's slot on every peer
};

// Peer exchange (bb_solver_peer_*): where this rank's partial goes on each rank
// (its slot in that rank's receive arena, for one parity) and the flag to raise.
constexpr int kMaxPeers = 16;
template <typename T>
struct PeerTable {
    T *dst[kMaxPeers];
    unsigned long long *flag[kMaxPeers];
};
// Health of this rank's peer exchange, in device memory.  `status` is sticky: once a
// wait has failed (time limit, or a peer reported its own failure) every later launch
// of this rank leaves X alone and pushes nothing.  `verdict` is the sequence number of
// the last exchange whose R partials all arrived, `decided` that of the last launch whose
// wait has ended either way: both written by ONE wave (the first of peer_receive_kernel);
// the other workgroups of that launch poll `decided`, then read `verdict`, so a step is
// applied by all of its workgroups or by none.
struct PeerState {
    int status;
    int pad;
    unsigned long long verdict;
    unsigned long long decided;   // sequence number of the last launch whose wait has ended, either way
};
// Flag value a failed rank leaves on every peer: their waits end at once and fail too.
constexpr unsigned long long kPeerPoison = ~0ull;

template <typename T>
struct ReduceParams {
    const T *__restrict__ part;              // rowpart | colpart
    const int64_t *__restrict__ blk_ptr;     // n_blocks + 1
    const int64_t *__restrict__ blk_chunk;   // element offsets into part
    const double *__restrict__ stresspart;
    T *__restrict__ X;                       // apply mode
    T *__restrict__ V;                       // apply mode: velocity (heavy-ball momentum)
    T mu;                                    // momentum coefficient, 0 = plain gradient step
    T scale;                                 // 2 for the gradient (SPEC 2.3), 1 for a matvec
    T *__restrict__ exch;                    // exchange mode: [3*n_pad | hi | lo]
    T *__restrict__ part_out;                // partial mode: CH elements per workgroup
    const PeerTable<T> *__restrict__ peer;   // peer mode: destinations, in device memory
    unsigned *__restrict__ peer_counter;     // peer mode: workgroups done (last one raises flags)
    const PeerState *peer_state;             // peer mode: a failed rank pushes nothing
    unsigned long long seq;                  // peer mode: value the flags take
    int n_peers;
    double *__restrict__ stress_out;         // apply / stress-only: where the stress goes
    int64_t n_pad;
    int n_waves;
    int mode;
    T lr;
    const double *__restrict__ stress_slot;  // per private column slot (see the sweep)
    int n_slots;
    // a step per bin (bb_solver_set_bin_steps / _block_steps; bb_solver_set_maps: the map's):
    // the gradient of bin i leaves the reduce as bin_scale[i] * g_i -- into the update, the
    // exchange buffer or the peers' arenas alike, so every rank and every later kernel steps
    // with the one uniform lr (nullptr for a plain sum: the matvec).  Several maps
    // (world = 1): the stress is folded per map -- map m's partials are
    // map_idx[map_ptr[m] .. map_ptr[m + 1]) (index < n_waves: stresspart, else stress_slot) --
    // into stress_out[m].  n_maps <= 1: one map, every partial, stress_out[0].
    const T *__restrict__ bin_scale;        // n_pad factors (element o of X belongs to bin o / 3)
    // peer exchange: which ranks hold units that touch block b (bit r of peer_mask[b]; the same
    // table on every rank, from the tile list and the partition).  A rank outside a block's
    // mask has nothing but zeros for it: it does not push them, and nobody waits for or adds
    // its slot.  nullptr: every rank, every block (BB_PEER_MASK=0).
    const unsigned *__restrict__ peer_mask;
    int rank;
    const int *__restrict__ map_ptr;
    const int *__restrict__ map_idx;
    int n_maps;
};

// this thread's share of map `m`'s stress partials (stride = threads of the workgroup)
template <typename T>
__device__ __forceinline__ double stress_share(const ReduceParams<T> &p, int m, int tid, int stride) {
    double s = 0.0;
    if (p.n_maps <= 1) {
        for (int i = tid; i < p.n_waves; i += stride) s += p.stresspart[i];
        for (int i = tid; i < p.n_slots; i += stride) s += p.stress_slot[i];
    } else {
        for (int i = p.map_ptr[m] + tid; i < p.map_ptr[m + 1]; i += stride) {
            const int idx = p.map_idx[i];
            s += idx < p.n_waves ? p.stresspart[idx] : p.stress_slot[idx - p.n_waves];
        }
    }
    return s;
}

constexpr int kRedWG = 128;  // threads per reduce workgroup = elements it sums
// Grid: x = the block (or stage-1 slice) of the list, y = which kRedWG of the block's
// 3*vw elements.  One element per thread: a thread's whole slice -- up to kBatch chunks
// -- is in flight before its first add, so a launch costs about one memory round trip,
// and a problem of B blocks puts B * (3*vw/128) workgroups on the chip instead of B
// (N=17,700: 35 blocks used 35 CUs and 18 us for the two stages).
template <typename T, bool W>
__global__ __launch_bounds__(kRedWG) void reduce_kernel(ReduceParams<T> p) {
    constexpr int CH = 3 * Lay<T, W>::VW;
    static_assert(CH % kRedWG == 0, "3*vw is a multiple of the workgroup size");
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int e = (int)blockIdx.y * kRedWG + tid;      // element of the block, < CH
    __shared__ __attribute__((aligned(16))) T push_stage[kRedWG];   // peer mode only
    // peer mode: once this rank's exchange has failed it stops delivering (the status
    // word is only ever written by peer_receive_kernel, i.e. between reduce launches: uniform)
    const bool peer_live =
        p.mode != kReducePeer ||
        __hip_atomic_load(&p.peer_state->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
    if (p.mode != kReduceStressOnly) {
        const int64_t k0 = p.blk_ptr[b], k1 = p.blk_ptr[b + 1];
        T acc = T(0);
        // Chunks are summed in list order (deterministic).
        constexpr int kBatch = kReduceSlice;
        for (int64_t k = k0; k < k1; k += kBatch) {
            T v[kBatch];
#pragma unroll
            for (int q = 0; q < kBatch; ++q) {
                const bool on = k + q < k1;
                v[q] = on ? p.part[p.blk_chunk[on ? k + q : k0] + e] : T(0);
            }
#pragma unroll
            for (int q = 0; q < kBatch; ++q) acc += v[q];
        }
        const int64_t o = (int64_t)b * CH + e;
        if (p.mode == kReducePartial) {
            p.part_out[o] = acc;
        } else {
            const T g = p.bin_scale ? p.bin_scale[o / 3] * (p.scale * acc) : p.scale * acc;
            if (p.mode == kReduceApply) {
                // SPEC 2.4: V <- mu V - lr g ; X <- X + V   (mu = 0: X -= lr g)
                const T v = p.mu * p.V[o] - p.lr * g;
                p.V[o] = v;
                p.X[o] += v;
            } else if (p.mode == kReducePeer) {
                push_stage[tid] = g;
            } else {
                p.exch[o] = g;
            }
        }
        if (p.mode == kReducePeer && peer_live) {
            // the workgroup's values go out as 16-byte stores, one wave instruction per
            // peer: what crosses xGMI is 512-byte (fp32) / 1-KiB (fp64) bursts
            typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));
            constexpr int NV = kRedWG * (int)sizeof(T) / 16;
            __syncthreads();
            if (tid < NV && (p.peer_mask == nullptr || (p.peer_mask[b] >> p.rank & 1u))) {
                const vec_t val = ((const vec_t *)push_stage)[tid];
                for (int q = 0; q < p.n_peers; ++q)
                    ((vec_t *)(p.peer->dst[q] + (int64_t)b * CH + (int64_t)blockIdx.y * kRedWG))[tid] = val;
            }
        }
    }
    if (p.mode == kReducePartial) return;
    if (b < (p.n_maps > 1 ? p.n_maps : 1) && blockIdx.y == 0) {
        __shared__ double sh[kRedWG];
        sh[tid] = stress_share(p, b, tid, kRedWG);
        __syncthreads();
        for (int off = kRedWG / 2; off > 0; off >>= 1) {
            if (tid < off) sh[tid] += sh[tid + off];
            __syncthreads();
        }
        if (tid == 0) {
            const double S = sh[0];
            if (p.mode == kReduceExchange) {
                const T hi = (T)S;
                p.exch[3 * p.n_pad] = hi;
                p.exch[3 * p.n_pad + 1] = (T)(S - (double)hi);
            } else if (p.mode == kReducePeer) {
                const T hi = (T)S, lo = (T)(S - (double)hi);
                for (int q = 0; q < (peer_live ? p.n_peers : 0); ++q) {
                    p.peer->dst[q][3 * p.n_pad] = hi;
                    p.peer->dst[q][3 * p.n_pad + 1] = lo;
                }
            } else {
                p.stress_out[b] = S;
            }
        }
    }
    if (p.mode == kReducePeer && peer_live) {
        // Every workgroup makes its stores visible system-wide and checks in; the
        // last one to do so raises this rank's flag on every peer (release).
        __shared__ int last;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope
        __syncthreads();
        if (tid == 0) last = atomicAdd(p.peer_counter, 1u) == gridDim.x * gridDim.y - 1;
        __syncthreads();
        if (last) {
            if (tid == 0) atomicExch(p.peer_counter, 0u);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (tid < p.n_peers)
                __hip_atomic_store(p.peer->flag[tid], p.seq, __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}


// ---- the reduce in ONE launch of one memory round trip (round 3) ------------------
// reduce_kernel above walks a block's list with one thread per element: a list of L chunks
// is ceil(L / 16) dependent round trips behind three more (kernel arguments -> blk_ptr ->
// blk_chunk), and lists beyond 128 chunks take a second launch.  The kernel trace has that
// chain at 6.2-7.4 us per iteration from N=8,000 to 17,700 and 11-13 us at N=24,926 -- 6 % of
// a 1/8 share of N=50k.  Here a workgroup is 128 elements x S slices of the list (S = 4 or
// 8): every thread has its whole slice -- 16 chunks per trip -- in flight at once, the
// slices meet in LDS and are added in slice order (fixed: bitwise reproducible), and the
// lists come as ONE table of fixed stride (`list_stride` entries per block, padded with the
// offset of a chunk of zeros), so the table's address does not wait for a blk_ptr and no
// load is predicated.  A wave is one slice of 64 elements: its table entries are
// wave-uniform and travel through the scalar cache.
template <typename T, bool W, int S>
__global__ __launch_bounds__(128 * S) void reduce_sliced_kernel(ReduceParams<T> p,
                                                                const int64_t *__restrict__ lists,
                                                                int list_stride) {
    constexpr int CH = 3 * Lay<T, W>::VW;
    static_assert(CH % kRedWG == 0, "3*vw is a multiple of 128");
    const int tid = threadIdx.x;
    const int el = tid & (kRedWG - 1);
    const int sl = __builtin_amdgcn_readfirstlane(tid >> 7);      // slice: uniform per wave
    const int b = blockIdx.x;
    const int e = (int)blockIdx.y * kRedWG + el;                  // element of the block, < CH
    __shared__ __attribute__((aligned(16))) T meet[S][kRedWG];
    const bool peer_live =
        p.mode != kReducePeer ||
        __hip_atomic_load(&p.peer_state->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
    if (p.mode != kReduceStressOnly) {
        const int lps = list_stride / S;                          // a multiple of 16
        const int64_t *list = lists + (int64_t)b * list_stride + (int64_t)sl * lps;
        const int64_t o = (int64_t)b * CH + e;
        T xo = T(0), vo = T(0);
        if (p.mode == kReduceApply && sl == 0) { xo = p.X[o]; vo = p.V[o]; }   // early: independent
        T acc = T(0);
        for (int k = 0; k < lps; k += 16) {
            T v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = p.part[list[k + q] + e];
#pragma unroll
            for (int q = 0; q < 16; ++q) acc += v[q];
        }
        meet[sl][el] = acc;
        __syncthreads();
        if (sl == 0) {
            T tot = meet[0][el];
#pragma unroll
            for (int q = 1; q < S; ++q) tot += meet[q][el];
            const T g = p.bin_scale ? p.bin_scale[o / 3] * (p.scale * tot) : p.scale * tot;
            if (p.mode == kReduceApply) {
                // SPEC 2.4: V <- mu V - lr g ; X <- X + V   (mu = 0: X -= lr g)
                const T v = p.mu * vo - p.lr * g;
                p.V[o] = v;
                p.X[o] = xo + v;
            } else if (p.mode == kReducePeer) {
                meet[0][el] = g;
            } else {
                p.exch[o] = g;
            }
        }
        if (p.mode == kReducePeer && peer_live) {
            typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));
            constexpr int NV = kRedWG * (int)sizeof(T) / 16;
            __syncthreads();
            if (tid < NV && (p.peer_mask == nullptr || (p.peer_mask[b] >> p.rank & 1u))) {
                const vec_t val = ((const vec_t *)meet[0])[tid];
                for (int q = 0; q < p.n_peers; ++q)
                    ((vec_t *)(p.peer->dst[q] + (int64_t)b * CH + (int64_t)blockIdx.y * kRedWG))[tid] = val;
            }
        }
    }
    if (b < (p.n_maps > 1 ? p.n_maps : 1) && blockIdx.y == 0) {
        // the stress: per-wave and per-slot partials of the sweep, fixed tree
        __shared__ double sh[128 * S];
        sh[tid] = stress_share(p, b, tid, 128 * S);
        __syncthreads();
        for (int off = 64 * S; off > 0; off >>= 1) {
            if (tid < off) sh[tid] += sh[tid + off];
            __syncthreads();
        }
        if (tid == 0) {
            const double Sx = sh[0];
            if (p.mode == kReduceExchange) {
                const T hi = (T)Sx;
                p.exch[3 * p.n_pad] = hi;
                p.exch[3 * p.n_pad + 1] = (T)(Sx - (double)hi);
            } else if (p.mode == kReducePeer) {
                const T hi = (T)Sx, lo = (T)(Sx - (double)hi);
                for (int q = 0; q < (peer_live ? p.n_peers : 0); ++q) {
                    p.peer->dst[q][3 * p.n_pad] = hi;
                    p.peer->dst[q][3 * p.n_pad + 1] = lo;
                }
            } else {
                p.stress_out[b] = Sx;
            }
        }
    }
    if (p.mode == kReducePeer && peer_live) {
        // Every workgroup makes its stores visible system-wide and checks in; the last one
        // raises this rank's flag on every peer (release).  ONE lane fences for the
        // workgroup, behind a barrier that every storing wave reaches with its stores
        // acknowledged (__syncthreads() waits vmcnt(0)): a release is a write-back of the
        // XCD's L2, and with all 16 waves of the workgroup issuing one the push cost 33 us
        // instead of 12 (tools/exchange_timing.py; reduce_kernel's workgroups have 2 waves).
        // The wait is spelled out in EVERY storing wave: that __syncthreads() emits one is a
        // code-generation detail (the memory model lets a workgroup-scope release omit it), and
        // a flag that overtakes the data would make a peer sum stale partials silently.  It
        // costs nothing: no cache write-back, and the barrier waits for the slowest wave anyway.
        __shared__ int last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the compiler may drop its own)
            // two levels: the gridDim.y workgroups of a block check in on the block's own
            // counter (its own 128-byte line), the last of them on the top counter -- 12 + 35
            // adds in a row instead of 420 on one word (11-13 ns each) at a 1/8 share of N=50k
            unsigned *mine = p.peer_counter + 32 * (b + 1);
            last = 0;
            if (atomicAdd(mine, 1u) == gridDim.y - 1) {
                atomicExch(mine, 0u);
                last = atomicAdd(p.peer_counter, 1u) == gridDim.x - 1;
            }
        }
        __syncthreads();
        if (last) {
            if (tid == 0) atomicExch(p.peer_counter, 0u);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (tid < p.n_peers)
                __hip_atomic_store(p.peer->flag[tid], p.seq, __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- the peer exchange in ONE launch: reduce, push, wait, sum, update (round 3) --------
// reduce_sliced_kernel in peer mode + peer_receive_kernel are two launches with a check-in of
// every workgroup in between (release fence, two atomic counters, the last arriver raises
// the rank's flag), and the receiving launch has one wave wait for the flags of all ranks
// before any of its workgroups may move: 10 + 6.6 us at a 1/8 share of N=50,000 against
// 5.1 us for the reduce that applies its own sum (kernel trace of tools/exchange_timing.py).
// Here the unit of exchange is what a workgroup reduces anyway -- 128 gradient elements of
// one block -- and the two waves that hold those sums see them through: each thread stores
// its element into its rank's slot on every peer, then reads the same element of every
// rank's slot in its own arena until all of them have ARRIVED, adds them in rank order,
// updates its coordinate and marks the words it consumed as empty again.
//   No flags, no fences: a word says by itself whether it has arrived.  An empty slot word
// holds kPeerEmpty -- all bits set, a NaN no arithmetic produces (the sender turns a sum that
// should ever carry those bits into the canonical NaN) -- and a 4- or 8-byte store is seen
// whole or not at all; nothing else is published with it, so nothing has to be ordered
// against it: relaxed system-scope stores and loads on uncached memory, one round trip from
// "the peer's store has landed" to "the sum is known".  The slots start out empty
// (bb_solver_peer_export), a consumer empties what it has read, and with two parities a
// word is written again only two exchanges later, by a sender that has seen this rank's NEXT
// push -- issued a whole launch after the emptying store.
//   Nobody waits for anybody's slowest workgroup but the owners of the same 64 elements.
// Progress: workgroups are dispatched in index order and each pushes BEFORE it waits, so the
// lowest unfinished index is resident (or next in line) on every rank and has pushed
// wherever it is resident: it completes everywhere, frees its slot, and so on.  That needs
// each rank's GPU to itself (the product's model: one process per GPU); ranks SHARING one
// device can fill its wave slots and LDS with waiting workgroups while the rank they wait for
// still sweeps -- bb_solver_peer_connect keeps the two-launch form then (BB_PEER_FUSED).
//   Failure: a wait that runs into the time limit, a peer's poison word, or this rank's own
// sticky status end the wave without an update; it sets the status, leaves the poison word
// on every peer, and every wave still waiting -- here and there -- leaves within a poll.
// Unlike the two-launch form this one can fail PARTIALLY (some 64-element pieces of the
// step applied, others not): the status is sticky, bb_solver_peer_status reports it, and the
// coordinates of a failed solver are not a result.
template <typename T>
struct PeerTableX {
    T *dst[kMaxPeers];                       // this rank's slot in rank q's arena (one parity)
    unsigned long long *poison[kMaxPeers];   // this rank's poison word in rank q's arena
};

__device__ __forceinline__ bool peer_word_empty(float v) { return __float_as_uint(v) == 0xffffffffu; }
__device__ __forceinline__ bool peer_word_empty(double v) { return __double_as_longlong(v) == -1ll; }
__device__ __forceinline__ float peer_empty_word(float) { return __uint_as_float(0xffffffffu); }
__device__ __forceinline__ double peer_empty_word(double) { return __longlong_as_double(-1ll); }
// what is pushed must not look empty: a sum with exactly those bits becomes the canonical NaN
template <typename T>
__device__ __forceinline__ T peer_sendable(T v) {
    return peer_word_empty(v) ? (T)__builtin_nanf("") : v;
}

// One wave's wait for `n_words` words per lane of every rank: base + r * slot_elems (+ 1 for
// the second word), r < R.  Returns true with the rank-ordered sum(s) in out0 / out1 and the
// words emptied; false after a time-out, a poison word or a set status (status set, peers
// poisoned).  `active` lanes have an element; the others only take part in the votes.
template <typename T, int NW>
__device__ __forceinline__ bool peer_wait_sum(const T *base, bool active, int R, int64_t slot_elems,
                                              const PeerTableX<T> *xt, const unsigned long long *my_poison,
                                              PeerState *state, long long limit, T &sum, double &pair_sum,
                                              unsigned mask = ~0u) {
    const int lane = threadIdx.x & 63;
    T v[NW][kMaxPeers];
    bool ok = false;
    const long long t0 = wall_clock64();
    for (;;) {
        bool missing = false;
#pragma unroll
        for (int r = 0; r < kMaxPeers; ++r) {
            if (r < R) {
                const bool sends = (mask >> r & 1u) != 0;        // (wave-uniform)
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    v[w][r] = (active && sends)
                                  ? __hip_atomic_load(base + (int64_t)r * slot_elems + w, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_SYSTEM)
                                  : T(0);
                    missing |= peer_word_empty(v[w][r]);
                }
            }
        }
        // lanes [0, R): rank `lane`'s poison word in this arena; lane R: this rank's status
        unsigned long long f = 0;
        if (lane < R)
            f = __hip_atomic_load(my_poison + 8 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (lane == R)
            f = (unsigned long long)__hip_atomic_load(&state->status, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
        if (__ballot(lane < R ? f == kPeerPoison : f != 0) != 0) break;
        if (__ballot(missing) == 0) { ok = true; break; }
        if (__ballot(wall_clock64() - t0 > limit) != 0) break;      // wave-uniform exits only
        __builtin_amdgcn_s_sleep(2);
    }
    if (!ok) {
        if (lane < R)
            __hip_atomic_store(xt->poison[lane], kPeerPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (lane == 0)
            __hip_atomic_store(&state->status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    // NW = 1: the element's sum over the ranks, in rank order, in T.  NW = 2: the words are a
    // (hi, lo) pair of one double per rank: their sum in double, rank by rank
    T s0 = T(0);
    double s2 = 0.0;
#pragma unroll
    for (int r = 0; r < kMaxPeers; ++r) {
        if (r < R) {
            s0 += v[0][r];
            if (NW > 1) s2 += (double)v[0][r] + (double)v[NW - 1][r];
            if (active && (mask >> r & 1u)) {
#pragma unroll
                for (int w = 0; w < NW; ++w)
                    __hip_atomic_store(const_cast<T *>(base) + (int64_t)r * slot_elems + w,
                                       peer_empty_word(T(0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    sum = s0;
    pair_sum = s2;
    return true;
}

template <typename T, bool W, int S>
__global__ __launch_bounds__(128 * S) void reduce_exchange_kernel(
    ReduceParams<T> p, const int64_t *__restrict__ lists, int list_stride,
    const PeerTableX<T> *__restrict__ xt, const T *arena, const unsigned long long *my_poison,
    int64_t slot_elems, PeerState *state, long long limit) {
    constexpr int CH = 3 * Lay<T, W>::VW;
    const int tid = threadIdx.x;
    const int el = tid & (kRedWG - 1);
    const int sl = __builtin_amdgcn_readfirstlane(tid >> 7);      // slice: uniform per wave
    const int b = blockIdx.x;
    const int e = (int)blockIdx.y * kRedWG + el;                  // element of the block, < CH
    const int R = p.n_peers;
    __shared__ __attribute__((aligned(16))) T meet[S][kRedWG];
    __shared__ double sh[128 * S];
    // a failed rank pushes nothing and leaves X alone (its peers have its poison word);
    // ONE thread asks, the first barrier below tells the others: the word can change under us
    const int dead_here =
        tid == 0 ? __hip_atomic_load(&state->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;

    const int lps = list_stride / S;                              // a multiple of 16
    const int64_t *list = lists + (int64_t)b * list_stride + (int64_t)sl * lps;
    const int64_t o = (int64_t)b * CH + e;
    T xo = T(0), vo = T(0);
    if (sl == 0) { xo = p.X[o]; vo = p.V[o]; }                    // early: independent
    T acc = T(0);
    for (int k = 0; k < lps; k += 16) {
        T v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = p.part[list[k + q] + e];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += v[q];
    }
    meet[sl][el] = acc;
    const bool first = b == 0 && blockIdx.y == 0;
    if (first)                                // the stress: the sweep's partials, fixed tree
        sh[tid] = stress_share(p, 0, tid, 128 * S);
    if (__syncthreads_or(dead_here) != 0) return;
    if (first) {
        for (int off = 64 * S; off > 0; off >>= 1) {
            if (tid < off) sh[tid] += sh[tid + off];
            __syncthreads();
        }
    }
    if (sl == 0) {
        // the two waves that own the workgroup's 128 elements: push, wait, sum, update
        T tot = meet[0][el];
#pragma unroll
        for (int q = 1; q < S; ++q) tot += meet[q][el];
        const T mine = peer_sendable(p.bin_scale ? p.bin_scale[o / 3] * (p.scale * tot) : p.scale * tot);
        const unsigned mask = p.peer_mask ? p.peer_mask[b] : ~0u;
        if (mask >> p.rank & 1u)
            for (int q = 0; q < R; ++q)
                __hip_atomic_store(xt->dst[q] + o, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        T g;
        double unused;
        if (peer_wait_sum<T, 1>(arena + o, true, R, slot_elems, xt, my_poison, state, limit, g, unused, mask)) {
            // SPEC 2.4: V <- mu V - lr g ; X <- X + V, g = the sum over ranks in rank order
            const T vv = p.mu * vo - p.lr * g;
            p.V[o] = vv;
            p.X[o] = xo + vv;
        }
    } else if (first && tid >= 128 && tid < 192) {
        // the stress travels the same way, as a pair (hi, lo) behind the 3 * n_pad elements:
        // one lane of a wave that has nothing else to do
        const bool lane0 = tid == 128;
        const int64_t n3 = 3 * p.n_pad;
        if (lane0) {
            const double Sx = sh[0];
            const T hi = (T)Sx;
            const T lo = (T)(Sx - (double)hi);
            for (int q = 0; q < R; ++q) {
                __hip_atomic_store(xt->dst[q] + n3, peer_sendable(hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(xt->dst[q] + n3 + 1, peer_sendable(lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        T unused;
        double Sx;
        if (peer_wait_sum<T, 2>(arena + n3, lane0, R, slot_elems, xt, my_poison, state, limit, unused, Sx) && lane0) {
            *p.stress_out = Sx;
            __hip_atomic_store(&state->verdict, p.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Peer exchange, receiving side -- one launch.  Wave 0 of workgroup 0 waits until every
// source rank's flag has reached `seq` and publishes the outcome (state->verdict = seq);
// wave 0 of every other workgroup polls that LOCAL word and the sticky status, so the
// step is applied by all workgroups or by none.  The wait is bounded: past `limit`
// ticks of the constant-rate clock -- or when a peer has left its poison flag -- the
// status word is set, the verdict is withheld, and this rank's own flag on every peer
// is poisoned so that nobody goes on consuming partials computed from coordinates that
// no longer move.  Deciding in one place is what keeps X whole: with every workgroup
// polling the remote flags for itself, some could time out while later ones saw them
// arrive.  Workgroup 0 is dispatched first, so the deciding wave is resident before any
// wave that waits for it; it waits for other devices only.
// Then X <- X + (mu V - lr * sum over ranks, in rank order).  `arena` is this parity's
// first slot; it is uncached memory, read past L1/L2.
template <typename T>
__global__ __launch_bounds__(256) void peer_receive_kernel(
    T *__restrict__ X, T *__restrict__ V, const T *arena, const unsigned long long *flags,
    unsigned long long *const *poison_flags, int world, int64_t slot_elems, int64_t n3, T lr, T mu,
    double *stress_out, unsigned long long seq, PeerState *state, long long limit,
    const unsigned *__restrict__ peer_mask, int ch) {
    const int tid = threadIdx.x;
    __shared__ int go;
    if (tid < 64) {
        bool all_ok = false;
        if (blockIdx.x == 0) {
            const int lane = tid;
            if (__hip_atomic_load(&state->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                // The whole wave polls together until EVERY flag is there: a lane whose
                // flag has arrived keeps reading it, so a poison that lands later is seen.
                const long long t0 = wall_clock64();
                for (;;) {
                    const unsigned long long f =
                        lane < world ? __hip_atomic_load(flags + 8 * lane, __ATOMIC_ACQUIRE,
                                                         __HIP_MEMORY_SCOPE_SYSTEM)
                                     : seq;
                    if (__ballot(f == kPeerPoison) != 0) break;
                    if (__ballot(f >= seq) == __ballot(1)) { all_ok = true; break; }
                    if (__ballot(wall_clock64() - t0 > limit) != 0) break;   // wave-uniform exits only
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            if (all_ok) {
                if (lane == 0)
                    __hip_atomic_store(&state->verdict, seq, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (lane < world)
                    __hip_atomic_store(poison_flags[lane], kPeerPoison, __ATOMIC_RELEASE,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
                if (lane == 0)
                    __hip_atomic_store(&state->status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // one word says "this launch's wait is over": what the other workgroups poll
            if (lane == 0)
                __hip_atomic_store(&state->decided, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // ends when workgroup 0 has decided, one way or the other (it always does:
            // its own wait is bounded)
            // (relaxed polls of ONE word, well apart, and ONE acquire fence at the end: an
            // acquire is a cache invalidate on this part -- 200 waves issuing one per poll
            // cost 12 us -- and a thousand waves hammering one line leave little of its
            // memory channel to the ranks that share the device in a rehearsal)
            while (__hip_atomic_load(&state->decided, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq)
                __builtin_amdgcn_s_sleep(32);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            all_ok = __hip_atomic_load(&state->verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq;
        }
        if (tid == 0) go = all_ok ? 1 : 0;
    }
    __syncthreads();
    if (!go) return;
    // The arena is read with system-scope loads, which go past L1/L2 to memory: issued
    // after the barrier they cannot be older than the flags wave 0 acquired.
    // The grid is capped (kPeerReceiveWGs workgroups, a thread takes several elements): a
    // grid the size of the update would fill every wave slot of the device with waves that
    // spin, and when several ranks share one GPU -- the test box, a rehearsal -- the kernel of
    // the rank they wait for could then not start at all (found by the four-rank test at
    // N=309,568: 3,629 workgroups waiting, nobody delivering, time-out).
    constexpr int EPT = 4;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t base = (int64_t)blockIdx.x * 256 + tid; base < n3; base += EPT * stride) {
        // all slots of all of the thread's elements are requested before the first add (one
        // memory round trip per eight ranks, not one per rank and element); the sum of an
        // element runs in rank order
        T g[EPT];
#pragma unroll
        for (int q = 0; q < EPT; ++q) g[q] = T(0);
        unsigned mk[EPT];                       // the ranks that sent something for the element's block
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int64_t e = base + q * stride;
            mk[q] = (peer_mask != nullptr && e < n3) ? peer_mask[e / ch] : ~0u;
        }
        for (int r0 = 0; r0 < world; r0 += 8) {
            T v[EPT][8];
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const int64_t e = base + q * stride;
#pragma unroll
                for (int p = 0; p < 8; ++p)
                    v[q][p] = (e < n3 && r0 + p < world && (mk[q] >> (r0 + p) & 1u))
                                  ? __hip_atomic_load(arena + (r0 + p) * slot_elems + e,
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                  : T(0);
            }
#pragma unroll
            for (int q = 0; q < EPT; ++q)
#pragma unroll
                for (int p = 0; p < 8; ++p)
                    if (r0 + p < world) g[q] += v[q][p];
        }
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int64_t e = base + q * stride;
            if (e < n3) {
                const T vv = mu * V[e] - lr * g[q];
                V[e] = vv;
                X[e] += vv;
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        double S = 0.0;
        for (int r = 0; r < world; ++r)
            S += (double)__hip_atomic_load(arena + r * slot_elems + n3, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM) +
                 (double)__hip_atomic_load(arena + r * slot_elems + n3 + 1, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
        *stress_out = S;
    }
}
constexpr unsigned kPeerReceiveWGs = 256;

template <typename T>
__global__ __launch_bounds__(256) void apply_kernel(T *__restrict__ X, T *__restrict__ V,
                                                    const T *__restrict__ exch, int64_t n3, T lr,
                                                    T mu, double *stress_out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n3) {
        const T v = mu * V[e] - lr * exch[e];
        V[e] = v;
        X[e] += v;
    }
    if (e == 0 && stress_out) *stress_out = (double)exch[n3] + (double)exch[n3 + 1];
}

// --------------------------------------------------------------------------
// packing kernels
// --------------------------------------------------------------------------
// staged fp64 rows (row-major, ld = VW) -> units of one run of tiles.
template <typename T, bool W>
__global__ __launch_bounds__(256) void convert_units_kernel(
    const double *__restrict__ stage, T *__restrict__ units_out, const int2 *__restrict__ udesc,
    int64_t ul0, int64_t stage_row0 /* global row of stage row 0 */, int64_t n_bins, int kind,
    double neg_inv_alpha) {
    constexpr int VW = Lay<T, W>::VW;
    const int64_t ul = ul0 + blockIdx.x;
    const int2 dsc = udesc[ul];
    constexpr int RPU = Lay<T, W>::RPU;
    T *out = units_out + ul * (RPU * VW);
    for (int e = threadIdx.x; e < RPU * VW; e += 256) {
        const int r = e / VW, c = e % VW;
        const int64_t i = (int64_t)dsc.x + r, j = (int64_t)dsc.y + c;
        double v = 0.0;
        if (j > i && j < n_bins) {
            v = stage[(i - stage_row0) * VW + c];
            const bool ok = (v > 0.0) && (v <= 1.7976931348623157e308);  // finite, positive
            if (!ok)
                v = 0.0;
            else if (kind == BB_KIND_COUNTS)
                v = pow(v, neg_inv_alpha);
        }
        // fp32: the kernel's 0/1 weight needs delta >= 2^-100; anything that small
        // is below the distance clamp eps = 1e-15 anyway and is stored as "none"
        if (v < wish_floor<T>()) v = 0.0;
        out[e] = (T)v;
    }
}

// A (d, d) float64 matrix that is ALREADY in HBM (a device-resident ContactMap,
// bb_cm_*) -> this rank's units, device to device: the same conversion as
// convert_units_kernel without the staging copy.  Only elements j > i are read.
template <typename T, bool W>
__global__ __launch_bounds__(256) void pack_units_from_matrix_kernel(
    const double *__restrict__ m, int64_t ld, T *__restrict__ units_out,
    const int2 *__restrict__ udesc, int64_t n_bins, int kind, double neg_inv_alpha, int64_t off) {
    // (off > 0: m is the matrix of the bins [off, n_bins) of a solver of several maps; units of
    // other maps are left alone)
    constexpr int VW = Lay<T, W>::VW, RPU = Lay<T, W>::RPU;
    const int64_t ul = blockIdx.x;
    const int2 dsc = udesc[ul];
    if (dsc.y < off || dsc.y >= n_bins || dsc.x < off) return;
    T *out = units_out + ul * (RPU * VW);
    for (int e = threadIdx.x; e < RPU * VW; e += 256) {
        const int r = e / VW, c = e % VW;
        const int64_t i = (int64_t)dsc.x + r, j = (int64_t)dsc.y + c;
        double v = 0.0;
        if (j > i && j < n_bins) {
            v = m[(i - off) * ld + (j - off)];
            const bool ok = (v > 0.0) && (v <= 1.7976931348623157e308);  // finite, positive
            if (!ok)
                v = 0.0;
            else if (kind == BB_KIND_COUNTS)
                v = pow(v, neg_inv_alpha);
        }
        if (v < wish_floor<T>()) v = 0.0;
        out[e] = (T)v;
    }
}

// Sparse (i, j, value) entries -> resident units (blocked-sparse input).  The
// units were zeroed ("no constraint") first.  tilemap[I * n_blocks + J] is the
// tile's index in the global list or -1.
//
// A bin pair that occurs more than once keeps its LAST entry, as in the reference's
// scatter (`matrix[j*d+k] = ...` in file order, blueberry/datatypes.pyx:110-116) and
// in bb_contactmap_scatter.  Three passes over the entries, nothing the size of the
// matrix: `phase` 0 clears every cell an entry names, 1 records the highest entry
// index + 1 per cell (integer atomicMax on the cell's own bits, which the clear left
// at 0), 2 lets that entry alone store its value.
// Where the entries come from: (rows, cols, vals) arrays, or -- `tr` set -- the (n, 3) triples
// [pos_i, pos_j, count] of a Rao-format file as ContactMap.__init__ reads them
// (blueberry/datatypes.pyx:100-113), resident on the device: numpy.nan_to_num applied to each
// value as it is read (pyx:102), bin = (int)(pos / resolution) (pyx:111-112), element (t, c)
// at tr[t * st + c * sc] (row-major rows or the reference's column-major array).
struct EntrySrc {
    const int64_t *rows, *cols;
    const double *vals;
    const double *tr;
    int64_t st, sc;
    double resolution;
};
__device__ __forceinline__ double nan_to_num_f64(double v) {
    return v != v ? 0.0 : (v > 1.7976931348623157e308 ? 1.7976931348623157e308
                                                      : (v < -1.7976931348623157e308 ? -1.7976931348623157e308 : v));
}
// the entry's bin pair; false: a position outside what an int can hold
__device__ __forceinline__ bool entry_bins(const EntrySrc &src, int64_t k, int64_t &i, int64_t &j) {
    if (src.tr == nullptr) { i = src.rows[k]; j = src.cols[k]; return true; }
    const double qi = nan_to_num_f64(src.tr[k * src.st]) / src.resolution,
                 qj = nan_to_num_f64(src.tr[k * src.st + src.sc]) / src.resolution;
    if (!(qi > -2147483648.0 && qi < 2147483648.0 && qj > -2147483648.0 && qj < 2147483648.0))
        return false;
    i = (int)qi; j = (int)qj;
    return true;
}

// Which tiles the entries name: present[I * n_blocks + J] = 1 (I <= J), so that the host can
// write down the tile list of a blocked-sparse solver without binning the entries itself.
__global__ __launch_bounds__(256) void entries_tiles_kernel(EntrySrc src, int64_t nnz, int64_t vw,
                                                            int64_t n_blocks, int64_t n_bins,
                                                            unsigned char *__restrict__ present,
                                                            int *__restrict__ bad) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nnz) return;
    int64_t i, j;
    if (!entry_bins(src, k, i, j)) { atomicExch(bad, 1); return; }
    if (i == j) return;
    if (i > j) { const int64_t t = i; i = j; j = t; }
    if (i < 0 || j >= n_bins) { atomicExch(bad, 1); return; }
    present[(i / vw) * n_blocks + j / vw] = 1;
}

template <typename T, bool W>
__global__ __launch_bounds__(256) void scatter_entries_kernel(
    EntrySrc src, int64_t nnz, const int32_t *__restrict__ tilemap,
    int64_t n_blocks, int64_t n_bins, int64_t u_begin, int64_t u_end, T *__restrict__ units,
    int kind, double neg_inv_alpha, const double *__restrict__ kr,
    const double *__restrict__ krexp, int *__restrict__ bad, int phase) {
    constexpr int VW = Lay<T, W>::VW, RPU = Lay<T, W>::RPU, UPT = VW / RPU;
    using Bits = typename std::conditional<sizeof(T) == 4, unsigned int, unsigned long long>::type;
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nnz) return;
    int64_t i, j;
    if (!entry_bins(src, k, i, j)) { atomicExch(bad, 1); return; }
    if (i == j) return;                     // the diagonal carries no pair
    if (i > j) { const int64_t t = i; i = j; j = t; }
    if (i < 0 || j >= n_bins) { atomicExch(bad, 1); return; }
    const int64_t I = i / VW, J = j / VW;
    const int32_t t = tilemap[I * n_blocks + J];
    if (t < 0) { atomicExch(bad, 2); return; }   // entry outside the tile list
    const int64_t ri = i - I * VW;
    const int64_t u = (int64_t)t * UPT + ri / RPU;
    if (u < u_begin || u >= u_end) return;  // another rank's unit
    T *cell = units + (u - u_begin) * (RPU * VW) + (ri % RPU) * VW + (j - J * VW);
    if (phase == 0) {
        *reinterpret_cast<Bits *>(cell) = 0;
        return;
    }
    if (phase == 1) {
        atomicMax(reinterpret_cast<Bits *>(cell), (Bits)(k + 1));
        return;
    }
    if (*reinterpret_cast<const Bits *>(cell) != (Bits)(k + 1)) return;   // a later entry won
    double v = src.tr ? nan_to_num_f64(src.tr[k * src.st + 2 * src.sc]) : src.vals[k];
    // KR balancing + observed/expected, the element-wise form of the loop at
    // reference datatypes.pyx:166-169 (same operation order)
    // (... followed by the reference's nan_to_num over the matrix, pyx:171: a NaN quotient is 0 =
    // no constraint, an overflowing one the largest double -- as ContactMap.normalize leaves it)
    if (kr != nullptr) v = nan_to_num_f64(v / (kr[i] * kr[j] * krexp[j - i]));
    const bool ok = (v > 0.0) && (v <= 1.7976931348623157e308);
    if (!ok)
        v = 0.0;
    else if (kind == BB_KIND_COUNTS)
        v = pow(v, neg_inv_alpha);
    if (v < wish_floor<T>()) v = 0.0;
    *cell = (T)v;
}

// delta_ij = |x*_i - x*_j| generated in place (synthetic inputs).
template <typename T, bool W>
__global__ __launch_bounds__(256) void gen_units_kernel(const double *__restrict__ xs,
                                                        T *__restrict__ units_out,
                                                        const int2 *__restrict__ udesc,
                                                        int64_t n_bins) {
    constexpr int VW = Lay<T, W>::VW;
    const int64_t ul = blockIdx.x;
    const int2 dsc = udesc[ul];
    constexpr int RPU = Lay<T, W>::RPU;
    T *out = units_out + ul * (RPU * VW);
    for (int e = threadIdx.x; e < RPU * VW; e += 256) {
        const int r = e / VW, c = e % VW;
        const int64_t i = (int64_t)dsc.x + r, j = (int64_t)dsc.y + c;
        double v = 0.0;
        if (j > i && j < n_bins) {
            const double dx = xs[3 * i] - xs[3 * j], dy = xs[3 * i + 1] - xs[3 * j + 1],
                         dz = xs[3 * i + 2] - xs[3 * j + 2];
            v = sqrt(dx * dx + dy * dy + dz * dz);
        }
        if (v < wish_floor<T>()) v = 0.0;
        out[e] = (T)v;
    }
}

// --------------------------------------------------------------------------
// row-owner path: small maps, one launch per iteration
// --------------------------------------------------------------------------
// For a map whose whole symmetric matrix stays in the caches (N up to a few thousand
// bins: chr21 at 50 kb is 963) the unit sweep above is launch-bound -- three dependent
// launches per iteration, two of them only to add up partial sums.  Here BOTH
// triangles are stored (`full`, n rows of `ld` elements, ld a multiple of 128, zero
// = no constraint, zero diagonal) and one wave owns one bin i: it reads row i and the
// coordinates X_k of every bin, so g_i is complete inside the wave -- no partial sums,
// no reduce launch, no atomics, and a fixed summation order.  X_{k+1} goes to a second
// buffer (the other waves are still reading X_k), so one launch is one iteration.
// Every pair is evaluated twice (once from each end), which at these sizes is cheaper
// than the two extra launches it saves.
//
// Stress: S(X_k) = 1/2 sum_i sum_j res_ij^2.  Each workgroup leaves the sum over its
// four rows in part_out[block]; the fold is done by ONE extra workgroup of the NEXT
// launch (blockIdx = gridDim - 1; it runs beside that launch's row workgroups) and by a
// fold-only launch after the last iteration of a bb_solver_iterate call.
template <typename T>
__device__ __forceinline__ void pair_row(T delta, T xi, T yi, T zi, T xj, T yj, T zj, T &gx,
                                         T &gy, T &gz, T &s) {
    const T dx = xi - xj, dy = yi - yj, dz = zi - zj;
    const T d2 = fma(dx, dx, fma(dy, dy, fma(dz, dz, Traits<T>::eps2())));  // SPEC 2.2
    T rinv, dist;
    if constexpr (sizeof(T) == 4) {
        rinv = __builtin_amdgcn_rsqf(d2);
        dist = d2 * rinv;
    } else {
        rsqrt_sqrt_f64(d2, rinv, dist);      // as pair_step<double>
    }
    T res;
    if constexpr (sizeof(T) == 4)
        res = delta > T(0) ? dist - delta : T(0);
    else
        res = fma(dist, weight01_f64(delta), -delta);        // as pair_step<double>
    s = fma(res, res, s);
    const T coef = res * rinv;
    gx = fma(coef, dx, gx);
    gy = fma(coef, dy, gy);
    gz = fma(coef, dz, gz);
}

// columns per loop trip of a wave: one 16-byte load per lane = 4 (fp32) / 2 (fp64) columns.
// Round 4: per pair the kernel issued one 4-byte (8-byte) load of the matrix and three scalar
// loads of the partner's coordinates.  A lane now takes 4 (2) CONSECUTIVE columns: one 16-byte
// load of the row and three 16-byte loads of the 12 (6) contiguous coordinates of those
// columns -- one load per pair (two in fp64) instead of four: fp64 -9 % at N=963, -12 % at
// 2,500, -19 % at 4,096 (28.4 -> 23.0 us); fp32 -4 % at 2,500, -8 % at 4,096
// (profiles/r04_row_owner_ab.txt; -DBB_ROW_OWNER_SCALAR keeps the old trip for the A/B).
// Packed fp32 pair math on top of it (v_pk_* straight on the AoS register pairs) changed
// nothing: from N ~ 4,000 the kernel moves both triangles at ~5 TB/s and that is its bound.
#ifdef BB_ROW_OWNER_SCALAR
constexpr int kRowTrip = 128;
template <typename T> struct RowTrip { static constexpr int COLS = 128; };
#else
constexpr int kRowTrip = 256;   // (the larger of the two: what `ld` is rounded up to)
template <typename T> struct RowTrip { static constexpr int COLS = 64 * (16 / (int)sizeof(T)); };
#endif

// WPR waves share one row (1, 2 or 4: the host picks it so that a small map still
// puts >= 16 waves on every CU): wave part p takes the 128-column trips p, p + WPR, ...;
// the parts meet in LDS and are added in part order by the row's first wave.
template <typename T, int WPR>
__global__ __launch_bounds__(256) void row_owner_kernel(
    const T *__restrict__ full, int64_t ld, int n, const T *__restrict__ Xin,
    T *__restrict__ Xout, T *__restrict__ V, T lr, T mu, const double *__restrict__ part_prev,
    int n_prev, double *__restrict__ hist_prev, double *__restrict__ part_out,
    const T *__restrict__ bin_scale) {
    constexpr int ROWS = 4 / WPR;                 // rows per workgroup
    __shared__ double sh[256];
    __shared__ T red[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (blockIdx.x == gridDim.x - 1) {
        // the fold workgroup: stress of the PREVIOUS launch, fixed order
        if (hist_prev == nullptr) return;
        double a = 0.0;
        for (int q = tid; q < n_prev; q += 256) a += part_prev[q];
        sh[tid] = a;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) sh[tid] += sh[tid + off];
            __syncthreads();
        }
        if (tid == 0) *hist_prev = 0.5 * sh[0];
        return;
    }
    const int i = blockIdx.x * ROWS + wv / WPR;
    const int part = wv % WPR;
    T s = T(0), gx = T(0), gy = T(0), gz = T(0);
    T xi = T(0), yi = T(0), zi = T(0);
    if (i < n) {
        xi = Xin[3 * (int64_t)i]; yi = Xin[3 * (int64_t)i + 1]; zi = Xin[3 * (int64_t)i + 2];
        const T *row = full + (int64_t)i * ld;
        constexpr int COLS = RowTrip<T>::COLS;
        const int trips = ((int)(ld / COLS) - part + WPR - 1) / WPR;   // trips part, part + WPR, ...
        auto trip = [&](int tr) __attribute__((always_inline)) {
#ifdef BB_ROW_OWNER_SCALAR
            const int c = (part + tr * WPR) * COLS;
#pragma unroll
            for (int u = 0; u < COLS / 64; ++u) {
                const int j = c + 64 * u + lane;
                const T *xj = Xin + 3 * (int64_t)j;
                pair_row<T>(row[j], xi, yi, zi, xj[0], xj[1], xj[2], gx, gy, gz, s);
            }
#else
            using Vec = typename Traits<T>::Vec;
            const int c = (part + tr * WPR) * COLS + lane * Traits<T>::VPL;
            const Vec dv = *reinterpret_cast<const Vec *>(row + c);
            const Vec *px = reinterpret_cast<const Vec *>(Xin + 3 * (int64_t)c);
            const Vec a = px[0], b = px[1], q = px[2];
            if constexpr (sizeof(T) == 4) {      // x0 y0 z0 x1 | y1 z1 x2 y2 | z2 x3 y3 z3
                pair_row<T>(dv.x, xi, yi, zi, a.x, a.y, a.z, gx, gy, gz, s);
                pair_row<T>(dv.y, xi, yi, zi, a.w, b.x, b.y, gx, gy, gz, s);
                pair_row<T>(dv.z, xi, yi, zi, b.z, b.w, q.x, gx, gy, gz, s);
                pair_row<T>(dv.w, xi, yi, zi, q.y, q.z, q.w, gx, gy, gz, s);
            } else {                             // x0 y0 | z0 x1 | y1 z1
                pair_row<T>(dv.x, xi, yi, zi, a.x, a.y, b.x, gx, gy, gz, s);
                pair_row<T>(dv.y, xi, yi, zi, b.y, q.x, q.y, gx, gy, gz, s);
            }
#endif
        };
        // fp32: 4 trips (8 pairs per lane) in flight; fp64: unrolling costs more in
        // registers than it hides (N=2,500: 18.3 us per iteration unrolled, 14.4 rolled)
        if constexpr (sizeof(T) == 4) {
#pragma unroll 4
            for (int tr = 0; tr < trips; ++tr) trip(tr);
        } else {
#pragma nounroll
            for (int tr = 0; tr < trips; ++tr) trip(tr);
        }
        wave_sum_hi3(gx, gy, gz);
        s = wave_sum_hi(s);
    }
    if constexpr (WPR > 1) {
        if (lane == 63) { red[wv][0] = gx; red[wv][1] = gy; red[wv][2] = gz; red[wv][3] = s; }
        __syncthreads();
        if (part == 0 && lane == 63) {
#pragma unroll
            for (int p = 1; p < WPR; ++p) {
                gx += red[wv + p][0]; gy += red[wv + p][1]; gz += red[wv + p][2];
                s += red[wv + p][3];
            }
        }
    }
    if (i < n && part == 0 && lane == 63) {
        // SPEC 2.3 / 2.4: g = 2 * sum (2.4.1: times the bin's factor); V <- mu V - lr g; X <- X + V
        const int64_t o = 3 * (int64_t)i;
        T g0 = T(2) * gx, g1 = T(2) * gy, g2 = T(2) * gz;
        if (bin_scale) {
            const T c = bin_scale[i];
            g0 = c * g0; g1 = c * g1; g2 = c * g2;
        }
        const T vx = mu * V[o] - lr * g0;
        const T vy = mu * V[o + 1] - lr * g1;
        const T vz = mu * V[o + 2] - lr * g2;
        V[o] = vx; V[o + 1] = vy; V[o + 2] = vz;
        Xout[o] = xi + vx; Xout[o + 1] = yi + vy; Xout[o + 2] = zi + vz;
    }
    if (lane == 63) sh[wv] = (part == 0 && i < n) ? (double)s : 0.0;
    __syncthreads();
    if (tid == 0) part_out[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// resident units -> both triangles of the full matrix (zeroed first; one-off)
template <typename T, bool W>
__global__ __launch_bounds__(256) void units_to_full_kernel(const T *__restrict__ units,
                                                            const int2 *__restrict__ udesc,
                                                            T *__restrict__ full, int64_t ld,
                                                            int64_t n_bins) {
    constexpr int VW = Lay<T, W>::VW, RPU = Lay<T, W>::RPU;
    const int64_t ul = blockIdx.x;
    const int2 dsc = udesc[ul];
    const T *in = units + ul * (RPU * VW);
    for (int e = threadIdx.x; e < RPU * VW; e += 256) {
        const int r = e / VW, c = e % VW;
        const int64_t i = (int64_t)dsc.x + r, j = (int64_t)dsc.y + c;
        if (j > i && j < n_bins) {
            const T v = in[e];
            full[i * ld + j] = v;
            full[j * ld + i] = v;
        }
    }
}

// Per bin: how many of this rank's stored pairs constrain it (delta > 0) -- the degrees a step
// per bin is made from (bb_solver_degrees; once per map, not per iteration).  A workgroup takes
// kDegUnits consecutive units: a row's count is a ballot per wave (one atomic per wave and
// row), a column's is kept by the thread that owns the column for as long as the units stay
// in one tile column (one atomic per column and run of units).
constexpr int kDegUnits = 32;
template <typename T, bool W>
__global__ __launch_bounds__(256) void unit_degrees_kernel(const T *__restrict__ units,
                                                           const int2 *__restrict__ udesc,
                                                           int64_t n_local, int64_t n_bins,
                                                           int *__restrict__ deg) {
    constexpr int VW = Lay<T, W>::VW, RPU = Lay<T, W>::RPU, CPT = (VW + 255) / 256;
    const int lane = threadIdx.x & 63;
    const int64_t u0 = (int64_t)blockIdx.x * kDegUnits;
    const int64_t u1 = u0 + kDegUnits < n_local ? u0 + kDegUnits : n_local;
    int col[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) col[k] = 0;
    int j0 = udesc[u0].y;
    auto flush = [&]() {
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int c = (int)threadIdx.x + 256 * k;
            if (c < VW && col[k] > 0) atomicAdd(deg + j0 + c, col[k]);
            col[k] = 0;
        }
    };
    for (int64_t ul = u0; ul < u1; ++ul) {
        const int2 dsc = udesc[ul];
        if (dsc.y != j0) {                              // (uniform: every thread reads the same word)
            flush();
            j0 = dsc.y;
        }
        const T *in = units + ul * (RPU * VW);
        for (int r = 0; r < RPU; ++r) {
            const int64_t i = (int64_t)dsc.x + r;
            int row = 0;
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int c = (int)threadIdx.x + 256 * k;
                const int64_t j = (int64_t)dsc.y + c;
                const bool on = c < VW && j > i && j < n_bins && in[r * VW + c] > T(0);
                col[k] += on ? 1 : 0;
                row += __popcll(__ballot(on));
            }
            if (lane == 0 && row > 0) atomicAdd(deg + i, row);
        }
    }
    flush();
}

// Measurement only: the same waves read the same units with the same rolling
// 8-row window, but do nothing with the data except fold it into a checksum --
// the practical HBM read ceiling for this access pattern on this box.
template <bool NT>
__global__ __launch_bounds__(256, 4) void stream_read_kernel(const float4 *__restrict__ units,
                                                             int chunk_q, int chunk_r,
                                                             float *__restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int ua = w * chunk_q + (w < chunk_r ? w : chunk_r);
    const int ub = ua + chunk_q + (w < chunk_r ? 1 : 0);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ua < ub) {
        float4 d[8];
        const float4 *first = units + (int64_t)ua * 512 + lane;
#pragma unroll
        for (int r = 0; r < 8; ++r) d[r] = stream_load<NT>(first + r * 64);
        for (int u = ua; u < ub; ++u) {
            const int un = u + 1 < ub ? u + 1 : u;
            const float4 *next = units + (int64_t)un * 512 + lane;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                acc.x += d[r].x; acc.y += d[r].y; acc.z += d[r].z; acc.w += d[r].w;
                d[r] = stream_load<NT>(next + r * 64);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[w] = acc.x;  // keep the loads alive
}

// --------------------------------------------------------------------------
// spectral start on the device (bb_solver_spectral_init): the N x 3 side of the block
// power iteration.  All of it is O(N): one workgroup, fixed-order sums.
// --------------------------------------------------------------------------
// out[0..8] = A^T B (3 x 3) and out[9..11] = column sums of A, over n rows of (n,3) arrays
template <typename TA, typename TB>
__global__ __launch_bounds__(1024) void gram3_kernel(const TA *__restrict__ A,
                                                     const TB *__restrict__ B, int64_t n,
                                                     double *__restrict__ out) {
    __shared__ double sh[12][16];
    double acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const double a0 = (double)A[3 * i], a1 = (double)A[3 * i + 1], a2 = (double)A[3 * i + 2];
        const double b0 = (double)B[3 * i], b1 = (double)B[3 * i + 1], b2 = (double)B[3 * i + 2];
        acc[0] = fma(a0, b0, acc[0]); acc[1] = fma(a0, b1, acc[1]); acc[2] = fma(a0, b2, acc[2]);
        acc[3] = fma(a1, b0, acc[3]); acc[4] = fma(a1, b1, acc[4]); acc[5] = fma(a1, b2, acc[5]);
        acc[6] = fma(a2, b0, acc[6]); acc[7] = fma(a2, b1, acc[7]); acc[8] = fma(a2, b2, acc[8]);
        acc[9] += a0; acc[10] += a1; acc[11] += a2;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        double v = acc[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) sh[q][wv] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        double v = 0.0;
        for (int k = 0; k < 16; ++k) v += sh[threadIdx.x][k];
        out[threadIdx.x] = v;
    }
}

// out_i = scale * ((in_i - mean) M), rows i < n_bins; M is 3 x 3 row-major; rows beyond
// n_bins (padding) are written as 0.  The 12 numbers travel as kernel arguments.
struct Affine3 {
    double mean[3], m[9], scale;
};
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void affine3_kernel(const TI *__restrict__ in, TO *__restrict__ out,
                                                      int64_t n_bins, int64_t n_pad, Affine3 a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pad) return;
    double o0 = 0.0, o1 = 0.0, o2 = 0.0;
    if (i < n_bins) {
        const double x0 = (double)in[3 * i] - a.mean[0], x1 = (double)in[3 * i + 1] - a.mean[1],
                     x2 = (double)in[3 * i + 2] - a.mean[2];
        o0 = a.scale * (x0 * a.m[0] + x1 * a.m[3] + x2 * a.m[6]);
        o1 = a.scale * (x0 * a.m[1] + x1 * a.m[4] + x2 * a.m[7]);
        o2 = a.scale * (x0 * a.m[2] + x1 * a.m[5] + x2 * a.m[8]);
    }
    out[3 * i] = (TO)o0; out[3 * i + 1] = (TO)o1; out[3 * i + 2] = (TO)o2;
}

// ---- the same N x 3 algebra WITHOUT a host round trip (round 4) -------------------------
// bb_solver_spectral_init used to read 12 sums back four times per product (centre, centre,
// two Cholesky-QR passes): four stream synchronisations around a sweep that takes 0.1 ms on a
// 1/8 share.  Now the 3 x 3 work stays on the device: every pass over an (n,3) array is ONE
// kernel that leaves the 12 sums of its OUTPUT (Gram matrix + column sums) as per-workgroup
// partials, and the NEXT pass adds them in a fixed order and does the 3 x 3 step (mean, or
// Cholesky factor and its inverse) in its own prologue.  A factor that is not positive definite
// (the iterate lost rank) raises a flag the host reads once, at the end.  Sums are in double,
// fixed order: bitwise reproducible, and identical on every rank of a multi-rank start.
constexpr int kSpWG = 256;          // threads per workgroup of the passes
constexpr int kSpMaxGroups = 256;   // partials per pass

// rows [r0, r1) of workgroup g of G over n_pad rows
__device__ __forceinline__ void sp_rows(int64_t n_pad, int64_t &r0, int64_t &r1) {
    const int64_t per = (n_pad + gridDim.x - 1) / gridDim.x;
    r0 = (int64_t)blockIdx.x * per;
    r1 = r0 + per < n_pad ? r0 + per : n_pad;
}
// the workgroup's 12 sums -> partial[blockIdx.x * 12 ..]: lanes by shuffle, waves through LDS
__device__ __forceinline__ void sp_store_partial(double (&acc)[12], double *__restrict__ partial) {
    __shared__ double sh[12][kSpWG / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        double v = acc[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) sh[q][wv] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < kSpWG / 64; ++k) v += sh[threadIdx.x][k];
        partial[(int64_t)blockIdx.x * 12 + threadIdx.x] = v;
    }
}
__device__ __forceinline__ void sp_accumulate(double (&acc)[12], double a0, double a1, double a2) {
    acc[0] = fma(a0, a0, acc[0]); acc[1] = fma(a0, a1, acc[1]); acc[2] = fma(a0, a2, acc[2]);
    acc[3] = fma(a1, a0, acc[3]); acc[4] = fma(a1, a1, acc[4]); acc[5] = fma(a1, a2, acc[5]);
    acc[6] = fma(a2, a0, acc[6]); acc[7] = fma(a2, a1, acc[7]); acc[8] = fma(a2, a2, acc[8]);
    acc[9] += a0; acc[10] += a1; acc[11] += a2;
}

// partial sums of in^T in and of in's columns (rows < n_bins)
template <typename TI>
__global__ __launch_bounds__(kSpWG) void sp_stats_kernel(const TI *__restrict__ in, int64_t n_bins,
                                                         int64_t n_pad, double *__restrict__ partial) {
    int64_t r0, r1;
    sp_rows(n_pad, r0, r1);
    double acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.0;
    for (int64_t i = r0 + threadIdx.x; i < r1 && i < n_bins; i += kSpWG)
        sp_accumulate(acc, (double)in[3 * i], (double)in[3 * i + 1], (double)in[3 * i + 2]);
    sp_store_partial(acc, partial);
}

// the 12 sums themselves (groups added in order), for the stopping rule's one small read-back
__global__ __launch_bounds__(64) void sp_fold_partials_kernel(const double *__restrict__ partial,
                                                              int groups, double *__restrict__ out) {
    if (threadIdx.x >= 12) return;
    double v = 0.0;
    for (int g = 0; g < groups; ++g) v += partial[(int64_t)g * 12 + threadIdx.x];
    out[threadIdx.x] = v;
}

// g = R^T R (R upper) -> R^-1 (upper), 3 x 3 row-major.  False if g is not positive definite.
__host__ __device__ inline bool sp_chol3_inv_upper(const double *g, double *rinv) {
    double r[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < 3; ++j) {
        double d = g[j * 3 + j];
        for (int k = 0; k < j; ++k) d -= r[k * 3 + j] * r[k * 3 + j];
        if (!(d > 0.0)) return false;
        r[j * 3 + j] = sqrt(d);
        for (int c = j + 1; c < 3; ++c) {
            double v = g[j * 3 + c];
            for (int k = 0; k < j; ++k) v -= r[k * 3 + j] * r[k * 3 + c];
            r[j * 3 + c] = v / r[j * 3 + j];
        }
    }
    for (int q = 0; q < 9; ++q) rinv[q] = 0.0;
    for (int j = 0; j < 3; ++j) {
        rinv[j * 3 + j] = 1.0 / r[j * 3 + j];
        for (int i = j - 1; i >= 0; --i) {
            double v = 0.0;
            for (int k = i + 1; k <= j; ++k) v -= r[i * 3 + k] * rinv[k * 3 + j];
            rinv[i * 3 + j] = v / r[i * 3 + i];
        }
    }
    return true;
}

// The 3 x 3 step BETWEEN two passes, done by every workgroup of the consuming pass for itself
// (round 4, second version: as a kernel of its own it cost what every tiny kernel costs here,
// 4.7 us by the kernel trace, three times per product).  The producer left `groups` partial
// sums of 12 values; 12 x 16 threads add them in a fixed order (thread (q, l) takes the groups
// l, l + 16, ... of value q, then a 16-lane tree), thread 0 does the 3 x 3 step, LDS hands the
// map to everybody.  Every workgroup -- and every rank -- computes the same bits.
//   kSpMean     A = { mean = column sums / n, M = I, scale }           (centring)
//   kSpChol     A = { 0, R^-1, 1 } with R^T R = the Gram matrix        (Cholesky-QR pass)
//   kSpCholMean the same, and B.mean = (column sums / n) R^-1: the mean of what the map is
//               about to produce (sp_affine_centre_kernel)
// A Gram matrix that is not positive definite sets *flag and leaves the identity.
enum { kSpMean = 0, kSpChol = 1, kSpCholMean = 2 };
__device__ __forceinline__ void sp_map_from_partials(const double *__restrict__ partial, int groups,
                                                     int64_t n_bins, int mode, double scale,
                                                     int *__restrict__ flag, Affine3 &A, Affine3 &B) {
    __shared__ double tot[12];
    __shared__ Affine3 maps[2];
    const int t = threadIdx.x;
    if (t < 192) {
        const int q = t >> 4, l = t & 15;
        double v = 0.0;
        for (int g = l; g < groups; g += 16) v += partial[(int64_t)g * 12 + q];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 16);
        if (l == 0) tot[q] = v;
    }
    __syncthreads();
    if (t == 0) {
        Affine3 a, b;
        for (int k = 0; k < 3; ++k) a.mean[k] = 0.0;
        for (int k = 0; k < 9; ++k) a.m[k] = (k % 4 == 0) ? 1.0 : 0.0;
        a.scale = 1.0;
        b = a;
        if (mode == kSpMean) {
            for (int k = 0; k < 3; ++k) a.mean[k] = tot[9 + k] / (double)n_bins;
            a.scale = scale;
        } else {
            double rinv[9];
            if (!sp_chol3_inv_upper(tot, rinv)) {
                *flag = 1;                       // (every workgroup writes the same 1)
            } else {
                for (int k = 0; k < 9; ++k) a.m[k] = rinv[k];
                b = a;
                const double m0 = tot[9] / (double)n_bins, m1 = tot[10] / (double)n_bins,
                             m2 = tot[11] / (double)n_bins;
                b.mean[0] = m0 * rinv[0] + m1 * rinv[3] + m2 * rinv[6];
                b.mean[1] = m0 * rinv[1] + m1 * rinv[4] + m2 * rinv[7];
                b.mean[2] = m0 * rinv[2] + m1 * rinv[5] + m2 * rinv[8];
            }
        }
        maps[0] = a;
        maps[1] = b;
    }
    __syncthreads();
    A = maps[0];
    B = maps[1];
}

// out_i = A.scale * ((in_i - A.mean) A.m) for rows < n_bins (0 beyond), A = the 3 x 3 step
// `mode` on the producer's partial sums, + the partial sums of the OUTPUT.  In place
// (in == out) is fine: a thread reads its row before it writes it.
template <typename TI>
__global__ __launch_bounds__(kSpWG) void sp_affine_stats_kernel(const TI *in, double *out,
                                                                int64_t n_bins, int64_t n_pad,
                                                                const double *__restrict__ partial_in,
                                                                int groups, int mode, double scale,
                                                                int *__restrict__ flag,
                                                                double *__restrict__ partial) {
    Affine3 A, unused;
    sp_map_from_partials(partial_in, groups, n_bins, mode, scale, flag, A, unused);
    int64_t r0, r1;
    sp_rows(n_pad, r0, r1);
    double acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.0;
    for (int64_t i = r0 + threadIdx.x; i < r1; i += kSpWG) {
        double o0 = 0.0, o1 = 0.0, o2 = 0.0;
        if (i < n_bins) {
            const double x0 = (double)in[3 * i] - A.mean[0], x1 = (double)in[3 * i + 1] - A.mean[1],
                         x2 = (double)in[3 * i + 2] - A.mean[2];
            o0 = A.scale * (x0 * A.m[0] + x1 * A.m[3] + x2 * A.m[6]);
            o1 = A.scale * (x0 * A.m[1] + x1 * A.m[4] + x2 * A.m[7]);
            o2 = A.scale * (x0 * A.m[2] + x1 * A.m[5] + x2 * A.m[8]);
            sp_accumulate(acc, o0, o1, o2);
        }
        out[3 * i] = o0; out[3 * i + 1] = o1; out[3 * i + 2] = o2;
    }
    sp_store_partial(acc, partial);
}

// The last pass of an orthonormalisation: V = V' R^-1, written as double, and the sweep's
// right-hand sides mv = (T)(V - mean V) -- V centred, in the solver's type -- in one go; R^-1 and
// the mean from the producer's partial sums (kSpCholMean).
template <typename T>
__global__ __launch_bounds__(kSpWG) void sp_affine_centre_kernel(const double *in, double *out_v,
                                                                 T *__restrict__ out_mv,
                                                                 int64_t n_bins, int64_t n_pad,
                                                                 const double *__restrict__ partial_in,
                                                                 int groups, int *__restrict__ flag) {
    Affine3 A, B;
    sp_map_from_partials(partial_in, groups, n_bins, kSpCholMean, 1.0, flag, A, B);
    const int64_t i = (int64_t)blockIdx.x * kSpWG + threadIdx.x;
    if (i >= n_pad) return;
    double o0 = 0.0, o1 = 0.0, o2 = 0.0, c0 = 0.0, c1 = 0.0, c2 = 0.0;
    if (i < n_bins) {
        const double x0 = in[3 * i], x1 = in[3 * i + 1], x2 = in[3 * i + 2];
        o0 = x0 * A.m[0] + x1 * A.m[3] + x2 * A.m[6];
        o1 = x0 * A.m[1] + x1 * A.m[4] + x2 * A.m[7];
        o2 = x0 * A.m[2] + x1 * A.m[5] + x2 * A.m[8];
        c0 = o0 - B.mean[0]; c1 = o1 - B.mean[1]; c2 = o2 - B.mean[2];
    }
    out_v[3 * i] = o0; out_v[3 * i + 1] = o1; out_v[3 * i + 2] = o2;
    out_mv[3 * i] = (T)c0; out_mv[3 * i + 1] = (T)c1; out_mv[3 * i + 2] = (T)c2;
}

template <typename T>
__global__ void f64_to_T_kernel(const double *__restrict__ in, T *__restrict__ out, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n) out[e] = (T)in[e];
}
template <typename T>
__global__ void T_to_f64_kernel(const T *__restrict__ in, double *__restrict__ out, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n) out[e] = (double)in[e];
}

}  // namespace
