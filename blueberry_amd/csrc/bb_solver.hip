// bb_solver.hip -- the 3D-structure solver of libblueberry_hip.so: stress +
// gradient over the packed upper triangle of the wish-distance matrix, the
// deterministic partial reduce, and the coordinate update.  gfx950 only.
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
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "bb_comm.h"
#include "bb_common.h"

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
    static constexpr int MIN_WG = 4;            // __launch_bounds__: <= 128 VGPRs
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
    static constexpr int MIN_WG = 4;
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
    T rinv, dist;
    if constexpr (sizeof(T) == 4) {
        rinv = __builtin_amdgcn_rsqf(d2);
        dist = d2 * rinv;
    } else {
#ifdef BB_ABL_F64_LIBM        // timing experiment: library sqrt + IEEE divide (~55 fp64 ops)
        dist = sqrt(d2);
        rinv = 1.0 / dist;
#else
        // v_rsq_f64 seed, two Newton steps on 1/sqrt, one on sqrt: ~12 fp64 ops,
        // both results within 1-2 ulp (the parity tolerance is 1e-12)
        T r = __builtin_amdgcn_rsq(d2);
        const T h = T(0.5) * d2;
        r = r * fma(-h * r, r, T(1.5));
        r = r * fma(-h * r, r, T(1.5));
        dist = d2 * r;
        dist = fma(T(0.5) * r, fma(-dist, dist, d2), dist);
        rinv = r;
#endif
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

// One unit, generic (fp64) form: RPU matrix rows of LPR wave-loads each.  Load k of
// row r of the CURRENT unit is consumed from d[r*LPR + k], which is then refilled at
// once with the same load of the NEXT unit, so 8 KiB per wave stay in flight with a
// single register window.  xrow.get(q): the unit's q-th row coordinate, wave-uniform.
template <typename T, bool W, bool NT, int OP, typename XR>
__device__ __forceinline__ void process_unit(typename Traits<T>::Vec (&d)[8], const XR &xrow,
                                             const typename Traits<T>::Vec *__restrict__ next,
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
            d[r * LPR + k] = stream_load<NT>(next + (r * LPR + k) * 64);
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
#ifdef BB_ABL_NORSQ  // timing experiment: wrong results
    rinv = d2 * eps2;
#else
    rinv.x = __builtin_amdgcn_rsqf(d2.x);
    rinv.y = __builtin_amdgcn_rsqf(d2.y);
#endif
#ifdef BB_ABL_NOMASK
    const f32x2 res = (d2 * rinv - delta);
#else
    const f32x2 res = (d2 * rinv - delta) * weight01(delta);  // (dist - delta) or 0
#endif
    s2 += res * res;
    const f32x2 coef = res * rinv;
    if constexpr (FIRST) {
        rx = coef * dx; ry = coef * dy; rz = coef * dz;
    } else {
        rx += coef * dx; ry += coef * dy; rz += coef * dz;
    }
#ifndef BB_ABL_NOCOL
    st.g[K][H][0] -= coef * dx;
    st.g[K][H][1] -= coef * dy;
    st.g[K][H][2] -= coef * dz;
#endif
}

// d[] holds the unit's 8 wave-loads in row order: d[2*r + k] = row r, load k.
template <bool NT, int OP, bool DEFER>
__device__ __forceinline__ void process_unit_f32(float4 (&d)[8], const float (&xrow)[12],
                                                 const float4 *__restrict__ next, StripF32 &st,
                                                 double &stress, __amdgpu_buffer_rsrc_t row_rsrc,
                                                 unsigned row_voff, int stage_idx) {
    extern __shared__ __attribute__((aligned(16))) float row_lds[];
    f32x2 s2 = {0.f, 0.f};
    // The 12 row sums of the unit are collected into lanes 48..59 of one register
    // (after the reduction every lane >= 48 holds the wave total) and leave with
    // ONE 48-byte store per unit: a 12-byte store per row costs as much VMEM issue
    // as a 1-KiB load and measured 5.6 % of the kernel.
    const int slot = (int)(threadIdx.x & 63) - 48;  // value index this lane keeps, if 0..11
    const int comp = slot - 3 * (slot / 3);     // 0,1,2 = x,y,z
    float keep = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float xs = xrow[3 * r], ys = xrow[3 * r + 1], zs = xrow[3 * r + 2];
        const f32x2 xi = {xs, xs}, yi = {ys, ys}, zi = {zs, zs};
        f32x2 rx, ry, rz, qx, qy, qz;  // row-side sums of load 0 / load 1
        pair_step2<0, 0, true, OP>(f32x2{d[2 * r].x, d[2 * r].y}, xi, yi, zi, st, rx, ry, rz, s2);
        pair_step2<0, 1, false, OP>(f32x2{d[2 * r].z, d[2 * r].w}, xi, yi, zi, st, rx, ry, rz, s2);
        d[2 * r] = stream_load<NT>(next + (2 * r) * 64);
        // (no sched_barrier here: letting the scheduler mix the rows of a unit measured
        // 1 % faster at N=50k and 6 % faster at 1/8 size; it stays within 125 VGPRs)
        pair_step2<1, 0, true, OP>(f32x2{d[2 * r + 1].x, d[2 * r + 1].y}, xi, yi, zi, st, qx, qy, qz, s2);
        pair_step2<1, 1, false, OP>(f32x2{d[2 * r + 1].z, d[2 * r + 1].w}, xi, yi, zi, st, qx, qy, qz, s2);
        d[2 * r + 1] = stream_load<NT>(next + (2 * r + 1) * 64);
        rx += qx; ry += qy; rz += qz;
        float gx = rx.x + rx.y, gy = ry.x + ry.y, gz = rz.x + rz.y;
#ifndef BB_ABL_NODPP
        wave_sum_hi3(gx, gy, gz);
#endif
        const float mine = comp == 0 ? gx : (comp == 1 ? gy : gz);
        keep = (slot >= 3 * r && slot < 3 * r + 3) ? mine : keep;
    }
#ifndef BB_ABL_NOSTORE
    // DEFER: both are issued for every unit and exactly one of them lands -- the LDS
    // slot is a dummy word while the unit is stored directly, the store's lanes are all
    // out of range (free) while the unit is parked (see the kernel)
    if constexpr (DEFER) row_lds[stage_idx] = keep;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(keep), row_rsrc, row_voff, 0, 0);
#else
    asm volatile("" ::"v"(keep));
#endif
    stress += (double)(s2.x + s2.y);
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

// One wave = one contiguous chunk of units; 4 independent waves per workgroup.
// No LDS, no barriers, no atomics: results are bitwise reproducible.
//
// Arguments are separate __restrict__ pointers (not a struct) so that the
// read-only index arrays are provably unclobbered and load through the scalar
// cache.  Unit indices are 32-bit (a rank holds < 2^31 units = 16 TiB).
//   units      this rank's units, 8 KiB (RPU rows x VW columns) each
//   X          (n_pad, 3) coordinates
//   udesc      per local unit {i0, j0}
//   wave_range per wave {first, end} local unit indices
//   wave_slot  first column-partial slot of each wave
//   rowpart    3*RPU elements per unit, base shifted to the rank's first tile
//   colpart    3*VW elements per slot
//   stresspart one double per wave
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
template <typename T, bool W, bool NT, int OP, bool DEFER>
__global__ __launch_bounds__(256, (Lay<T, W>::MIN_WG)) void stress_grad_kernel(
    const T *__restrict__ units, const T *__restrict__ X, const int2 *__restrict__ udesc,
    const int2 *__restrict__ wave_range, const int32_t *__restrict__ wave_slot,
    T *__restrict__ rowpart, T *__restrict__ colpart, double *__restrict__ stresspart,
    int cap_units) {
    using Vec = typename Traits<T>::Vec;
    constexpr int VPL = Traits<T>::VPL;
    constexpr int VW = Lay<T, W>::VW;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int2 range = wave_range[w];
    const int ua = range.x, ub = range.y;
    double stress = 0.0;
    // DEFER: this wave's parking space, cap_units * 12 floats + 4 dummy words
    extern __shared__ __attribute__((aligned(16))) float row_lds[];
    const int stage0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * (cap_units * 12 + 4);
    const int park_from = (ub - ua) > cap_units ? (ub - ua) - cap_units : 0;

    if (ua < ub) {
        int slot = wave_slot[w];
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

        int2 dc = udesc[ua];                                   // current unit
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
#ifdef BB_ABL_XROW_VECTOR
        using XRow = XRowV;
#else
        using XRow = typename std::conditional<Lay<T, W>::SCALAR_XROW, XRowS, XRowV>::type;
#endif
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
        {
            const Vec *first = unit_ptr<T>(units, ua, lane);
#pragma unroll
            for (int r = 0; r < 8; ++r) d[r] = stream_load<NT>(first + r * 64);
        }
        // this wave's row partials: 3*RPU elements per unit of its group's chunk
        constexpr unsigned kRowBytes = 3 * Lay<T, W>::RPU * sizeof(T);
        const __amdgpu_buffer_rsrc_t row_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            rowpart + (int64_t)ua * (3 * Lay<T, W>::RPU), 0, (int)((unsigned)(ub - ua) * kRowBytes),
            0x00020000);

        auto unit_step = [&](int u) __attribute__((always_inline)) {
            // the wave's last unit "prefetches" itself: harmless, stays in bounds
            const int un = u + 1 < ub ? u + 1 : u;
            const XRow xrn = xrow_load(dn.x);
            const int2 dnn = udesc[un + 1 < ub ? un + 1 : un];
            unsigned row_voff;
            int stage_slot = 0;
            if constexpr (sizeof(T) == 4) {
                // fp32: lanes 48..59 hold one of the unit's 12 sums each.  Units before
                // park_from are stored directly; the later ones are parked in LDS slot
                // (k - park_from) and their store is dropped (every lane out of range).
                const int k = u - ua;
                const bool parked = DEFER && k >= park_from;
                const bool mine = lane >= 48 && lane < 60;
                row_voff = (mine && !parked) ? (unsigned)k * kRowBytes + (unsigned)(lane - 48) * 4u
                                             : kDropOffset;
                stage_slot = stage0 + ((mine && parked) ? (k - park_from) * 12 + (lane - 48)
                                                        : cap_units * 12 + (lane & 3));
            } else {                        // fp64: lane 63 stores each row's three sums
                row_voff = lane == 63 ? (unsigned)(u - ua) * kRowBytes : kDropOffset;
            }
            if constexpr (sizeof(T) == 4) {
#ifdef BB_ABL_XROW_VECTOR
                float xs12[12];
#pragma unroll
                for (int q = 0; q < 12; ++q) xs12[q] = lane_value(xr.v, q);
                process_unit_f32<NT, OP, DEFER>(d, xs12, unit_ptr<T>(units, un, lane), st, stress,
                                                row_rsrc, row_voff, stage_slot);
#else
                process_unit_f32<NT, OP, DEFER>(d, xr.v, unit_ptr<T>(units, un, lane), st, stress,
                                                row_rsrc, row_voff, stage_slot);
#endif
            }
            else
                process_unit<T, W, NT, OP>(d, xr, unit_ptr<T>(units, un, lane), st.xj, st.gc, stress,
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
        int u = ua;
        for (;;) {
            const int curj = dc.y;
            strip_load(curj);
            unit_step(u);
            ++u;
            while (u < ub && dc.y == curj) {
                unit_step(u);
                ++u;
            }
            strip_store(slot);
            ++slot;
            if (u >= ub) break;
        }
        if constexpr (DEFER && sizeof(T) == 4) {
            // the chunk's row sums, (ub - ua) * 12 floats, in one contiguous burst.
            // Lanes read what other lanes of this wave parked: LDS operations of one
            // wave execute in program order; the fence is for the compiler.
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int n4 = ((ub - ua) - park_from) * 3;   // float4 count
            float4 *dst = reinterpret_cast<float4 *>(rowpart + ((int64_t)ua + park_from) * 12);
            for (int q = lane; q < n4; q += 64) {
                const float *src = row_lds + stage0 + 4 * q;
                dst[q] = make_float4(src[0], src[1], src[2], src[3]);
            }
        }
    }

    // per-wave stress (fixed shuffle tree)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) stress += __shfl_down(stress, off, 64);
    if (lane == 0) stresspart[w] = stress;
}

// --------------------------------------------------------------------------
// reduce (+ update) kernel: one workgroup per vw-bin block
// --------------------------------------------------------------------------
constexpr int kReduceSlice = 16;  // chunks per stage-1 slice = chunks loaded per round trip

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
    unsigned long long seq;                  // peer mode: value the flags take
    int n_peers;
    double *__restrict__ stress_out;         // apply / stress-only: where the stress goes
    int64_t n_pad;
    int n_waves;
    int mode;
    T lr;
};

template <typename T, bool W>
__global__ __launch_bounds__(256) void reduce_kernel(ReduceParams<T> p) {
    constexpr int CH = 3 * Lay<T, W>::VW;
    constexpr int NE = (CH + 255) / 256;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    __shared__ __attribute__((aligned(16))) T push_stage[CH];   // peer mode only
    if (p.mode != kReduceStressOnly) {
        const int64_t k0 = p.blk_ptr[b], k1 = p.blk_ptr[b + 1];
        T acc[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) acc[j] = T(0);
        // Chunks are summed in list order (deterministic).  A whole slice of up to
        // kReduceSlice chunks is loaded before the first add, so a slice costs one
        // memory round trip, not one per chunk: this kernel is latency-bound.
        // (fp64: half as many chunks per round trip, for the same 96 VGPRs of loads)
        constexpr int kBatch = kReduceSlice * 4 / (int)sizeof(T);
        for (int64_t k = k0; k < k1; k += kBatch) {
            T v[kBatch][NE];
#pragma unroll
            for (int q = 0; q < kBatch; ++q) {
                const bool on = k + q < k1;
                const T *src = p.part + p.blk_chunk[on ? k + q : k0];
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const int e = tid + 256 * j;
                    v[q][j] = (on && e < CH) ? src[e] : T(0);
                }
            }
#pragma unroll
            for (int q = 0; q < kBatch; ++q)
#pragma unroll
                for (int j = 0; j < NE; ++j) acc[j] += v[q][j];
        }
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int e = tid + 256 * j;
            if (e < CH) {
                const int64_t o = (int64_t)b * CH + e;
                if (p.mode == kReducePartial) {
                    p.part_out[o] = acc[j];
                } else {
                    const T g = p.scale * acc[j];
                    if (p.mode == kReduceApply) {
                        // SPEC 2.4: V <- mu V - lr g ; X <- X + V   (mu = 0: X -= lr g)
                        const T v = p.mu * p.V[o] - p.lr * g;
                        p.V[o] = v;
                        p.X[o] += v;
                    } else if (p.mode == kReducePeer) {
                        push_stage[e] = g;
                    } else {
                        p.exch[o] = g;
                    }
                }
            }
        }
        if (p.mode == kReducePeer) {
            // the block's 3*vw values go out as 16-byte stores, 1 KiB per wave
            // instruction and peer: what crosses xGMI is long contiguous bursts
            typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));
            constexpr int NV = CH * (int)sizeof(T) / 16;
            __syncthreads();
            const vec_t *src = (const vec_t *)push_stage;
            for (int v = tid; v < NV; v += 256) {
                const vec_t val = src[v];
                for (int q = 0; q < p.n_peers; ++q)
                    ((vec_t *)(p.peer->dst[q] + (int64_t)b * CH))[v] = val;
            }
        }
    }
    if (p.mode == kReducePartial) return;
    if (b == 0) {
        __shared__ double sh[256];
        double s = 0.0;
        for (int i = tid; i < p.n_waves; i += 256) s += p.stresspart[i];
        sh[tid] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
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
                for (int q = 0; q < p.n_peers; ++q) {
                    p.peer->dst[q][3 * p.n_pad] = hi;
                    p.peer->dst[q][3 * p.n_pad + 1] = lo;
                }
            } else {
                *p.stress_out = S;
            }
        }
    }
    if (p.mode == kReducePeer) {
        // Every workgroup makes its stores visible system-wide and checks in; the
        // last one to do so raises this rank's flag on every peer (release).
        __shared__ int last;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope
        __syncthreads();
        if (tid == 0) last = atomicAdd(p.peer_counter, 1u) == gridDim.x - 1;
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


// Peer exchange, receiving side: wait until every source rank's flag has reached
// `seq`, then X <- X + (mu V - lr * sum over ranks, in rank order).  The wait is
// bounded: past `limit` ticks of the constant-rate clock the status word is set,
// and every later launch returns at once without touching X ("sticky" failure,
// reported by bb_solver_peer_status).  `arena` is this parity's first slot.
template <typename T>
__global__ __launch_bounds__(256) void peer_apply_kernel(
    T *__restrict__ X, T *__restrict__ V, const T *arena, const unsigned long long *flags,
    int world, int64_t slot_elems, int64_t n3, T lr, T mu, double *stress_out,
    unsigned long long seq, int *status, long long limit) {
    __shared__ int ok;
    const int tid = threadIdx.x;
    if (tid == 0) ok = 1;
    __syncthreads();
    if (tid < world) {
        if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            ok = 0;
        } else {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(flags + 8 * tid, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) <
                   seq) {
                if (wall_clock64() - t0 > limit) {
                    atomicExch(status, 1);
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
        }
    }
    __syncthreads();
    if (!ok) return;
    // The pollers' acquire + the barrier order every thread's loads after the
    // peers' stores; the arena is uncached memory, read past L1/L2 (sc0 sc1).
    const int64_t e = (int64_t)blockIdx.x * 256 + tid;
    if (e < n3) {
        // all slots are requested before the first add (one memory round trip per
        // eight ranks, not one per rank); the sum itself runs in rank order
        T g = T(0);
        for (int r0 = 0; r0 < world; r0 += 8) {
            T v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                v[q] = r0 + q < world
                           ? __hip_atomic_load(arena + (r0 + q) * slot_elems + e, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_SYSTEM)
                           : T(0);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (r0 + q < world) g += v[q];
        }
        const T v = mu * V[e] - lr * g;
        V[e] = v;
        X[e] += v;
    }
    if (e == 0) {
        double S = 0.0;
        for (int r = 0; r < world; ++r)
            S += (double)arena[r * slot_elems + n3] + (double)arena[r * slot_elems + n3 + 1];
        *stress_out = S;
    }
}

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
        if (sizeof(T) == 4 && v < 1e-30) v = 0.0;
        out[e] = (T)v;
    }
}

// Sparse (i, j, value) entries -> resident units (blocked-sparse input).  The
// units were zeroed ("no constraint") first.  tilemap[I * n_blocks + J] is the
// tile's index in the global list or -1.
template <typename T, bool W>
__global__ __launch_bounds__(256) void scatter_entries_kernel(
    const int64_t *__restrict__ rows, const int64_t *__restrict__ cols,
    const double *__restrict__ vals, int64_t nnz, const int32_t *__restrict__ tilemap,
    int64_t n_blocks, int64_t n_bins, int64_t u_begin, int64_t u_end, T *__restrict__ units,
    int kind, double neg_inv_alpha, const double *__restrict__ kr,
    const double *__restrict__ krexp, int *__restrict__ bad) {
    constexpr int VW = Lay<T, W>::VW, RPU = Lay<T, W>::RPU, UPT = VW / RPU;
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nnz) return;
    int64_t i = rows[k], j = cols[k];
    if (i == j) return;                     // the diagonal carries no pair
    if (i > j) { const int64_t t = i; i = j; j = t; }
    if (i < 0 || j >= n_bins) { atomicExch(bad, 1); return; }
    const int64_t I = i / VW, J = j / VW;
    const int32_t t = tilemap[I * n_blocks + J];
    if (t < 0) { atomicExch(bad, 2); return; }   // entry outside the tile list
    const int64_t ri = i - I * VW;
    const int64_t u = (int64_t)t * UPT + ri / RPU;
    if (u < u_begin || u >= u_end) return;  // another rank's unit
    double v = vals[k];
    // KR balancing + observed/expected, the element-wise form of the loop at
    // reference datatypes.pyx:166-169 (same operation order)
    if (kr != nullptr) v = v / (kr[i] * kr[j] * krexp[j - i]);
    const bool ok = (v > 0.0) && (v <= 1.7976931348623157e308);
    if (!ok)
        v = 0.0;
    else if (kind == BB_KIND_COUNTS)
        v = pow(v, neg_inv_alpha);
    if (sizeof(T) == 4 && v < 1e-30) v = 0.0;
    units[(u - u_begin) * (RPU * VW) + (ri % RPU) * VW + (j - J * VW)] = (T)v;
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
        if (sizeof(T) == 4 && v < 1e-30) v = 0.0;
        out[e] = (T)v;
    }
}

// Measurement only: the same waves read the same units with the same rolling
// 8-row window, but do nothing with the data except fold it into a checksum --
// the practical HBM read ceiling for this access pattern on this box.
template <bool NT>
__global__ __launch_bounds__(256, 4) void stream_read_kernel(const float4 *__restrict__ units,
                                                             const int2 *__restrict__ wave_range,
                                                             float *__restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int ua = wave_range[w].x, ub = wave_range[w].y;
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

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------
struct bb_solver {
    int dtype = BB_F32, device = 0, rank = 0, world = 1;
    bool wide = true;  // unit shape (Lay<T, wide>): false only for small fp64 problems
    bb_layout_info L{};
    std::vector<int32_t> tile_I, tile_J;  // global tile list (device order)
    int64_t u_begin = 0, u_end = 0, n_local = 0;
    int64_t t_first = 0, n_local_tiles = 0;
    std::vector<int2> udesc;  // host copy

    hipStream_t stream = nullptr;
    bool own_stream = false;

    void *d_units = nullptr, *d_X = nullptr, *d_V = nullptr, *d_part = nullptr, *d_exch = nullptr;
    double momentum = 0.0;
    bool own_exch = false;
    int2 *d_udesc = nullptr;
    int2 *d_wave_range = nullptr;
    bool nontemporal = true;
    int32_t *d_wave_slot = nullptr;
    double *d_stresspart = nullptr;
    int64_t *d_blk_ptr = nullptr, *d_blk_chunk = nullptr;    // final stage: one list per block
    int64_t *d_s1_ptr = nullptr, *d_s1_chunk = nullptr;      // stage 1: slices of long lists
    int64_t n_slices = 0, part2_off = 0;
    double *d_stress_hist = nullptr, *d_stress_scalar = nullptr;
    double *d_f64_tmp = nullptr;  // (n_pad,3) staging for coordinate I/O
    int64_t rowpart_elems = 0, colpart_elems = 0;
    int n_waves = 0, n_slots = 0;
    int defer_cap_units = 0;       // fp32: units of row sums a wave can park in LDS (0 = none)
    int64_t defer_lds_bytes = 0;   // dynamic LDS per workgroup for that, 0 = per-unit stores
    unsigned defer_attr_done = 0;  // kernel variants whose dynamic-LDS ceiling was raised
    int64_t hist_cap = 0, hist_n = 0;
    bool have_wish = false, have_coords = false, grad_pending = false;

    bb::Rccl::Comm comm = nullptr;  // direct RCCL path (bb_solver_comm_init), else null

    // peer exchange (bb_solver_peer_*)
    void *peer_arena = nullptr;             // this rank's receive arena (uncached)
    int64_t peer_slot_elems = 0, peer_arena_bytes = 0;
    std::vector<void *> peer_mapped;        // arenas of all ranks as mapped here (own = peer_arena)
    std::vector<void *> peer_opened;        // the ones that came from hipIpcOpenMemHandle
    void *d_peer_table = nullptr;           // PeerTable<T>[2], one per parity
    int *d_peer_status = nullptr;           // [0] sticky time-out flag
    unsigned *d_peer_counter = nullptr;
    unsigned long long peer_seq = 0;        // iterations exchanged so far
    long long peer_limit_ticks = 0;
    bool peer_connected = false;

    bool timing = false;
    int timing_stride = 1;      // events on every timing_stride-th iteration
    int64_t timing_iter = 0;    // iterations seen since timing was (re-)enabled
    std::vector<hipEvent_t> ev;  // triples: start, after grad, after reduce
    size_t ev_used = 0;
};

namespace {

constexpr int64_t kHistCap = 1 << 20;
constexpr size_t kMaxTimedLaunches = 4096;

template <typename T>
int dev_alloc(T **p, int64_t count) {
    *p = nullptr;
    if (count <= 0) count = 1;
    hipError_t e = hipMalloc((void **)p, (size_t)count * sizeof(T));
    if (e != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
    return BB_OK;
}

#define BB_TRY(expr)                 \
    do {                             \
        int _rc = (expr);            \
        if (_rc != BB_OK) return _rc; \
    } while (0)

// Waves per CU (4 waves = one workgroup).  Measured on MI355X (profiles/
// r01_sweeps.txt): with the rolling 8-KiB window one wave per SIMD already keeps
// enough bytes in flight, and fewer, longer chunks mean fewer column-partial
// slots and less prologue per byte: 4/CU wins below ~400k units per rank
// (-12 % kernel time at 1/8 of the N=50k matrix), 4 and 8 tie above, 16 is
// 2 % slower.  BB_WAVES_PER_CU overrides.
int waves_per_cu(int64_t n_local) {
    const char *e = getenv("BB_WAVES_PER_CU");
    int v = e ? atoi(e) : (n_local >= 400000 ? 8 : 4);
    if (v < 1) v = 1;
    if (v > 32) v = 32;
    return v;
}

// Build every index the kernels need from (tile list, unit range).
int build_indices(bb_solver *s) {
    const int64_t upt = s->L.units_per_tile, vw = s->L.vw;
    const int64_t ch = 3 * vw;
    s->n_local = s->u_end - s->u_begin;
    s->t_first = s->n_local > 0 ? s->u_begin / upt : 0;
    const int64_t t_last = s->n_local > 0 ? (s->u_end - 1) / upt : -1;
    s->n_local_tiles = s->n_local > 0 ? t_last - s->t_first + 1 : 0;

    s->udesc.resize((size_t)std::max<int64_t>(s->n_local, 1));
    for (int64_t ul = 0; ul < s->n_local; ++ul) {
        const int64_t u = s->u_begin + ul, t = u / upt, sub = u % upt;
        s->udesc[ul] = make_int2((int)(s->tile_I[t] * vw + sub * s->L.rows_per_unit),
                                 (int)(s->tile_J[t] * vw));
    }

    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device);
    int64_t want = (int64_t)cus * waves_per_cu(s->n_local);
    int64_t nw = std::max<int64_t>(1, std::min<int64_t>(want, s->n_local));
    nw = bb::round_up(nw, 4);
    s->n_waves = (int)nw;

    if (s->n_local >= ((int64_t)1 << 31))
        return bb::fail(BB_ERR_INVALID, "bb_solver_create: more than 2^31 units on one rank");
    {
        // non-temporal matrix loads are the default (measured: +2.5 % on the
        // kernel, +9 % on a pure read sweep); BB_NT=0 turns them off
        const char *e = getenv("BB_NT");
        s->nontemporal = !(e && atoi(e) == 0);
    }
    // wave w owns the contiguous chunk [n_local*w/nw, n_local*(w+1)/nw)
    std::vector<int2> wave_range(nw);
    std::vector<int32_t> wave_slot(nw);
    int64_t chunk_max = 0;
    for (int64_t w = 0; w < nw; ++w) {
        wave_range[w] = make_int2((int)((__int128)s->n_local * w / nw),
                                  (int)((__int128)s->n_local * (w + 1) / nw));
        chunk_max = std::max<int64_t>(chunk_max, wave_range[w].y - wave_range[w].x);
    }
    {
        // fp32: a wave parks the row sums of (the last cap units of) its chunk in LDS
        // until it has finished reading (stress_grad_kernel, DEFER); cap is what the
        // workgroups sharing a CU can hold.  BB_DEFER_ROWS=0 turns it off.
        const char *e = getenv("BB_DEFER_ROWS");
        const int64_t wgs_per_cu = std::max<int64_t>(1, (nw / 4 + cus - 1) / cus);
        const int64_t budget = 156 * 1024 / wgs_per_cu;               // of the CU's 160 KiB
        const int64_t cap = std::min<int64_t>(chunk_max, (budget / 4 - 16) / 48);
        const bool on = !(e && atoi(e) == 0);
        if (s->dtype == BB_F32 && on && cap > 0) {
            s->defer_cap_units = (int)cap;
            s->defer_lds_bytes = 4 * (cap * 48 + 16);
        }
    }
    // column-partial slots: one per (wave, strip) intersection, in wave order
    std::vector<int32_t> slot_strip;
    for (int64_t w = 0; w < nw; ++w) {
        wave_slot[w] = (int32_t)slot_strip.size();
        int cur = -1;
        for (int64_t ul = wave_range[w].x; ul < wave_range[w].y; ++ul) {
            const int J = s->udesc[ul].y / (int)vw;
            if (J != cur) {
                slot_strip.push_back(J);
                cur = J;
            }
        }
    }
    s->n_slots = (int)slot_strip.size();
    s->rowpart_elems = s->n_local_tiles * ch;
    s->colpart_elems = (int64_t)s->n_slots * ch;

    // reduce CSR: block b <- row chunks of local tiles with I == b, then column slots of strip b
    const int64_t nb = s->L.n_blocks;
    std::vector<int64_t> blk_ptr(nb + 1, 0);
    for (int64_t lt = 0; lt < s->n_local_tiles; ++lt) blk_ptr[s->tile_I[s->t_first + lt] + 1]++;
    for (int sl = 0; sl < s->n_slots; ++sl) blk_ptr[slot_strip[sl] + 1]++;
    for (int64_t b = 0; b < nb; ++b) blk_ptr[b + 1] += blk_ptr[b];
    std::vector<int64_t> blk_chunk((size_t)std::max<int64_t>(blk_ptr[nb], 1));
    std::vector<int64_t> fill(blk_ptr.begin(), blk_ptr.end() - 1);
    for (int64_t lt = 0; lt < s->n_local_tiles; ++lt)
        blk_chunk[fill[s->tile_I[s->t_first + lt]]++] = lt * ch;
    for (int sl = 0; sl < s->n_slots; ++sl)
        blk_chunk[fill[slot_strip[sl]]++] = s->rowpart_elems + (int64_t)sl * ch;

    // Two-stage reduce: a block's list can hold hundreds of chunks (one per strip
    // of its tile row).  Lists longer than kSlice are cut into slices that stage 1
    // sums in parallel into `part2`; the final stage then sums the slice results.
    constexpr int64_t kSlice = kReduceSlice;
    std::vector<int64_t> s1_ptr(1, 0), s1_chunk, fin_ptr(nb + 1, 0), fin_chunk;
    s->part2_off = s->rowpart_elems + s->colpart_elems;
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t k0 = blk_ptr[b], k1 = blk_ptr[b + 1];
        if (k1 - k0 <= kSlice) {
            for (int64_t k = k0; k < k1; ++k) fin_chunk.push_back(blk_chunk[k]);
        } else {
            for (int64_t k = k0; k < k1; k += kSlice) {
                const int64_t ke = std::min(k + kSlice, k1);
                fin_chunk.push_back(s->part2_off + (int64_t)(s1_ptr.size() - 1) * ch);
                for (int64_t q = k; q < ke; ++q) s1_chunk.push_back(blk_chunk[q]);
                s1_ptr.push_back((int64_t)s1_chunk.size());
            }
        }
        fin_ptr[b + 1] = (int64_t)fin_chunk.size();
    }
    s->n_slices = (int64_t)s1_ptr.size() - 1;
    if (s1_chunk.empty()) s1_chunk.push_back(0);
    if (fin_chunk.empty()) fin_chunk.push_back(0);
    const int64_t part_total = s->part2_off + s->n_slices * ch;

    const int64_t es = bb::elem_size(s->dtype);
    BB_TRY(dev_alloc((char **)&s->d_units, std::max<int64_t>(s->n_local, 1) * bb::kUnitBytes));
    BB_TRY(dev_alloc((char **)&s->d_X, s->L.n_pad * 3 * es));
    BB_TRY(dev_alloc((char **)&s->d_V, s->L.n_pad * 3 * es));
    BB_TRY(dev_alloc((char **)&s->d_part, part_total * es));
    BB_TRY(dev_alloc(&s->d_udesc, (int64_t)s->udesc.size()));
    BB_TRY(dev_alloc(&s->d_wave_range, nw));
    BB_TRY(dev_alloc(&s->d_wave_slot, nw));
    BB_TRY(dev_alloc(&s->d_stresspart, nw));
    BB_TRY(dev_alloc(&s->d_blk_ptr, nb + 1));
    BB_TRY(dev_alloc(&s->d_blk_chunk, (int64_t)fin_chunk.size()));
    BB_TRY(dev_alloc(&s->d_s1_ptr, (int64_t)s1_ptr.size()));
    BB_TRY(dev_alloc(&s->d_s1_chunk, (int64_t)s1_chunk.size()));
    BB_TRY(dev_alloc(&s->d_stress_hist, kHistCap));
    BB_TRY(dev_alloc(&s->d_stress_scalar, 1));
    BB_TRY(dev_alloc(&s->d_f64_tmp, s->L.n_pad * 3));
    BB_TRY(dev_alloc((char **)&s->d_exch, (3 * s->L.n_pad + 2) * es));
    s->own_exch = true;
    s->hist_cap = kHistCap;

    hipStream_t st = s->stream;
    BB_HIP_CHECK(hipMemcpyAsync(s->d_udesc, s->udesc.data(), s->udesc.size() * sizeof(int2),
                                hipMemcpyHostToDevice, st));
    BB_HIP_CHECK(hipMemcpyAsync(s->d_wave_range, wave_range.data(), nw * sizeof(int2),
                                hipMemcpyHostToDevice, st));
    BB_HIP_CHECK(hipMemcpyAsync(s->d_wave_slot, wave_slot.data(), nw * sizeof(int32_t),
                                hipMemcpyHostToDevice, st));
    BB_HIP_CHECK(hipMemcpyAsync(s->d_blk_ptr, fin_ptr.data(), (nb + 1) * sizeof(int64_t),
                                hipMemcpyHostToDevice, st));
    BB_HIP_CHECK(hipMemcpyAsync(s->d_blk_chunk, fin_chunk.data(),
                                fin_chunk.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    BB_HIP_CHECK(hipMemcpyAsync(s->d_s1_ptr, s1_ptr.data(), s1_ptr.size() * sizeof(int64_t),
                                hipMemcpyHostToDevice, st));
    BB_HIP_CHECK(hipMemcpyAsync(s->d_s1_chunk, s1_chunk.data(),
                                s1_chunk.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    // rows of boundary tiles owned by another rank are never written: keep them 0
    BB_HIP_CHECK(hipMemsetAsync(s->d_part, 0, (size_t)part_total * es, st));
    BB_HIP_CHECK(hipMemsetAsync(s->d_X, 0, (size_t)(s->L.n_pad * 3 * es), st));
    BB_HIP_CHECK(hipMemsetAsync(s->d_V, 0, (size_t)(s->L.n_pad * 3 * es), st));
    BB_HIP_CHECK(hipMemsetAsync(s->d_exch, 0, (size_t)((3 * s->L.n_pad + 2) * es), st));
    BB_HIP_CHECK(hipMemsetAsync(s->d_stresspart, 0, (size_t)nw * sizeof(double), st));
    BB_HIP_CHECK(hipStreamSynchronize(st));  // the host vectors above die with this scope
    return BB_OK;
}

// (dtype, layout) -> template arguments.  F(float, wide), F(double, wide), F(double, narrow).
#define BB_BY_LAYOUT(s, F, ...)                                          \
    ((s)->dtype == BB_F32 ? F<float, true>(__VA_ARGS__)                  \
                          : ((s)->wide ? F<double, true>(__VA_ARGS__)    \
                                       : F<double, false>(__VA_ARGS__)))

template <typename T, bool W>
int launch_grad_t(bb_solver *s, int op, const void *x_in) {
    const T *units = (const T *)s->d_units;
    const T *X = (const T *)x_in;
    // row partials are indexed by local unit; the buffer starts at the rank's first tile
    T *rowpart = (T *)s->d_part +
                 (s->u_begin - s->t_first * s->L.units_per_tile) * (3 * s->L.rows_per_unit);
    T *colpart = (T *)s->d_part + s->rowpart_elems;
    const dim3 grid(s->n_waves / 4), block(256);
    // fp32 with the whole chunk of row sums parked in LDS (s->defer_lds_bytes > 0), or the
    // per-unit store.  The dynamic-LDS ceiling of a kernel is raised once per instantiation.
#define BB_LAUNCH2(NTV, OPV, DEF, LDS)                                                          \
    do {                                                                                        \
        auto kern = stress_grad_kernel<T, W, NTV, OPV, DEF>;                                    \
        constexpr unsigned bit = 1u << ((NTV ? 2 : 0) + (OPV == kOpMatvec2 ? 1 : 0));          \
        if ((LDS) > 0 && !(s->defer_attr_done & bit)) {                                        \
            BB_HIP_CHECK(hipFuncSetAttribute((const void *)kern,                                \
                                             hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                             (int)(LDS)));                                      \
            s->defer_attr_done |= bit;                                                          \
        }                                                                                       \
        hipLaunchKernelGGL(kern, grid, block, (size_t)(LDS), s->stream, units, X, s->d_udesc,   \
                           s->d_wave_range, s->d_wave_slot, rowpart, colpart, s->d_stresspart,  \
                           s->defer_cap_units);                                                 \
    } while (0)
#define BB_LAUNCH(NTV, OPV)                                                                     \
    do {                                                                                        \
        if (sizeof(T) == 4 && s->defer_lds_bytes > 0)                                           \
            BB_LAUNCH2(NTV, OPV, (sizeof(T) == 4), s->defer_lds_bytes);                         \
        else                                                                                    \
            BB_LAUNCH2(NTV, OPV, false, 0);                                                     \
    } while (0)
    if (op == kOpMatvec2) {
        if (s->nontemporal) BB_LAUNCH(true, kOpMatvec2); else BB_LAUNCH(false, kOpMatvec2);
    } else {
        if (s->nontemporal) BB_LAUNCH(true, kOpStress); else BB_LAUNCH(false, kOpStress);
    }
#undef BB_LAUNCH
#undef BB_LAUNCH2
    BB_HIP_CHECK(hipGetLastError());
    return BB_OK;
}

template <typename T, bool W>
int launch_reduce_t(bb_solver *s, int mode, double lr, double *stress_out, double scale) {
    ReduceParams<T> p;
    p.part = (const T *)s->d_part;
    p.stresspart = s->d_stresspart;
    p.X = (T *)s->d_X;
    p.V = (T *)s->d_V;
    p.mu = (T)s->momentum;
    p.scale = (T)scale;
    p.exch = (T *)s->d_exch;
    p.part_out = (T *)s->d_part + s->part2_off;
    p.stress_out = stress_out;
    p.n_pad = s->L.n_pad;
    p.n_waves = s->n_waves;
    p.lr = (T)lr;
    p.peer = nullptr;
    p.peer_counter = s->d_peer_counter;
    p.seq = 0;
    p.n_peers = 0;
    if (mode == kReducePeer) {
        // the caller has bumped peer_seq: iteration k (1-based) uses parity k & 1
        p.peer = (const PeerTable<T> *)s->d_peer_table + (s->peer_seq & 1);
        p.seq = s->peer_seq;
        p.n_peers = s->world;
    }
    if (mode != kReduceStressOnly && s->n_slices > 0) {
        p.blk_ptr = s->d_s1_ptr;
        p.blk_chunk = s->d_s1_chunk;
        p.mode = kReducePartial;
        hipLaunchKernelGGL((reduce_kernel<T, W>), dim3((unsigned)s->n_slices), dim3(256), 0, s->stream, p);
        BB_HIP_CHECK(hipGetLastError());
    }
    p.blk_ptr = s->d_blk_ptr;
    p.blk_chunk = s->d_blk_chunk;
    p.mode = mode;
    const int grid = mode == kReduceStressOnly ? 1 : (int)s->L.n_blocks;
    hipLaunchKernelGGL((reduce_kernel<T, W>), dim3(grid), dim3(256), 0, s->stream, p);
    BB_HIP_CHECK(hipGetLastError());
    return BB_OK;
}

int launch_grad(bb_solver *s, int op = kOpStress, const void *x_in = nullptr) {
    if (!x_in) x_in = s->d_X;
    return BB_BY_LAYOUT(s, launch_grad_t, s, op, x_in);
}
int launch_reduce(bb_solver *s, int mode, double lr, double *stress_out, double scale = 2.0) {
    return BB_BY_LAYOUT(s, launch_reduce_t, s, mode, lr, stress_out, scale);
}

hipEvent_t *timing_slot(bb_solver *s) {
    if (!s->timing) return nullptr;
    if (s->timing_iter++ % s->timing_stride != 0 || s->ev_used + 3 > s->ev.size()) return nullptr;
    hipEvent_t *e = &s->ev[s->ev_used];
    s->ev_used += 3;
    return e;
}

int check_ready(const bb_solver *s, const char *who) {
    if (!s) return bb::fail(BB_ERR_INVALID, std::string(who) + ": solver is NULL");
    if (!s->have_wish)
        return bb::fail(BB_ERR_STATE, std::string(who) + ": no wish distances set");
    if (!s->have_coords)
        return bb::fail(BB_ERR_STATE, std::string(who) + ": no coordinates set");
    return BB_OK;
}

template <typename T, bool W>
int set_wish_dense_t(bb_solver *s, const double *host, int64_t ld, int kind, double alpha) {
    constexpr int VW = Lay<T, W>::VW;
    const int64_t upt = s->L.units_per_tile, n = s->L.n_bins;
    constexpr int64_t kRunTiles = 4;  // tiles staged per copy
    double *stage = nullptr;
    BB_TRY(dev_alloc(&stage, kRunTiles * VW * VW));
    // Copy, convert and the next copy are all enqueued on the solver's stream,
    // so one staging buffer is enough and nothing depends on null-stream rules.
    int rc = BB_OK;
    int64_t ul = 0;
    while (ul < s->n_local && rc == BB_OK) {
        // a run: consecutive local units whose rows are consecutive in one strip
        const int j0 = s->udesc[ul].y;
        const int64_t i_start = s->udesc[ul].x;
        int64_t ue = ul + 1;
        while (ue < s->n_local && ue - ul < kRunTiles * upt && s->udesc[ue].y == j0 &&
               s->udesc[ue].x == i_start + (ue - ul) * s->L.rows_per_unit)
            ++ue;
        const int64_t rows = (ue - ul) * s->L.rows_per_unit;
        const int64_t rows_valid =
            std::max<int64_t>(0, std::min<int64_t>(i_start + rows, n) - i_start);
        const int64_t cols_valid =
            std::max<int64_t>(0, std::min<int64_t>((int64_t)j0 + VW, n) - j0);
        hipError_t e = hipSuccess;
        if (rows_valid < rows || cols_valid < VW)
            e = hipMemsetAsync(stage, 0, (size_t)rows * VW * sizeof(double), s->stream);
        if (e == hipSuccess && rows_valid > 0 && cols_valid > 0)
            e = hipMemcpy2DAsync(stage, VW * sizeof(double), host + i_start * ld + j0,
                                 (size_t)ld * sizeof(double), (size_t)cols_valid * sizeof(double),
                                 (size_t)rows_valid, hipMemcpyHostToDevice, s->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL((convert_units_kernel<T, W>), dim3((unsigned)(ue - ul)), dim3(256), 0,
                               s->stream, stage, (T *)s->d_units, s->d_udesc, ul, i_start, n,
                               kind, -1.0 / alpha);
            e = hipGetLastError();
        }
        if (e != hipSuccess)
            rc = bb::fail(BB_ERR_HIP, std::string("set_wish_dense: ") + hipGetErrorString(e));
        ul = ue;
    }
    hipError_t e = hipStreamSynchronize(s->stream);
    if (rc == BB_OK && e != hipSuccess)
        rc = bb::fail(BB_ERR_HIP, std::string("set_wish_dense: ") + hipGetErrorString(e));
    hipFree(stage);
    return rc;
}

}  // namespace

// ---- peer exchange ---------------------------------------------------------
namespace {

// What travels between ranks: the IPC handle of the arena plus what is needed to
// check that both sides agree on its shape (and to short-cut ranks that live in
// the same process, where an IPC handle cannot be opened).
struct PeerHandle {
    hipIpcMemHandle_t ipc;   // 64 bytes
    uint64_t magic;
    int64_t arena_bytes, slot_elems;
    int64_t pid;
    uint64_t raw;            // the exporter's own pointer (same-process ranks only)
    int32_t rank, world, dtype, device;
};
static_assert(sizeof(PeerHandle) <= BB_PEER_HANDLE_BYTES, "handle blob too small");
constexpr uint64_t kPeerMagic = 0x6262706565723031ull;  // "bbpeer01"

int64_t peer_flags_offset(const bb_solver *s) {
    return 2 * (int64_t)s->world * s->peer_slot_elems * bb::elem_size(s->dtype);
}

template <typename T>
int build_peer_tables(bb_solver *s) {
    PeerTable<T> tab[2];
    memset(tab, 0, sizeof(tab));
    const int64_t foff = peer_flags_offset(s);
    for (int par = 0; par < 2; ++par)
        for (int q = 0; q < s->world; ++q) {
            char *base = (char *)s->peer_mapped[q];
            tab[par].dst[q] = (T *)base + ((int64_t)par * s->world + s->rank) * s->peer_slot_elems;
            tab[par].flag[q] = (unsigned long long *)(base + foff) + 8 * s->rank;
        }
    BB_TRY(dev_alloc((char **)&s->d_peer_table, (int64_t)sizeof(tab)));
    BB_HIP_CHECK(hipMemcpy(s->d_peer_table, tab, sizeof(tab), hipMemcpyHostToDevice));
    return BB_OK;
}

}  // namespace

extern "C" {

int bb_solver_create(bb_solver **out, int64_t n_bins, int dtype, int device, int rank, int world,
                     const int32_t *tile_I, const int32_t *tile_J, int64_t n_tiles) {
    BB_REQUIRE(out != nullptr, "bb_solver_create: out is NULL");
    *out = nullptr;
    BB_REQUIRE(dtype == BB_F32 || dtype == BB_F64, "bb_solver_create: bad dtype");
    BB_REQUIRE(n_bins >= 2, "bb_solver_create: n_bins must be >= 2");
    BB_REQUIRE(n_bins <= (int64_t)700000000, "bb_solver_create: n_bins too large");
    BB_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bb_solver_create: bad rank/world");
    BB_REQUIRE((tile_I == nullptr) == (tile_J == nullptr) && (tile_I != nullptr || n_tiles == 0),
               "bb_solver_create: tile_I/tile_J/n_tiles inconsistent");
    BB_TRY(bb::use_device(device));

    bb_solver *s = new (std::nothrow) bb_solver();
    if (!s) return bb::fail(BB_ERR_NOMEM, "bb_solver_create: out of host memory");
    s->dtype = dtype;
    s->device = device;
    s->rank = rank;
    s->world = world;
    int rc = bb_layout_dense_info(n_bins, dtype, &s->L);
    s->wide = bb::wide_layout(dtype, n_bins);
    if (rc == BB_OK) {
        if (tile_I == nullptr) {
            s->tile_I.resize((size_t)s->L.n_tiles);
            s->tile_J.resize((size_t)s->L.n_tiles);
            rc = bb_layout_dense_tiles(n_bins, dtype, s->tile_I.data(), s->tile_J.data(),
                                       s->L.n_tiles);
        } else {
            // blocked-sparse: validate order (J, then I ascending; I <= J < n_blocks)
            for (int64_t t = 0; t < n_tiles && rc == BB_OK; ++t) {
                const bool in_range = tile_I[t] >= 0 && tile_I[t] <= tile_J[t] &&
                                      tile_J[t] < s->L.n_blocks;
                const bool ordered =
                    t == 0 || tile_J[t] > tile_J[t - 1] ||
                    (tile_J[t] == tile_J[t - 1] && tile_I[t] > tile_I[t - 1]);
                if (!in_range || !ordered)
                    rc = bb::fail(BB_ERR_INVALID,
                                  "bb_solver_create: tile list must be strictly ordered by "
                                  "(J, I) with 0 <= I <= J < n_blocks");
            }
            s->tile_I.assign(tile_I, tile_I + n_tiles);
            s->tile_J.assign(tile_J, tile_J + n_tiles);
            s->L.n_tiles = n_tiles;
            s->L.n_units = n_tiles * s->L.units_per_tile;
        }
    }
    if (rc == BB_OK) rc = bb_layout_rank_units(s->L.n_units, rank, world, &s->u_begin, &s->u_end);
    if (rc == BB_OK) {
        hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
        if (e != hipSuccess)
            rc = bb::fail(BB_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
        else
            s->own_stream = true;
    }
    if (rc == BB_OK) rc = build_indices(s);
    if (rc != BB_OK) {
        std::string keep = bb_last_error();
        bb_solver_destroy(s);
        bb::set_error(keep);
        return rc;
    }
    *out = s;
    return BB_OK;
}

int bb_solver_destroy(bb_solver *s) {
    if (!s) return BB_OK;
    hipSetDevice(s->device);
    if (s->stream || !s->own_stream) hipStreamSynchronize(s->stream);
    if (s->comm) bb::rccl().CommDestroy(s->comm);
    for (void *m : s->peer_opened) hipIpcCloseMemHandle(m);
    hipFree(s->peer_arena);
    hipFree(s->d_peer_table);
    hipFree(s->d_peer_status);
    hipFree(s->d_peer_counter);
    for (hipEvent_t e : s->ev) hipEventDestroy(e);
    hipFree(s->d_units);
    hipFree(s->d_X);
    hipFree(s->d_V);
    hipFree(s->d_part);
    if (s->own_exch) hipFree(s->d_exch);
    hipFree(s->d_udesc);
    hipFree(s->d_wave_range);
    hipFree(s->d_wave_slot);
    hipFree(s->d_stresspart);
    hipFree(s->d_blk_ptr);
    hipFree(s->d_blk_chunk);
    hipFree(s->d_s1_ptr);
    hipFree(s->d_s1_chunk);
    hipFree(s->d_stress_hist);
    hipFree(s->d_stress_scalar);
    hipFree(s->d_f64_tmp);
    if (s->own_stream && s->stream) hipStreamDestroy(s->stream);
    delete s;
    return BB_OK;
}

int bb_solver_set_stream(bb_solver *s, void *hip_stream) {
    BB_REQUIRE(s != nullptr, "bb_solver_set_stream: solver is NULL");
    BB_TRY(bb::enter_device(s->device));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    if (s->own_stream && s->stream) hipStreamDestroy(s->stream);
    s->stream = (hipStream_t)hip_stream;
    s->own_stream = false;
    return BB_OK;
}

int bb_solver_layout(const bb_solver *s, bb_layout_info *info, int64_t *u_begin, int64_t *u_end) {
    BB_REQUIRE(s != nullptr, "bb_solver_layout: solver is NULL");
    if (info) *info = s->L;
    if (u_begin) *u_begin = s->u_begin;
    if (u_end) *u_end = s->u_end;
    return BB_OK;
}

int bb_solver_set_wish_dense(bb_solver *s, const double *host, int64_t ld, int kind,
                             double alpha) {
    BB_REQUIRE(s != nullptr && host != nullptr, "bb_solver_set_wish_dense: NULL argument");
    BB_REQUIRE(ld >= s->L.n_bins, "bb_solver_set_wish_dense: ld < n_bins");
    BB_REQUIRE(kind == BB_KIND_WISH || kind == BB_KIND_COUNTS,
               "bb_solver_set_wish_dense: bad kind");
    BB_REQUIRE(kind == BB_KIND_WISH || alpha > 0.0, "bb_solver_set_wish_dense: alpha must be > 0");
    BB_TRY(bb::enter_device(s->device));
    int rc = BB_BY_LAYOUT(s, set_wish_dense_t, s, host, ld, kind, alpha);
    if (rc == BB_OK) s->have_wish = true;
    return rc;
}

int bb_solver_set_wish_sparse(bb_solver *s, const int64_t *rows, const int64_t *cols,
                              const double *vals, int64_t nnz, int kind, double alpha,
                              const double *KRnorm, const double *KRexpected) {
    BB_REQUIRE(s != nullptr, "bb_solver_set_wish_sparse: solver is NULL");
    BB_REQUIRE(nnz >= 0 && (nnz == 0 || (rows && cols && vals)),
               "bb_solver_set_wish_sparse: NULL entries");
    BB_REQUIRE(kind == BB_KIND_WISH || kind == BB_KIND_COUNTS, "bb_solver_set_wish_sparse: bad kind");
    BB_REQUIRE(kind == BB_KIND_WISH || alpha > 0.0, "bb_solver_set_wish_sparse: alpha must be > 0");
    BB_REQUIRE((KRnorm == nullptr) == (KRexpected == nullptr),
               "bb_solver_set_wish_sparse: KRnorm and KRexpected go together");
    BB_TRY(bb::enter_device(s->device));
    const int64_t nb = s->L.n_blocks;
    std::vector<int32_t> tilemap((size_t)(nb * nb), -1);
    for (size_t t = 0; t < s->tile_I.size(); ++t)
        tilemap[(size_t)s->tile_I[t] * nb + s->tile_J[t]] = (int32_t)t;
    int32_t *d_map = nullptr;
    int *d_bad = nullptr;
    int64_t *d_rows = nullptr, *d_cols = nullptr;
    double *d_vals = nullptr, *d_kr = nullptr, *d_ke = nullptr;
    constexpr int64_t kChunk = 1 << 22;  // entries staged per copy
    const int64_t cap = std::max<int64_t>(1, std::min(nnz, kChunk));
    int rc = dev_alloc(&d_map, nb * nb);
    if (rc == BB_OK) rc = dev_alloc(&d_bad, 1);
    if (rc == BB_OK) rc = dev_alloc(&d_rows, cap);
    if (rc == BB_OK) rc = dev_alloc(&d_cols, cap);
    if (rc == BB_OK) rc = dev_alloc(&d_vals, cap);
    if (rc == BB_OK && KRnorm) rc = dev_alloc(&d_kr, s->L.n_bins);
    if (rc == BB_OK && KRnorm) rc = dev_alloc(&d_ke, s->L.n_bins);
    hipError_t e = hipSuccess;
    int host_bad = 0;
    if (rc == BB_OK) {
        e = hipMemcpyAsync(d_map, tilemap.data(), tilemap.size() * sizeof(int32_t),
                           hipMemcpyHostToDevice, s->stream);
        if (e == hipSuccess && KRnorm)
            e = hipMemcpyAsync(d_kr, KRnorm, (size_t)s->L.n_bins * 8, hipMemcpyHostToDevice, s->stream);
        if (e == hipSuccess && KRnorm)
            e = hipMemcpyAsync(d_ke, KRexpected, (size_t)s->L.n_bins * 8, hipMemcpyHostToDevice,
                               s->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(int), s->stream);
        if (e == hipSuccess)
            e = hipMemsetAsync(s->d_units, 0, (size_t)std::max<int64_t>(s->n_local, 1) * bb::kUnitBytes,
                               s->stream);
        for (int64_t k0 = 0; k0 < nnz && e == hipSuccess; k0 += kChunk) {
            const int64_t m = std::min(kChunk, nnz - k0);
            e = hipMemcpyAsync(d_rows, rows + k0, (size_t)m * 8, hipMemcpyHostToDevice, s->stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(d_cols, cols + k0, (size_t)m * 8, hipMemcpyHostToDevice, s->stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(d_vals, vals + k0, (size_t)m * 8, hipMemcpyHostToDevice, s->stream);
            if (e != hipSuccess) break;
            const unsigned grid = (unsigned)((m + 255) / 256);
#define BB_SCATTER(TT, WW)                                                                        \
    hipLaunchKernelGGL((scatter_entries_kernel<TT, WW>), dim3(grid), dim3(256), 0, s->stream, d_rows, \
                       d_cols, d_vals, m, d_map, nb, s->L.n_bins, s->u_begin, s->u_end,              \
                       (TT *)s->d_units, kind, -1.0 / alpha, d_kr, d_ke, d_bad)
            if (s->dtype == BB_F32) BB_SCATTER(float, true);
            else if (s->wide) BB_SCATTER(double, true);
            else BB_SCATTER(double, false);
#undef BB_SCATTER
            e = hipGetLastError();
            // the staging buffers are reused by the next chunk: stream order makes that safe
        }
        if (e == hipSuccess)
            e = hipMemcpyAsync(&host_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    }
    hipFree(d_map);
    hipFree(d_bad);
    hipFree(d_rows);
    hipFree(d_cols);
    hipFree(d_vals);
    hipFree(d_kr);
    hipFree(d_ke);
    if (rc != BB_OK) return rc;
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_solver_set_wish_sparse: ") + hipGetErrorString(e));
    if (host_bad == 1)
        return bb::fail(BB_ERR_INVALID, "bb_solver_set_wish_sparse: an index is outside [0, n_bins)");
    if (host_bad == 2)
        return bb::fail(BB_ERR_INVALID,
                        "bb_solver_set_wish_sparse: an entry falls in a tile that is not in the "
                        "solver's tile list");
    s->have_wish = true;
    return BB_OK;
}

int bb_solver_set_wish_from_coords(bb_solver *s, const double *xstar) {
    BB_REQUIRE(s != nullptr && xstar != nullptr, "bb_solver_set_wish_from_coords: NULL argument");
    BB_TRY(bb::enter_device(s->device));
    BB_HIP_CHECK(hipMemsetAsync(s->d_f64_tmp, 0, (size_t)s->L.n_pad * 3 * sizeof(double), s->stream));
    BB_HIP_CHECK(hipMemcpyAsync(s->d_f64_tmp, xstar, (size_t)s->L.n_bins * 3 * sizeof(double),
                                hipMemcpyHostToDevice, s->stream));
    if (s->n_local > 0) {
#define BB_GEN(TT, WW)                                                                          \
    hipLaunchKernelGGL((gen_units_kernel<TT, WW>), dim3((unsigned)s->n_local), dim3(256), 0, s->stream, \
                       s->d_f64_tmp, (TT *)s->d_units, s->d_udesc, s->L.n_bins)
        if (s->dtype == BB_F32) BB_GEN(float, true);
        else if (s->wide) BB_GEN(double, true);
        else BB_GEN(double, false);
#undef BB_GEN
        BB_HIP_CHECK(hipGetLastError());
    }
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    s->have_wish = true;
    return BB_OK;
}

int bb_solver_set_coords(bb_solver *s, const double *xyz) {
    BB_REQUIRE(s != nullptr && xyz != nullptr, "bb_solver_set_coords: NULL argument");
    BB_TRY(bb::enter_device(s->device));
    const int64_t n3 = s->L.n_pad * 3;
    BB_HIP_CHECK(hipMemsetAsync(s->d_f64_tmp, 0, (size_t)n3 * sizeof(double), s->stream));
    BB_HIP_CHECK(hipMemcpyAsync(s->d_f64_tmp, xyz, (size_t)s->L.n_bins * 3 * sizeof(double),
                                hipMemcpyHostToDevice, s->stream));
    const unsigned grid = (unsigned)((n3 + 255) / 256);
    if (s->dtype == BB_F32)
        hipLaunchKernelGGL(f64_to_T_kernel<float>, dim3(grid), dim3(256), 0, s->stream,
                           s->d_f64_tmp, (float *)s->d_X, n3);
    else
        hipLaunchKernelGGL(f64_to_T_kernel<double>, dim3(grid), dim3(256), 0, s->stream,
                           s->d_f64_tmp, (double *)s->d_X, n3);
    BB_HIP_CHECK(hipGetLastError());
    BB_HIP_CHECK(hipMemsetAsync(s->d_V, 0, (size_t)n3 * bb::elem_size(s->dtype), s->stream));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    s->have_coords = true;
    s->hist_n = 0;
    s->grad_pending = false;
    return BB_OK;
}

int bb_solver_get_coords(bb_solver *s, double *xyz) {
    BB_REQUIRE(s != nullptr && xyz != nullptr, "bb_solver_get_coords: NULL argument");
    if (!s->have_coords) return bb::fail(BB_ERR_STATE, "bb_solver_get_coords: no coordinates set");
    BB_TRY(bb::enter_device(s->device));
    const int64_t n3 = s->L.n_pad * 3;
    const unsigned grid = (unsigned)((n3 + 255) / 256);
    if (s->dtype == BB_F32)
        hipLaunchKernelGGL(T_to_f64_kernel<float>, dim3(grid), dim3(256), 0, s->stream,
                           (const float *)s->d_X, s->d_f64_tmp, n3);
    else
        hipLaunchKernelGGL(T_to_f64_kernel<double>, dim3(grid), dim3(256), 0, s->stream,
                           (const double *)s->d_X, s->d_f64_tmp, n3);
    BB_HIP_CHECK(hipGetLastError());
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    BB_HIP_CHECK(hipMemcpy(xyz, s->d_f64_tmp, (size_t)s->L.n_bins * 3 * sizeof(double),
                           hipMemcpyDeviceToHost));
    return BB_OK;
}

int bb_solver_set_momentum(bb_solver *s, double mu) {
    BB_REQUIRE(s != nullptr, "bb_solver_set_momentum: solver is NULL");
    BB_REQUIRE(mu >= 0.0 && mu < 1.0, "bb_solver_set_momentum: need 0 <= mu < 1");
    s->momentum = mu;
    return BB_OK;
}

int bb_solver_iterate(bb_solver *s, int64_t iters, double lr) {
    BB_TRY(check_ready(s, "bb_solver_iterate"));
    BB_REQUIRE(iters >= 0, "bb_solver_iterate: iters < 0");
    if (s->world != 1)
        return bb::fail(BB_ERR_STATE,
                        "bb_solver_iterate: world > 1 needs bb_solver_grad / all-reduce / "
                        "bb_solver_apply");
    if (s->hist_n + iters > s->hist_cap)
        return bb::fail(BB_ERR_STATE, "bb_solver_iterate: stress history full");
    BB_TRY(bb::enter_device(s->device));
    for (int64_t k = 0; k < iters; ++k) {
        hipEvent_t *ev = timing_slot(s);
        if (ev) BB_HIP_CHECK(hipEventRecord(ev[0], s->stream));
        BB_TRY(launch_grad(s));
        if (ev) BB_HIP_CHECK(hipEventRecord(ev[1], s->stream));
        BB_TRY(launch_reduce(s, kReduceApply, lr, s->d_stress_hist + s->hist_n));
        if (ev) BB_HIP_CHECK(hipEventRecord(ev[2], s->stream));
        s->hist_n++;
    }
    return BB_OK;
}

int bb_solver_grad(bb_solver *s) {
    BB_TRY(check_ready(s, "bb_solver_grad"));
    BB_TRY(bb::enter_device(s->device));
    hipEvent_t *ev = timing_slot(s);
    if (ev) BB_HIP_CHECK(hipEventRecord(ev[0], s->stream));
    BB_TRY(launch_grad(s));
    if (ev) BB_HIP_CHECK(hipEventRecord(ev[1], s->stream));
    BB_TRY(launch_reduce(s, kReduceExchange, 0.0, nullptr));
    if (ev) BB_HIP_CHECK(hipEventRecord(ev[2], s->stream));
    s->grad_pending = true;
    return BB_OK;
}

int bb_solver_apply(bb_solver *s, double lr) {
    BB_TRY(check_ready(s, "bb_solver_apply"));
    if (!s->grad_pending)
        return bb::fail(BB_ERR_STATE, "bb_solver_apply: no bb_solver_grad pending");
    if (s->hist_n + 1 > s->hist_cap)
        return bb::fail(BB_ERR_STATE, "bb_solver_apply: stress history full");
    BB_TRY(bb::enter_device(s->device));
    const int64_t n3 = s->L.n_pad * 3;
    const unsigned grid = (unsigned)((n3 + 255) / 256);
    if (s->dtype == BB_F32)
        hipLaunchKernelGGL(apply_kernel<float>, dim3(grid), dim3(256), 0, s->stream,
                           (float *)s->d_X, (float *)s->d_V, (const float *)s->d_exch, n3,
                           (float)lr, (float)s->momentum, s->d_stress_hist + s->hist_n);
    else
        hipLaunchKernelGGL(apply_kernel<double>, dim3(grid), dim3(256), 0, s->stream,
                           (double *)s->d_X, (double *)s->d_V, (const double *)s->d_exch, n3, lr,
                           s->momentum, s->d_stress_hist + s->hist_n);
    BB_HIP_CHECK(hipGetLastError());
    s->hist_n++;
    s->grad_pending = false;
    return BB_OK;
}

int bb_comm_unique_id(void *id_out) {
    BB_REQUIRE(id_out != nullptr, "bb_comm_unique_id: id_out is NULL");
    const bb::Rccl &R = bb::rccl();
    if (!R.ok) return bb::fail(BB_ERR_HIP, "bb_comm_unique_id: librccl is not loadable");
    bb::Rccl::UniqueId id;
    const int rc = R.GetUniqueId(&id);
    if (rc != bb::Rccl::kSuccess)
        return bb::fail(BB_ERR_HIP, std::string("ncclGetUniqueId: ") + R.GetErrorString(rc));
    memcpy(id_out, id.internal, bb::kUniqueIdBytes);
    return BB_OK;
}

int bb_solver_comm_init(bb_solver *s, const void *unique_id) {
    BB_REQUIRE(s != nullptr && unique_id != nullptr, "bb_solver_comm_init: NULL argument");
    if (s->comm) return bb::fail(BB_ERR_STATE, "bb_solver_comm_init: communicator already made");
    const bb::Rccl &R = bb::rccl();
    if (!R.ok) return bb::fail(BB_ERR_HIP, "bb_solver_comm_init: librccl is not loadable");
    BB_TRY(bb::enter_device(s->device));
    bb::Rccl::UniqueId id;
    memcpy(id.internal, unique_id, bb::kUniqueIdBytes);
    const int rc = R.CommInitRank(&s->comm, s->world, id, s->rank);
    if (rc != bb::Rccl::kSuccess) {
        s->comm = nullptr;
        return bb::fail(BB_ERR_HIP, std::string("ncclCommInitRank: ") + R.GetErrorString(rc));
    }
    return BB_OK;
}

namespace {
int enqueue_allreduce(bb_solver *s) {
    const bb::Rccl &R = bb::rccl();
    const int rc = R.AllReduce(s->d_exch, s->d_exch, (size_t)(3 * s->L.n_pad + 2),
                               s->dtype == BB_F32 ? bb::Rccl::kFloat32 : bb::Rccl::kFloat64,
                               bb::Rccl::kSum, s->comm, s->stream);
    if (rc != bb::Rccl::kSuccess)
        return bb::fail(BB_ERR_HIP, std::string("ncclAllReduce: ") + R.GetErrorString(rc));
    return BB_OK;
}
}  // namespace

int bb_solver_allreduce(bb_solver *s) {
    BB_REQUIRE(s != nullptr, "bb_solver_allreduce: solver is NULL");
    if (!s->comm) return bb::fail(BB_ERR_STATE, "bb_solver_allreduce: no communicator");
    if (!s->grad_pending) return bb::fail(BB_ERR_STATE, "bb_solver_allreduce: no bb_solver_grad pending");
    BB_TRY(bb::enter_device(s->device));
    return enqueue_allreduce(s);
}

int bb_solver_iterate_dist(bb_solver *s, int64_t iters, double lr) {
    BB_REQUIRE(iters >= 0, "bb_solver_iterate_dist: iters < 0");
    for (int64_t k = 0; k < iters; ++k) {
        BB_TRY(bb_solver_grad(s));
        BB_TRY(bb_solver_allreduce(s));
        BB_TRY(bb_solver_apply(s, lr));
    }
    return BB_OK;
}

int bb_solver_peer_export(bb_solver *s, void *handle_out) {
    BB_REQUIRE(s != nullptr && handle_out != nullptr, "bb_solver_peer_export: NULL argument");
    BB_REQUIRE(s->world <= kMaxPeers, "bb_solver_peer_export: world > 16");
    if (s->peer_arena) return bb::fail(BB_ERR_STATE, "bb_solver_peer_export: already exported");
    BB_TRY(bb::enter_device(s->device));
    const int64_t es = bb::elem_size(s->dtype);
    s->peer_slot_elems = bb::round_up(3 * s->L.n_pad + 2, 256 / es);
    s->peer_arena_bytes = peer_flags_offset(s) + (int64_t)s->world * 64;
    // Uncached: written by the peers' kernels while ours is running, so nothing of
    // it may live in this GPU's L2.  (RCCL allocates its own buffers the same way.)
    hipError_t e = hipExtMallocWithFlags(&s->peer_arena, (size_t)s->peer_arena_bytes,
                                         hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&s->peer_arena, (size_t)s->peer_arena_bytes,
                                  hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) {
        s->peer_arena = nullptr;
        return bb::fail(BB_ERR_NOMEM, std::string("bb_solver_peer_export: arena: ") +
                                          hipGetErrorString(e));
    }
    BB_HIP_CHECK(hipMemset(s->peer_arena, 0, (size_t)s->peer_arena_bytes));
    BB_HIP_CHECK(hipDeviceSynchronize());
    PeerHandle h;
    memset(&h, 0, sizeof(h));
    e = hipIpcGetMemHandle(&h.ipc, s->peer_arena);
    if (e != hipSuccess) {
        hipFree(s->peer_arena);
        s->peer_arena = nullptr;
        return bb::fail(BB_ERR_HIP, std::string("bb_solver_peer_export: hipIpcGetMemHandle: ") +
                                        hipGetErrorString(e));
    }
    h.magic = kPeerMagic;
    h.arena_bytes = s->peer_arena_bytes;
    h.slot_elems = s->peer_slot_elems;
    h.pid = (int64_t)getpid();
    h.raw = (uint64_t)(uintptr_t)s->peer_arena;
    h.rank = s->rank;
    h.world = s->world;
    h.dtype = s->dtype;
    h.device = s->device;
    memset(handle_out, 0, BB_PEER_HANDLE_BYTES);
    memcpy(handle_out, &h, sizeof(h));
    return BB_OK;
}

int bb_solver_peer_connect(bb_solver *s, const void *handles) {
    BB_REQUIRE(s != nullptr && handles != nullptr, "bb_solver_peer_connect: NULL argument");
    if (!s->peer_arena) return bb::fail(BB_ERR_STATE, "bb_solver_peer_connect: export first");
    if (s->peer_connected) return bb::fail(BB_ERR_STATE, "bb_solver_peer_connect: already connected");
    BB_TRY(bb::enter_device(s->device));
    s->peer_mapped.assign((size_t)s->world, nullptr);
    for (int r = 0; r < s->world; ++r) {
        PeerHandle h;
        memcpy(&h, (const char *)handles + (size_t)r * BB_PEER_HANDLE_BYTES, sizeof(h));
        if (h.magic != kPeerMagic || h.rank != r || h.world != s->world || h.dtype != s->dtype ||
            h.arena_bytes != s->peer_arena_bytes || h.slot_elems != s->peer_slot_elems)
            return bb::fail(BB_ERR_INVALID, "bb_solver_peer_connect: handle " + std::to_string(r) +
                                                " does not match this solver (rank order, world, "
                                                "dtype and n_bins must agree)");
        if (r == s->rank) {
            s->peer_mapped[r] = s->peer_arena;
        } else if (h.pid == (int64_t)getpid()) {
            // same process: the exporter's pointer is valid here, but only from the
            // same device or with peer access
            if (h.device != s->device) {
                hipError_t pe = hipDeviceEnablePeerAccess(h.device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                    return bb::fail(BB_ERR_HIP, std::string("bb_solver_peer_connect: peer access: ") +
                                                    hipGetErrorString(pe));
                (void)hipGetLastError();
            }
            s->peer_mapped[r] = (void *)(uintptr_t)h.raw;
        } else {
            void *ptr = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&ptr, h.ipc, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess)
                return bb::fail(BB_ERR_HIP, "bb_solver_peer_connect: hipIpcOpenMemHandle(rank " +
                                                std::to_string(r) + "): " + hipGetErrorString(e));
            s->peer_mapped[r] = ptr;
            s->peer_opened.push_back(ptr);
        }
    }
    BB_TRY(s->dtype == BB_F32 ? build_peer_tables<float>(s) : build_peer_tables<double>(s));
    BB_TRY(dev_alloc(&s->d_peer_status, 1));
    BB_TRY(dev_alloc(&s->d_peer_counter, 1));
    BB_HIP_CHECK(hipMemset(s->d_peer_status, 0, sizeof(int)));
    BB_HIP_CHECK(hipMemset(s->d_peer_counter, 0, sizeof(unsigned)));
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, s->device) != hipSuccess || khz <= 0)
        khz = 100000;  // gfx9: 100 MHz
    (void)hipGetLastError();
    long long ms = 10000;
    if (const char *env = getenv("BB_PEER_TIMEOUT_MS")) {
        const long long v = atoll(env);
        if (v > 0) ms = v;
    }
    s->peer_limit_ticks = ms * khz;
    s->peer_seq = 0;
    s->peer_connected = true;
    BB_HIP_CHECK(hipDeviceSynchronize());
    return BB_OK;
}

int bb_solver_peer_status(bb_solver *s, int *status) {
    BB_REQUIRE(s != nullptr, "bb_solver_peer_status: solver is NULL");
    if (!s->peer_connected) return bb::fail(BB_ERR_STATE, "bb_solver_peer_status: not connected");
    BB_TRY(bb::enter_device(s->device));
    int st = 0;
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    BB_HIP_CHECK(hipMemcpy(&st, s->d_peer_status, sizeof(int), hipMemcpyDeviceToHost));
    if (status) *status = st;
    if (st != 0)
        return bb::fail(BB_ERR_STATE,
                        "peer exchange: a rank did not deliver its partial within the time limit "
                        "(BB_PEER_TIMEOUT_MS); coordinates were left at the last completed step");
    return BB_OK;
}

int bb_solver_iterate_peer(bb_solver *s, int64_t iters, double lr) {
    BB_TRY(check_ready(s, "bb_solver_iterate_peer"));
    BB_REQUIRE(iters >= 0, "bb_solver_iterate_peer: iters < 0");
    if (!s->peer_connected)
        return bb::fail(BB_ERR_STATE, "bb_solver_iterate_peer: bb_solver_peer_connect first");
    if (s->grad_pending)
        return bb::fail(BB_ERR_STATE, "bb_solver_iterate_peer: a bb_solver_grad is pending");
    if (s->hist_n + iters > s->hist_cap)
        return bb::fail(BB_ERR_STATE, "bb_solver_iterate_peer: stress history full");
    BB_TRY(bb::enter_device(s->device));
    const int64_t n3 = s->L.n_pad * 3, es = bb::elem_size(s->dtype);
    const unsigned grid = (unsigned)((n3 + 255) / 256);
    for (int64_t k = 0; k < iters; ++k) {
        hipEvent_t *ev = timing_slot(s);
        if (ev) BB_HIP_CHECK(hipEventRecord(ev[0], s->stream));
        BB_TRY(launch_grad(s));
        if (ev) BB_HIP_CHECK(hipEventRecord(ev[1], s->stream));
        s->peer_seq++;
        BB_TRY(launch_reduce(s, kReducePeer, 0.0, nullptr));
        if (ev) BB_HIP_CHECK(hipEventRecord(ev[2], s->stream));
        const char *arena = (const char *)s->peer_arena +
                            (int64_t)(s->peer_seq & 1) * s->world * s->peer_slot_elems * es;
        const unsigned long long *flags =
            (const unsigned long long *)((const char *)s->peer_arena + peer_flags_offset(s));
        if (s->dtype == BB_F32)
            hipLaunchKernelGGL(peer_apply_kernel<float>, dim3(grid), dim3(256), 0, s->stream,
                               (float *)s->d_X, (float *)s->d_V, (const float *)arena, flags,
                               s->world, s->peer_slot_elems, n3, (float)lr, (float)s->momentum,
                               s->d_stress_hist + s->hist_n, s->peer_seq, s->d_peer_status,
                               s->peer_limit_ticks);
        else
            hipLaunchKernelGGL(peer_apply_kernel<double>, dim3(grid), dim3(256), 0, s->stream,
                               (double *)s->d_X, (double *)s->d_V, (const double *)arena, flags,
                               s->world, s->peer_slot_elems, n3, lr, s->momentum,
                               s->d_stress_hist + s->hist_n, s->peer_seq, s->d_peer_status,
                               s->peer_limit_ticks);
        BB_HIP_CHECK(hipGetLastError());
        s->hist_n++;
    }
    return BB_OK;
}

int bb_solver_exchange_size(const bb_solver *s, int64_t *n_elems) {
    BB_REQUIRE(s != nullptr && n_elems != nullptr, "bb_solver_exchange_size: NULL argument");
    *n_elems = 3 * s->L.n_pad + 2;
    return BB_OK;
}

int bb_solver_get_exchange_buffer(bb_solver *s, void **dev_ptr) {
    BB_REQUIRE(s != nullptr && dev_ptr != nullptr, "bb_solver_get_exchange_buffer: NULL argument");
    *dev_ptr = s->d_exch;
    return BB_OK;
}

int bb_solver_set_exchange_buffer(bb_solver *s, void *dev_ptr) {
    BB_REQUIRE(s != nullptr && dev_ptr != nullptr, "bb_solver_set_exchange_buffer: NULL argument");
    BB_TRY(bb::enter_device(s->device));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    if (s->own_exch) hipFree(s->d_exch);
    s->d_exch = dev_ptr;
    s->own_exch = false;
    return BB_OK;
}

int bb_solver_read_exchange(bb_solver *s, double *host, int64_t n) {
    BB_REQUIRE(s != nullptr && host != nullptr, "bb_solver_read_exchange: NULL argument");
    BB_REQUIRE(n == 3 * s->L.n_pad + 2, "bb_solver_read_exchange: n != exchange size");
    BB_TRY(bb::enter_device(s->device));
    double *tmp = nullptr;
    BB_TRY(dev_alloc(&tmp, n));
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (s->dtype == BB_F32)
        hipLaunchKernelGGL(T_to_f64_kernel<float>, dim3(grid), dim3(256), 0, s->stream,
                           (const float *)s->d_exch, tmp, n);
    else
        hipLaunchKernelGGL(T_to_f64_kernel<double>, dim3(grid), dim3(256), 0, s->stream,
                           (const double *)s->d_exch, tmp, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(host, tmp, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    hipFree(tmp);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_solver_read_exchange: ") + hipGetErrorString(e));
    return BB_OK;
}

int bb_solver_write_exchange(bb_solver *s, const double *host, int64_t n) {
    BB_REQUIRE(s != nullptr && host != nullptr, "bb_solver_write_exchange: NULL argument");
    BB_REQUIRE(n == 3 * s->L.n_pad + 2, "bb_solver_write_exchange: n != exchange size");
    BB_TRY(bb::enter_device(s->device));
    double *tmp = nullptr;
    BB_TRY(dev_alloc(&tmp, n));
    hipError_t e = hipMemcpyAsync(tmp, host, (size_t)n * sizeof(double), hipMemcpyHostToDevice,
                                  s->stream);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)((n + 255) / 256);
        if (s->dtype == BB_F32)
            hipLaunchKernelGGL(f64_to_T_kernel<float>, dim3(grid), dim3(256), 0, s->stream, tmp,
                               (float *)s->d_exch, n);
        else
            hipLaunchKernelGGL(f64_to_T_kernel<double>, dim3(grid), dim3(256), 0, s->stream, tmp,
                               (double *)s->d_exch, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    hipFree(tmp);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_solver_write_exchange: ") + hipGetErrorString(e));
    return BB_OK;
}

int bb_solver_matvec_sq(bb_solver *s, const double *x, double *y) {
    BB_REQUIRE(s != nullptr && x != nullptr && y != nullptr, "bb_solver_matvec_sq: NULL argument");
    if (!s->have_wish) return bb::fail(BB_ERR_STATE, "bb_solver_matvec_sq: no wish distances set");
    if (s->grad_pending)
        return bb::fail(BB_ERR_STATE, "bb_solver_matvec_sq: a bb_solver_grad is pending");
    BB_TRY(bb::enter_device(s->device));
    const int64_t n3 = s->L.n_pad * 3, es = bb::elem_size(s->dtype);
    void *d_in = nullptr;
    BB_TRY(dev_alloc((char **)&d_in, n3 * es));
    hipError_t e = hipMemsetAsync(s->d_f64_tmp, 0, (size_t)n3 * 8, s->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(s->d_f64_tmp, x, (size_t)s->L.n_bins * 24, hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)((n3 + 255) / 256);
        if (s->dtype == BB_F32)
            hipLaunchKernelGGL(f64_to_T_kernel<float>, dim3(grid), dim3(256), 0, s->stream,
                               s->d_f64_tmp, (float *)d_in, n3);
        else
            hipLaunchKernelGGL(f64_to_T_kernel<double>, dim3(grid), dim3(256), 0, s->stream,
                               s->d_f64_tmp, (double *)d_in, n3);
        e = hipGetLastError();
    }
    int rc = BB_OK;
    if (e != hipSuccess) rc = bb::fail(BB_ERR_HIP, std::string("bb_solver_matvec_sq: ") + hipGetErrorString(e));
    if (rc == BB_OK) rc = launch_grad(s, kOpMatvec2, d_in);
    if (rc == BB_OK) rc = launch_reduce(s, kReduceExchange, 0.0, nullptr, 1.0);
    if (rc == BB_OK) {
        const unsigned grid = (unsigned)((n3 + 255) / 256);
        if (s->dtype == BB_F32)
            hipLaunchKernelGGL(T_to_f64_kernel<float>, dim3(grid), dim3(256), 0, s->stream,
                               (const float *)s->d_exch, s->d_f64_tmp, n3);
        else
            hipLaunchKernelGGL(T_to_f64_kernel<double>, dim3(grid), dim3(256), 0, s->stream,
                               (const double *)s->d_exch, s->d_f64_tmp, n3);
        e = hipGetLastError();
        if (e == hipSuccess)
            e = hipMemcpyAsync(y, s->d_f64_tmp, (size_t)s->L.n_bins * 24, hipMemcpyDeviceToHost,
                               s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        if (e != hipSuccess)
            rc = bb::fail(BB_ERR_HIP, std::string("bb_solver_matvec_sq: ") + hipGetErrorString(e));
    }
    hipStreamSynchronize(s->stream);
    hipFree(d_in);
    return rc;
}

int bb_solver_stress(bb_solver *s, double *stress) {
    BB_TRY(check_ready(s, "bb_solver_stress"));
    BB_REQUIRE(stress != nullptr, "bb_solver_stress: stress is NULL");
    BB_TRY(bb::enter_device(s->device));
    BB_TRY(launch_grad(s));
    BB_TRY(launch_reduce(s, kReduceStressOnly, 0.0, s->d_stress_scalar));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    BB_HIP_CHECK(hipMemcpy(stress, s->d_stress_scalar, sizeof(double), hipMemcpyDeviceToHost));
    return BB_OK;
}

int bb_solver_get_stress_history(bb_solver *s, double *out, int64_t cap, int64_t *n) {
    BB_REQUIRE(s != nullptr && n != nullptr, "bb_solver_get_stress_history: NULL argument");
    BB_TRY(bb::enter_device(s->device));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    *n = s->hist_n;
    const int64_t m = std::min(cap, s->hist_n);
    if (out && m > 0)
        BB_HIP_CHECK(hipMemcpy(out, s->d_stress_hist, (size_t)m * sizeof(double),
                               hipMemcpyDeviceToHost));
    return BB_OK;
}

int bb_solver_sync(bb_solver *s) {
    BB_REQUIRE(s != nullptr, "bb_solver_sync: solver is NULL");
    BB_TRY(bb::enter_device(s->device));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    return BB_OK;
}

int bb_solver_set_timing(bb_solver *s, int enabled) {
    BB_REQUIRE(s != nullptr, "bb_solver_set_timing: solver is NULL");
    BB_TRY(bb::enter_device(s->device));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    if (enabled && s->ev.empty()) {
        s->ev.resize(3 * kMaxTimedLaunches);
        for (auto &e : s->ev) BB_HIP_CHECK(hipEventCreate(&e));
    }
    s->timing = enabled != 0;
    s->timing_stride = enabled > 1 ? enabled : 1;
    s->timing_iter = 0;
    s->ev_used = 0;
    return BB_OK;
}

int bb_solver_get_timing(bb_solver *s, double *grad_ms_avg, double *reduce_ms_avg,
                         int64_t *launches) {
    BB_REQUIRE(s != nullptr, "bb_solver_get_timing: solver is NULL");
    BB_TRY(bb::enter_device(s->device));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    double g = 0.0, r = 0.0;
    const size_t n = s->ev_used / 3;
    for (size_t k = 0; k < n; ++k) {
        float a = 0.f, b = 0.f;
        BB_HIP_CHECK(hipEventElapsedTime(&a, s->ev[3 * k], s->ev[3 * k + 1]));
        BB_HIP_CHECK(hipEventElapsedTime(&b, s->ev[3 * k + 1], s->ev[3 * k + 2]));
        g += a;
        r += b;
    }
    if (grad_ms_avg) *grad_ms_avg = n ? g / n : 0.0;
    if (reduce_ms_avg) *reduce_ms_avg = n ? r / n : 0.0;
    if (launches) *launches = (int64_t)n;
    return BB_OK;
}

int bb_solver_get_step_timing(bb_solver *s, double *step_ms_avg) {
    BB_REQUIRE(s != nullptr && step_ms_avg != nullptr, "bb_solver_get_step_timing: NULL argument");
    BB_TRY(bb::enter_device(s->device));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    const size_t n = s->ev_used / 3;
    double t = 0.0;
    for (size_t k = 0; k + 1 < n; ++k) {
        float a = 0.f;
        BB_HIP_CHECK(hipEventElapsedTime(&a, s->ev[3 * k], s->ev[3 * k + 3]));
        t += a;
    }
    *step_ms_avg = n > 1 ? t / (double)(n - 1) / s->timing_stride : 0.0;
    return BB_OK;
}

int bb_solver_measure_stream_read(bb_solver *s, int launches, double *ms_avg) {
    BB_REQUIRE(s != nullptr && ms_avg != nullptr, "bb_solver_measure_stream_read: NULL argument");
    BB_REQUIRE(launches >= 1 && launches <= 1000, "bb_solver_measure_stream_read: bad launches");
    if (!s->have_wish)
        return bb::fail(BB_ERR_STATE, "bb_solver_measure_stream_read: no wish distances set");
    BB_TRY(bb::enter_device(s->device));
    hipEvent_t e0, e1;
    BB_HIP_CHECK(hipEventCreate(&e0));
    BB_HIP_CHECK(hipEventCreate(&e1));
    auto launch = [&]() {
        if (s->nontemporal)
            hipLaunchKernelGGL((stream_read_kernel<true>), dim3(s->n_waves / 4), dim3(256), 0,
                               s->stream, (const float4 *)s->d_units, s->d_wave_range,
                               (float *)s->d_f64_tmp);
        else
            hipLaunchKernelGGL((stream_read_kernel<false>), dim3(s->n_waves / 4), dim3(256), 0,
                               s->stream, (const float4 *)s->d_units, s->d_wave_range,
                               (float *)s->d_f64_tmp);
    };
    launch();  // warm-up
    BB_HIP_CHECK(hipEventRecord(e0, s->stream));
    for (int k = 0; k < launches; ++k) launch();
    BB_HIP_CHECK(hipEventRecord(e1, s->stream));
    BB_HIP_CHECK(hipGetLastError());
    BB_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    BB_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *ms_avg = ms / launches;
    return BB_OK;
}

int bb_solver_traffic(const bb_solver *s, int64_t *unit_bytes, int64_t *pairs_dense) {
    BB_REQUIRE(s != nullptr, "bb_solver_traffic: solver is NULL");
    if (unit_bytes) *unit_bytes = s->n_local * bb::kUnitBytes;
    if (pairs_dense) *pairs_dense = s->L.n_bins * (s->L.n_bins - 1) / 2;
    return BB_OK;
}

}  // extern "C"
