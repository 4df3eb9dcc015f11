// bb_ablate.h -- timing-only ablation switches of the solver kernels.  NEVER defined in
// the product build (build.sh): each BB_ABL_* macro removes one part of
// stress_grad_kernel so that tools/tools_abl.sh can price it (the results are wrong by
// design); BB_WAVE_TRACE keeps the results and adds five time stamps per wave.  The kernels test the constexpr flags below, so the product source carries
// no #ifdef and the product object no trace of them.
#pragma once

namespace abl {
#define BB_ABL_FLAG(name, macro) constexpr bool name = macro
#ifdef BB_ABL_NODPP          // no cross-lane reduction of the row sums
BB_ABL_FLAG(kNoDpp, true);
#else
BB_ABL_FLAG(kNoDpp, false);
#endif
#ifdef BB_ABL_NORSQ          // a multiply instead of v_rsq_f32
BB_ABL_FLAG(kNoRsq, true);
#else
BB_ABL_FLAG(kNoRsq, false);
#endif
#ifdef BB_ABL_NOMASK         // no "no constraint" weight
BB_ABL_FLAG(kNoMask, true);
#else
BB_ABL_FLAG(kNoMask, false);
#endif
#ifdef BB_ABL_NOCOL          // no column-side accumulation
BB_ABL_FLAG(kNoCol, true);
#else
BB_ABL_FLAG(kNoCol, false);
#endif
#ifdef BB_ABL_NOSTORE        // row sums are not stored
BB_ABL_FLAG(kNoStore, true);
#else
BB_ABL_FLAG(kNoStore, false);
#endif
#ifdef BB_ABL_XROW_VECTOR    // row coordinates by per-lane load + v_readlane, not scalar loads
BB_ABL_FLAG(kXrowVector, true);
#else
BB_ABL_FLAG(kXrowVector, false);
#endif
#ifdef BB_ABL_F64_LIBM       // fp64: library sqrt + IEEE divide instead of rsq + Newton
BB_ABL_FLAG(kF64Libm, true);
#else
BB_ABL_FLAG(kF64Libm, false);
#endif
#ifdef BB_ABL_F64_GENERIC    // fp64 2 x 512 units through the generic unit (DPP row reduction,
BB_ABL_FLAG(kF64Generic, true);   // per-row stores) -- run it with BB_DEFER_ROWS=0
#else
BB_ABL_FLAG(kF64Generic, false);
#endif
#ifdef BB_WAVE_TRACE          // diagnostic build: per-wave time stamps (tools/wave_trace.py)
BB_ABL_FLAG(kWaveTrace, true);
#else
BB_ABL_FLAG(kWaveTrace, false);
#endif
#ifdef BB_UNIT_TRACE          // diagnostic build: a time stamp at the top of EVERY unit of a wave
BB_ABL_FLAG(kUnitTrace, true);    // (kept in LDS while the wave sweeps; tools/unit_trace.py) -- the
#else                             // unit loop is the product's: no peeled unit, no store in it
BB_ABL_FLAG(kUnitTrace, false);
#endif
constexpr int kUnitTraceSlots = 320;   // stamps kept per wave
#ifdef BB_BUFFER_WINDOW        // A/B: the window refilled by range-checked buffer loads
BB_ABL_FLAG(kGlobalWindow, false);
#else
BB_ABL_FLAG(kGlobalWindow, true);
#endif
#ifdef BB_MFMA_ROWSUM          // A/B: fp32 row sums through v_mfma_f32_16x16x4_f32 (as the fp64 unit
BB_ABL_FLAG(kDppRowSum, false);   // does) instead of the 64-lane DPP tree: 2.7-6 % SLOWER at every
#else                             // size (profiles/r03_mfma_rowsum_ab.txt), so not the product
BB_ABL_FLAG(kDppRowSum, true);
#endif
#ifdef BB_SCHED_REFILL         // experiment: sched_group_barrier, one refill per n VALU instructions
constexpr int kSchedRefill = BB_SCHED_REFILL;
#else
constexpr int kSchedRefill = 0;
#endif
#ifdef BB_DPP_ROWSUM           // A/B: fp32 row sums by the 64-lane DPP tree of rounds 1-2 (18 VALU
BB_ABL_FLAG(kSwapRowSum, false);  // instructions per row) instead of the lane swaps (10):
#else                             // profiles/r03_swap_rowsum_ab.txt
BB_ABL_FLAG(kSwapRowSum, true);
#endif
#ifdef BB_ABL_NOREFILL_FIRST  // the first unit of a strip does not refill the window (wrong results:
BB_ABL_FLAG(kNoRefillFirst, true);   // is that unit slow because its refills cannot be issued?)
#else
BB_ABL_FLAG(kNoRefillFirst, false);
#endif
#ifdef BB_PEEL_FIRST_UNIT      // A/B: round 1's loop shape, the first unit of a strip peeled
BB_ABL_FLAG(kPeelFirstUnit, true);
#else
BB_ABL_FLAG(kPeelFirstUnit, false);
#endif
#ifdef BB_WARMUP               // experiment: n x 4 dummy packed FMAs per wave while its first loads
constexpr int kWarmup = BB_WARMUP;   // are in flight (does the first unit run slow because the VALUs start cold?)
#else
constexpr int kWarmup = 0;
#endif
#undef BB_ABL_FLAG
}  // namespace abl
