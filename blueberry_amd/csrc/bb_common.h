// bb_common.h -- internal helpers shared by the translation units of
// libblueberry_hip.so (error state, HIP status checks, layout arithmetic).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "blueberry_hip.h"

namespace bb {

// Thread-local last-error message behind bb_last_error().
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

#define BB_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess)                                                           \
            return ::bb::fail(BB_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define BB_REQUIRE(cond, msg)                                   \
    do {                                                        \
        if (!(cond)) return ::bb::fail(BB_ERR_INVALID, (msg));  \
    } while (0)

// ---- layout constants (docs/SPEC.md 3) -----------------------------------
// A unit is always 8 KiB = 8 wave-loads of 64 lanes x 16 B.
//   fp32: 4 matrix rows x 512 columns (2 loads per row)
//   fp64: 2 rows x 512 columns (4 loads per row) above kF64WideFrom bins, where the
//         sweep is what matters; 8 rows x 128 columns (1 load per row) up to it,
//         where small column partials and many blocks matter (launch-bound regime).
// Crossover measured on MI355X: N=2,500 is 24 us per iteration narrow / 30 wide,
// N=6,000 is 64 narrow / 52 wide.
constexpr int kUnitBytes = 8192;
constexpr int64_t kF64WideFrom = 4096;

inline int64_t elem_size(int dtype) { return dtype == BB_F64 ? 8 : 4; }
inline bool wide_layout(int dtype, int64_t n_bins) { return dtype != BB_F64 || n_bins > kF64WideFrom; }
inline int64_t tile_width(int dtype, int64_t n_bins) { return wide_layout(dtype, n_bins) ? 512 : 128; }
inline int64_t rows_per_unit(int dtype, int64_t n_bins) {
    return dtype != BB_F64 ? 4 : (wide_layout(dtype, n_bins) ? 2 : 8);
}
inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// Select the device, failing loudly when there is none.
int use_device(int device);
// Entry of every call on an existing solver: select its device and drop whatever
// error an earlier HIP call of this thread left behind (hipGetLastError is
// per-thread and sticky, and the thread is shared with the caller's other HIP
// users -- torch, RCCL), so that a launch check reports our launch only.
int enter_device(int device);

// A device allocation that frees itself (host-staged entry points).
struct DevBuf {
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
};

}  // namespace bb
