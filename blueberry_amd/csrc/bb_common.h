// bb_common.h -- internal helpers shared by the translation units of
// libblueberry_hip.so (error state, HIP status checks, layout arithmetic).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <tuple>
#include <utility>

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

#define BB_TRY(expr)                  \
    do {                              \
        int _rc = (expr);             \
        if (_rc != BB_OK) return _rc; \
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
// Up to this many bins a one-rank solver iterates on the row-owner path (both
// triangles resident, one wave per bin, one launch per iteration: DESIGN.md 4.10).
constexpr int64_t kRowOwnerMaxBins = 4096;

inline int64_t elem_size(int dtype) { return dtype == BB_F64 ? 8 : 4; }
inline bool wide_layout(int dtype, int64_t n_bins) { return dtype != BB_F64 || n_bins > kF64WideFrom; }
inline int64_t tile_width(int dtype, int64_t n_bins) { return wide_layout(dtype, n_bins) ? 512 : 128; }
inline int64_t rows_per_unit(int dtype, int64_t n_bins) {
    return dtype != BB_F64 ? 4 : (wide_layout(dtype, n_bins) ? 2 : 8);
}
inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// Select the device, failing loudly when there is none.
int use_device(int device);
// Entry of every call on an existing handle: select its device.  Nothing else: the
// thread's sticky "last error" is left alone (see launch() below).
int enter_device(int device);

// Non-blocking streams are handed out from a small per-device pool: creating one costs
// 1.3-3 ms and destroying one 1.6 ms on this platform (tools/api_cost_probe.py) -- together
// two thirds of a whole chr21-sized fit.  acquire gives an idle stream of the CURRENT
// device (a pooled one, or a new one); release synchronises it and keeps up to
// kPooledStreams per device for the next handle.
hipError_t acquire_stream(int device, hipStream_t *out);
void release_stream(int device, hipStream_t stream);

// Kernel launch that hands back the launch's OWN status (hipLaunchKernel returns it).
// The library never reads hipGetLastError(): that word is per-thread, sticky, and shared
// with every other HIP user of the calling thread (torch, RCCL), so a launch check made
// through it can report somebody else's old failure -- or, if it is cleared first, hide
// one.  Every HIP call here is checked by its own return value instead; the few probes
// that are allowed to fail (an attribute an older runtime lacks, a memory flavour the
// device does not offer) consume their error on the spot with hipGetLastError().
template <typename... P, size_t... I>
inline hipError_t launch_impl(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds,
                              hipStream_t stream, std::tuple<P...> &vals,
                              std::index_sequence<I...>) {
    void *ptrs[sizeof...(P) ? sizeof...(P) : 1] = {(void *)&std::get<I>(vals)...};
    return hipLaunchKernel((const void *)kernel, grid, block, ptrs, lds, stream);
}
template <typename... P, typename... A>
inline hipError_t launch(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds,
                         hipStream_t stream, A &&...args) {
    static_assert(sizeof...(P) == sizeof...(A), "kernel argument count");
    std::tuple<P...> vals{static_cast<P>(std::forward<A>(args))...};
    return launch_impl(kernel, grid, block, lds, stream, vals, std::index_sequence_for<P...>{});
}

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
