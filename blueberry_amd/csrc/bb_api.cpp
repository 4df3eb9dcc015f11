// bb_api.cpp -- version, error state, device probe and the host-only layout
// arithmetic of libblueberry_hip.so (include/blueberry_hip.h).
#include "bb_common.h"

#include <map>
#include <mutex>
#include <vector>

namespace bb {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

int use_device(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(BB_ERR_HIP, std::string("no usable HIP device (hipGetDeviceCount: ") +
                                    hipGetErrorString(e) + ")");
    if (device < 0 || device >= count)
        return fail(BB_ERR_INVALID, "device index out of range");
    BB_HIP_CHECK(hipSetDevice(device));
    return BB_OK;
}

int enter_device(int device) {
    BB_HIP_CHECK(hipSetDevice(device));
    return BB_OK;
}

namespace {
constexpr size_t kPooledStreams = 8;
std::mutex g_stream_mu;
std::map<int, std::vector<hipStream_t>> g_streams;
}  // namespace

hipError_t acquire_stream(int device, hipStream_t *out) {
    {
        std::lock_guard<std::mutex> lock(g_stream_mu);
        auto &v = g_streams[device];
        if (!v.empty()) {
            *out = v.back();
            v.pop_back();
            return hipSuccess;
        }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void release_stream(int device, hipStream_t stream) {
    if (!stream) return;
    if (hipStreamSynchronize(stream) == hipSuccess) {
        std::lock_guard<std::mutex> lock(g_stream_mu);
        auto &v = g_streams[device];
        if (v.size() < kPooledStreams) {
            v.push_back(stream);
            return;
        }
    }
    (void)hipStreamDestroy(stream);    // pool full, or the stream is in an error state
}

}  // namespace bb

extern "C" {

int bb_version(void) { return BB_VERSION; }

const char *bb_last_error(void) { return bb::g_last_error.c_str(); }

int bb_device_count(int *count) {
    BB_REQUIRE(count != nullptr, "bb_device_count: count is NULL");
    *count = 0;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0)
        return bb::fail(BB_ERR_HIP, std::string("no usable HIP device (hipGetDeviceCount: ") +
                                        hipGetErrorString(e) + ")");
    *count = c;
    return BB_OK;
}

int bb_layout_dense_info(int64_t n_bins, int dtype, bb_layout_info *info) {
    BB_REQUIRE(info != nullptr, "bb_layout_dense_info: info is NULL");
    BB_REQUIRE(n_bins >= 1, "bb_layout_dense_info: n_bins must be >= 1");
    BB_REQUIRE(dtype == BB_F32 || dtype == BB_F64, "bb_layout_dense_info: bad dtype");
    const int64_t vw = bb::tile_width(dtype, n_bins);
    info->n_bins = n_bins;
    info->vw = vw;
    info->n_pad = bb::round_up(n_bins, vw);
    info->rows_per_unit = bb::rows_per_unit(dtype, n_bins);
    info->units_per_tile = vw / info->rows_per_unit;
    info->n_blocks = info->n_pad / vw;
    info->n_tiles = info->n_blocks * (info->n_blocks + 1) / 2;
    info->n_units = info->n_tiles * info->units_per_tile;
    return BB_OK;
}

int bb_layout_dense_tiles(int64_t n_bins, int dtype, int32_t *tile_I, int32_t *tile_J,
                          int64_t cap) {
    bb_layout_info info;
    int rc = bb_layout_dense_info(n_bins, dtype, &info);
    if (rc != BB_OK) return rc;
    BB_REQUIRE(tile_I != nullptr && tile_J != nullptr, "bb_layout_dense_tiles: NULL output");
    BB_REQUIRE(cap >= info.n_tiles, "bb_layout_dense_tiles: cap < n_tiles");
    int64_t t = 0;
    for (int64_t J = 0; J < info.n_blocks; ++J)
        for (int64_t I = 0; I <= J; ++I) {
            tile_I[t] = (int32_t)I;
            tile_J[t] = (int32_t)J;
            ++t;
        }
    return BB_OK;
}

int bb_layout_rank_units(int64_t n_units, int rank, int world, int64_t *u_begin,
                         int64_t *u_end) {
    BB_REQUIRE(u_begin != nullptr && u_end != nullptr, "bb_layout_rank_units: NULL output");
    BB_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bb_layout_rank_units: bad rank/world");
    BB_REQUIRE(n_units >= 0, "bb_layout_rank_units: n_units < 0");
    // floor(n*r/world) without overflow for n up to 2^62/world.
    *u_begin = (int64_t)((__int128)n_units * rank / world);
    *u_end = (int64_t)((__int128)n_units * (rank + 1) / world);
    return BB_OK;
}

}  // extern "C"
