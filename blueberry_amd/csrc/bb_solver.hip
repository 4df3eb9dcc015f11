// bb_solver.hip -- host side of the 3D-structure solver of libblueberry_hip.so: the
// bb_solver handle (layout, partition into ranks / waves / partial-sum slots, device
// buffers), kernel launches, the multi-rank exchanges (RCCL, peer arenas) and the
// bb_solver_* / bb_comm_* entry points of include/blueberry_hip.h.  The kernels are in
// bb_solver_kernels.h.  gfx950 only.
//
// Specification: docs/SPEC.md (build-authored; the reference has no solver,
// SURVEY.md section 0).  Data layout and kernel design: DESIGN.md 3-4.
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

#include <map>
#include <mutex>
#include <tuple>

#include "bb_comm.h"
#include "bb_common.h"
#include "bb_solver_kernels.h"

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
    bool stream_stuck = false;     // behind a collective that cannot be aborted: never waited on again

    void *d_units = nullptr, *d_X = nullptr, *d_V = nullptr, *d_part = nullptr, *d_exch = nullptr;
    double momentum = 0.0;
    bool own_exch = false;
    int2 *d_udesc = nullptr;
    int chunk_q = 0, chunk_r = 0;  // units per wave: n_local = n_waves * q + r
    bool nontemporal = true;
    char *d_arena = nullptr;       // the one allocation behind every d_* below but d_exch / peer / mv_in
    int2 *d_wave_slots = nullptr;  // per wave {first private slot, the workgroup's shared slot}
    int lds_wave_floats = 0;       // LDS region per wave of the sweep, in 4-byte words
    double *d_stresspart = nullptr;
    double *d_stress_slot = nullptr;     // one per column slot: the stress of a strip a wave left
    std::vector<int32_t> slot_strip;     // strip (block column) of every column slot
    std::vector<int32_t> wave_last_strip;  // strip each wave ends in, -1: no units
    // several maps in one solver (bb_solver_set_maps)
    int n_maps = 1;
    void *d_bin_scale = nullptr;         // T per bin (n_pad): the factor on its gradient
    int *d_map_ptr = nullptr, *d_map_idx = nullptr;
    double *d_map_scalar = nullptr;      // per-map stress of bb_solver_stress_maps
    std::vector<int64_t> map_begin;      // first bin of every map, + n_bins
    int64_t *d_blk_ptr = nullptr, *d_blk_chunk = nullptr;    // final stage: one list per block
    int64_t *d_s1_ptr = nullptr, *d_s1_chunk = nullptr;      // stage 1: slices of long lists
    int64_t n_slices = 0, part2_off = 0;
    int64_t *d_red_lists = nullptr;   // reduce_sliced_kernel: one padded chunk list per block
    int red_stride = 0, red_slices = 0;   // entries per block (multiple of 16 * slices); 4 or 8 slices, 0 = old reduce
    double *d_stress_hist = nullptr, *d_stress_scalar = nullptr;
    double *d_f64_tmp = nullptr;  // (n_pad,3) staging for coordinate I/O
    void *d_mv_in = nullptr;      // (n_pad,3) right-hand sides of bb_solver_matvec_sq, kept
    int64_t rowpart_elems = 0, colpart_elems = 0;
    int n_waves = 0, n_slots = 0;
    int wpb = 4;                   // waves per workgroup of the sweep: 4, or 8 (paired, see kernel)
    int wg_map = 0;                // block index -> run of chunks (stress_grad_kernel): 0 identity
    int dense_u0 = -1;             // dense tile list: this rank's first global unit (the kernel then
                                   // computes a wave's first descriptor), else -1
    int defer_cap_units = 0;       // fp32: units of row sums a wave can park in LDS (0 = none)
    int64_t defer_lds_bytes = 0;   // dynamic LDS per workgroup for that, 0 = per-unit stores
    unsigned defer_attr_done = 0;  // kernel variants whose dynamic-LDS ceiling was raised
    int64_t hist_cap = 0, hist_n = 0;
    bool have_wish = false, have_coords = false, grad_pending = false;
    // exchange_sum(): while set, the reduce / exchange kernels leave the plain sum over the
    // ranks HERE instead of stepping the coordinates (X := this buffer, zeroed; V := d_V;
    // mu = 0, lr = -1: 0 + (0 * v - (-1) * sum) is the sum, exactly)
    void *sum_target = nullptr;

    bb::Rccl::Comm comm = nullptr;  // direct RCCL path (bb_solver_comm_init), else null
    bool comm_cached = false;       // the communicator belongs to the process-wide cache
    bool comm_suspect = false;      // a collective on it failed to enqueue: never handed out again

    // peer exchange (bb_solver_peer_*)
    void *peer_arena = nullptr;             // this rank's receive arena (uncached)
    int64_t peer_slot_elems = 0, peer_arena_bytes = 0;
    std::vector<void *> peer_mapped;        // arenas of all ranks as mapped here (own = peer_arena)
    std::vector<void *> peer_opened;        // the ones that came from hipIpcOpenMemHandle
    void *d_peer_table = nullptr;           // PeerTable<T>[2], one per parity
    void *d_peer_table_x = nullptr;         // PeerTableX<T>[2]: the one-launch exchange
    bool peer_fused = true;                 // reduce_exchange_kernel (BB_PEER_FUSED=0: two launches)
    PeerState *d_peer_state = nullptr;      // sticky failure flag + last complete exchange
    int peer_clock_khz = 100000;            // constant-rate clock behind wall_clock64()
    int peer_ranks_on_gpu = 1;              // most ranks of the job on one GPU (a rehearsal: > 1)
    unsigned *d_peer_mask = nullptr;        // per block: the ranks whose units touch it (bit r)
    unsigned *d_peer_counter = nullptr;
    unsigned long long peer_seq = 0;        // iterations exchanged so far
    long long peer_limit_ticks = 0;
    bool peer_connected = false;

    // row-owner path (small maps, world = 1): both triangles resident, one launch per
    // iteration (row_owner_kernel); the units stay resident too and serve every other
    // entry point (grad / apply, stress, matvec)
    bool row_owner = false;
    bool row_owner_built = false;        // created with the row-owner buffers (bb_solver_set_maps
                                         // switches row_owner off)
    bool bin_steps = false;              // d_bin_scale set by bb_solver_set_bin_steps / _block_steps
    void *d_full = nullptr, *d_X2 = nullptr;
    int64_t full_ld = 0;
    double *d_ro_part = nullptr;   // 2 x ro_blocks per-workgroup stress sums (ping-pong)
    int ro_blocks = 0, ro_wpr = 1;  // workgroups holding rows; waves per row (1, 2, 4)
    int64_t ro_launches = 0;       // row-owner launches so far (parity of the ping-pong)

    bool timing = false;
    int timing_stride = 1;      // events on every timing_stride-th iteration
    int64_t timing_iter = 0;    // iterations seen since timing was (re-)enabled
    std::vector<hipEvent_t> ev;  // triples: start, after grad, after reduce
    size_t ev_used = 0;
};

namespace {

void comm_release(bb_solver *s, bool destroy);   // the communicator cache, further down
int64_t peer_xpoison_offset(const bb_solver *s);  // the peer arena's layout, further down

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

// Waves per CU (4 waves = one workgroup).  Measured on MI355X: with the rolling
// 8-KiB window one wave per SIMD already keeps enough bytes in flight, and fewer,
// longer chunks mean fewer column-partial slots and less prologue per byte; two
// waves per SIMD overlap one wave's VALU work with the other's waits.  With the
// current kernel (profiles/archive/r01_sweep_wpc.txt): 8/CU is 2-4 % faster from ~150k units
// per rank (N=24,926) up, 4 and 8 tie at 77k units (1/8 of the N=50k matrix, where
// 4 means half as many column partials for the reduce), 6 is always worse (a
// workgroup count that is not a multiple of the CU count), 16 is 2 % slower.
// Round 3, with the one-launch reduce (whose time no longer grows with the number of
// column slots) and by the kernel trace (tools/wpc_timeline.sh, profiles/r03_wpc_ab.txt): 8
// per CU takes 110.3-111.9 us per iteration at N=17,700 against 116.3-116.5 with 4, and ties
// at N=12,000 and 14,500: the switch is at 65k units (N ~ 16,200 on one rank).
// BB_WAVES_PER_CU overrides.
int waves_per_cu(int64_t n_local) {
    const char *e = getenv("BB_WAVES_PER_CU");
    int v = e ? atoi(e) : (n_local >= 65000 ? 8 : 4);
    if (v < 1) v = 1;
    if (v > 32) v = 32;
    return v;
}

// Largest map (bins) that iterates on the row-owner path.  Measured on MI355X
// (tools/small_n_timing.py, profiles/archive/r02_small_n.txt); BB_ROW_OWNER_MAX overrides,
// 0 turns the path off.
int64_t row_owner_max() {
    const char *e = getenv("BB_ROW_OWNER_MAX");
    return e ? atoll(e) : bb::kRowOwnerMaxBins;
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
    {
        // 8 waves per CU: put the two waves of a SIMD into one workgroup so that they can
        // keep each other's pace (stress_grad_kernel, WPB = 8).  BB_PAIR=0 turns it off.
        const char *e = getenv("BB_PAIR");
        const bool pair = !(e && atoi(e) == 0);
        s->wpb = (pair && (s->dtype == BB_F32 || s->wide) && waves_per_cu(s->n_local) >= 8 && nw >= 8)
                     ? 8 : 4;
    }
    nw = bb::round_up(nw, s->wpb);
    s->n_waves = (int)nw;
    {
        // BB_WG_MAP: 0 identity, m > 0 multiplicative permutation (made coprime with the
        // workgroup count), -1 one contiguous eighth of the chunks per XCD
        const char *e = getenv("BB_WG_MAP");
        const int64_t nwg = nw / s->wpb;
        int m = e ? atoi(e) : 0;
        if (m < 0 && nwg % 8 != 0) m = 0;
        if (m > 0) {
            auto gcd = [](int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; };
            while (gcd(m, nwg) != 1) ++m;
            m = (int)(m % nwg);
            if (nwg * (int64_t)m >= ((int64_t)1 << 32)) m = 0;
        }
        s->wg_map = m;
    }

    if (s->n_local >= ((int64_t)1 << 31))
        return bb::fail(BB_ERR_INVALID, "bb_solver_create: more than 2^31 units on one rank");
    {
        // Matrix loads are non-temporal (measured: +2.5 % on the kernel, +9 % on a pure read
        // sweep at N=50k) unless this rank's units fit the 256-MB Infinity Cache: then they
        // are read again from it on the next iteration, and plain loads keep them there
        // (N=8,000 / 10,000: kernel -8 %, step -3.5 / -6 %; equal at N=12,000 = 288 MB;
        // non-temporal 3-4 % better at N=17,700; profiles/archive/r02_nt_ab.txt).  BB_NT=0|1 overrides.
        const char *e = getenv("BB_NT");
        s->nontemporal = e ? atoi(e) != 0 : s->n_local * bb::kUnitBytes > ((int64_t)240 << 20);
    }
    // wave w owns the contiguous chunk [w*q + min(w, r), +q (+1 if w < r)), q = n_local / nw,
    // r = n_local % nw: the kernels compute it the same way (no table to load)
    s->chunk_q = (int)(s->n_local / nw);
    s->chunk_r = (int)(s->n_local % nw);
    std::vector<int2> wave_range(nw);
    int64_t chunk_max = 0;
    for (int64_t w = 0; w < nw; ++w) {
        const int64_t a = w * s->chunk_q + std::min<int64_t>(w, s->chunk_r);
        wave_range[w] = make_int2((int)a, (int)(a + s->chunk_q + (w < s->chunk_r ? 1 : 0)));
        chunk_max = std::max<int64_t>(chunk_max, wave_range[w].y - wave_range[w].x);
    }
    {
        // fp32: a wave parks the row sums of (the last cap units of) its chunk in LDS
        // until it has finished reading (stress_grad_kernel, DEFER); cap is what the
        // workgroups sharing a CU can hold.  BB_DEFER_ROWS=0 turns it off.
        const char *e = getenv("BB_DEFER_ROWS");
        const int64_t wgs_per_cu = std::max<int64_t>(1, (nw / s->wpb + cus - 1) / cus);
        const int64_t budget = 156 * 1024 / wgs_per_cu;               // of the CU's 160 KiB
        const int64_t cap = std::min<int64_t>(chunk_max, (budget / s->wpb - 16 - 8) / 48);
        const bool on = !(e && atoi(e) == 0);
        if ((s->dtype == BB_F32 || s->wide) && on && cap > 0) s->defer_cap_units = (int)cap;
        // a wave's LDS region: the parked row sums while it sweeps, its last column partial
        // (3 * vw elements) at the end; + the 8 progress words of a paired workgroup
        const int64_t col_words = 3 * vw * bb::elem_size(s->dtype) / 4;
        s->lds_wave_floats = (int)std::max<int64_t>((int64_t)s->defer_cap_units * 12 + 4, col_words);
        s->defer_lds_bytes = (int64_t)s->wpb * s->lds_wave_floats * 4 + 32 +
                             (abl::kUnitTrace ? (int64_t)s->wpb * abl::kUnitTraceSlots * 8 : 0);
    }
    // Column-partial slots.  A wave's strips but the last get private slots (it writes them
    // itself, mid-sweep: rare); the strip a wave ENDS in shares one slot with the other
    // waves of its workgroup that end in the same strip (summed in LDS by the sweep's
    // epilogue).  BB_WG_COLSUM=0 gives every wave its own last slot too (for A/B).
    std::vector<int32_t> &slot_strip = s->slot_strip;
    slot_strip.clear();
    s->wave_last_strip.assign((size_t)nw, -1);
    std::vector<int2> wave_slots(nw);
    {
        const char *e = getenv("BB_WG_COLSUM");
        const bool share = !(e && atoi(e) == 0);
        for (int64_t w = 0; w < nw; ++w) {
            wave_slots[w] = make_int2((int)slot_strip.size(), -1);
            if (wave_range[w].x >= wave_range[w].y) continue;           // no units
            int cur = -1;
            std::vector<int> strips;
            for (int64_t ul = wave_range[w].x; ul < wave_range[w].y; ++ul) {
                const int J = s->udesc[ul].y / (int)vw;
                if (J != cur) { strips.push_back(J); cur = J; }
            }
            for (size_t q = 0; q + 1 < strips.size(); ++q) slot_strip.push_back(strips[q]);
            const int Jlast = strips.back();
            s->wave_last_strip[(size_t)w] = Jlast;
            const bool same_wg = w % s->wpb != 0;
            if (share && same_wg && wave_slots[w - 1].y >= 0 &&
                slot_strip[(size_t)wave_slots[w - 1].y] == Jlast) {
                wave_slots[w].y = wave_slots[w - 1].y;
            } else {
                wave_slots[w].y = (int)slot_strip.size();
                slot_strip.push_back(Jlast);
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

    // Reduce: a block's list can hold hundreds of chunks (one per strip of its tile row, one
    // per wave that crossed its strip).  One element per thread and kReduceSlice loads in
    // flight make a list of L chunks cost about L / kReduceSlice memory round trips, so while
    // the longest list has at most 128 chunks (N <= ~35k on one rank, every 1/8 share of
    // N=50k) ONE launch sums every list whole; beyond that lists are cut into slices that a
    // first launch sums in parallel into `part2`, which the final stage then combines.
    // Measured (profiles/archive/r02_reduce_ab.txt): one launch saves 1-1.6 us per iteration at
    // N=17,700 / 24,926, nothing at 50k, and loses 1-2 us at 61,914 -- the reduce is bound by
    // the 20-56 MB of partials the sweep has just written, not by its launches.
    // BB_REDUCE_SINGLE_MAX overrides.
    constexpr int64_t kSlice = kReduceSlice;
    int64_t single_max = 128;
    if (const char *e = getenv("BB_REDUCE_SINGLE_MAX")) single_max = atoll(e);
    int64_t longest = 0;
    for (int64_t b = 0; b < nb; ++b) longest = std::max(longest, blk_ptr[b + 1] - blk_ptr[b]);
    const int64_t slice_from = longest <= single_max ? longest : kSlice;   // lists longer than this are sliced
    std::vector<int64_t> s1_ptr(1, 0), s1_chunk, fin_ptr(nb + 1, 0), fin_chunk;
    s->part2_off = s->rowpart_elems + s->colpart_elems;
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t k0 = blk_ptr[b], k1 = blk_ptr[b + 1];
        if (k1 - k0 <= slice_from) {
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
    // reduce_sliced_kernel (the default; BB_REDUCE_OLD=1 keeps the two-stage reduce above):
    // every block's whole list in one table of fixed stride, padded with the offset of a
    // chunk of zeros that sits behind the stage-1 partials
    const int64_t zero_off = s->part2_off + s->n_slices * ch;
    const int64_t part_total = zero_off + ch;
    std::vector<int64_t> red_lists;
    {
        const char *e = getenv("BB_REDUCE_OLD");
        s->red_slices = (e && atoi(e) != 0) ? 0 : (longest <= 64 ? 4 : 8);
        if (const char *f = getenv("BB_REDUCE_SLICES")) s->red_slices = atoi(f) == 4 ? 4 : 8;
        if (s->red_slices > 0) {
            const int64_t quantum = 16 * s->red_slices;
            s->red_stride = (int)bb::round_up(std::max<int64_t>(longest, 1), quantum);
            red_lists.assign((size_t)(nb * s->red_stride), zero_off);
            for (int64_t b = 0; b < nb; ++b)
                for (int64_t k = blk_ptr[b]; k < blk_ptr[b + 1]; ++k)
                    red_lists[(size_t)(b * s->red_stride + (k - blk_ptr[b]))] = blk_chunk[k];
        }
    }
    if (red_lists.empty()) red_lists.push_back(0);

    const int64_t es = bb::elem_size(s->dtype);
    s->hist_cap = kHistCap;
    if (s->row_owner) {
        // a whole number of trips per row: 256 columns in fp32, 128 in fp64 (RowTrip<T>::COLS)
        s->full_ld = bb::round_up(s->L.n_bins, s->dtype == BB_F32 ? RowTrip<float>::COLS
                                                                  : RowTrip<double>::COLS);
        // waves per row: enough of them that a small map still covers the chip with
        // about 16 waves per CU (N=963: 4 per row); BB_ROW_OWNER_WPR overrides
        const char *ew = getenv("BB_ROW_OWNER_WPR");
        int wpr = ew ? atoi(ew) : (s->L.n_bins <= 1024 ? 4 : (s->L.n_bins <= 2048 ? 2 : 1));
        s->ro_wpr = wpr >= 4 ? 4 : (wpr >= 2 ? 2 : 1);
        const int rows_per_wg = 4 / s->ro_wpr;
        s->ro_blocks = (int)((s->L.n_bins + rows_per_wg - 1) / rows_per_wg);
    }
    {
        // ONE device allocation for everything the solver owns (the exchange buffer apart:
        // a caller may replace it).  Twenty hipMalloc + hipFree pairs were 3 of the 4.5 ms a
        // whole chr21-sized fit took (tools/small_fit_timing.py).
        std::vector<std::pair<void **, size_t>> want;
        auto add = [&](auto **ptr, int64_t count) {
            want.push_back({(void **)ptr, (size_t)std::max<int64_t>(count, 1) * sizeof(**ptr)});
        };
        add((char **)&s->d_units, std::max<int64_t>(s->n_local, 1) * bb::kUnitBytes);
        add((char **)&s->d_X, s->L.n_pad * 3 * es);
        add((char **)&s->d_V, s->L.n_pad * 3 * es);
        add((char **)&s->d_part, part_total * es);
        add(&s->d_udesc, (int64_t)s->udesc.size());
        add(&s->d_wave_slots, nw);
        // nw partials (+ 8 stamps per wave, + one stamp per unit: diagnostic builds)
        add(&s->d_stresspart, nw * (9 + (abl::kUnitTrace ? abl::kUnitTraceSlots : 0)));
        add(&s->d_stress_slot, std::max<int64_t>(s->n_slots, 1));
        add(&s->d_blk_ptr, nb + 1);
        add(&s->d_blk_chunk, (int64_t)fin_chunk.size());
        add(&s->d_s1_ptr, (int64_t)s1_ptr.size());
        add(&s->d_s1_chunk, (int64_t)s1_chunk.size());
        add(&s->d_red_lists, (int64_t)red_lists.size());
        add(&s->d_stress_hist, kHistCap);
        add(&s->d_stress_scalar, 1);
        add(&s->d_f64_tmp, s->L.n_pad * 3);
        if (s->row_owner) {
            add((char **)&s->d_full, s->L.n_bins * s->full_ld * es);
            add((char **)&s->d_X2, s->L.n_pad * 3 * es);
            add(&s->d_ro_part, 2 * (int64_t)s->ro_blocks);
        }
        size_t total = 0;
        for (auto &w : want) total += (w.second + 255) & ~(size_t)255;
        BB_TRY(dev_alloc(&s->d_arena, (int64_t)total));
        size_t off = 0;
        for (auto &w : want) {
            *w.first = s->d_arena + off;
            off += (w.second + 255) & ~(size_t)255;
        }
    }
    BB_TRY(dev_alloc((char **)&s->d_exch, (3 * s->L.n_pad + 2) * es));
    s->own_exch = true;

    hipStream_t st = s->stream;
    BB_HIP_CHECK(hipMemcpyAsync(s->d_udesc, s->udesc.data(), s->udesc.size() * sizeof(int2),
                                hipMemcpyHostToDevice, st));
    BB_HIP_CHECK(hipMemcpyAsync(s->d_wave_slots, wave_slots.data(), nw * sizeof(int2),
                                hipMemcpyHostToDevice, st));
    BB_HIP_CHECK(hipMemcpyAsync(s->d_blk_ptr, fin_ptr.data(), (nb + 1) * sizeof(int64_t),
                                hipMemcpyHostToDevice, st));
    BB_HIP_CHECK(hipMemcpyAsync(s->d_blk_chunk, fin_chunk.data(),
                                fin_chunk.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    BB_HIP_CHECK(hipMemcpyAsync(s->d_s1_ptr, s1_ptr.data(), s1_ptr.size() * sizeof(int64_t),
                                hipMemcpyHostToDevice, st));
    BB_HIP_CHECK(hipMemcpyAsync(s->d_s1_chunk, s1_chunk.data(),
                                s1_chunk.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    BB_HIP_CHECK(hipMemcpyAsync(s->d_red_lists, red_lists.data(),
                                red_lists.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    // rows of boundary tiles owned by another rank are never written: keep them 0
    BB_HIP_CHECK(hipMemsetAsync(s->d_part, 0, (size_t)part_total * es, st));
    BB_HIP_CHECK(hipMemsetAsync(s->d_X, 0, (size_t)(s->L.n_pad * 3 * es), st));
    BB_HIP_CHECK(hipMemsetAsync(s->d_V, 0, (size_t)(s->L.n_pad * 3 * es), st));
    if (s->row_owner)
        BB_HIP_CHECK(hipMemsetAsync(s->d_X2, 0, (size_t)(s->L.n_pad * 3 * es), st));
    BB_HIP_CHECK(hipMemsetAsync(s->d_exch, 0, (size_t)((3 * s->L.n_pad + 2) * es), st));
    BB_HIP_CHECK(hipMemsetAsync(s->d_stresspart, 0, (size_t)nw * sizeof(double), st));
    // (shared slots -- the strip a wave ENDS in -- carry no stress of their own: they stay 0)
    BB_HIP_CHECK(hipMemsetAsync(s->d_stress_slot, 0,
                                (size_t)std::max<int64_t>(s->n_slots, 1) * sizeof(double), st));
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
    const dim3 grid(s->n_waves / s->wpb), block(64 * s->wpb);
    // fp32 with the whole chunk of row sums parked in LDS (s->defer_lds_bytes > 0), or the
    // per-unit store.  The dynamic-LDS ceiling of a kernel is raised once per instantiation.
#define BB_LAUNCH3(NTV, OPV, DEF, LDS, WPBV)                                                    \
    do {                                                                                        \
        auto kern = stress_grad_kernel<T, W, NTV, OPV, DEF, WPBV>;                              \
        constexpr unsigned bit =                                                                \
            1u << ((WPBV == 8 ? 4 : 0) + (NTV ? 2 : 0) + (OPV == kOpMatvec2 ? 1 : 0));          \
        if ((LDS) > 0 && !(s->defer_attr_done & bit)) {                                        \
            BB_HIP_CHECK(hipFuncSetAttribute((const void *)kern,                                \
                                             hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                             (int)(LDS)));                                      \
            s->defer_attr_done |= bit;                                                          \
        }                                                                                       \
        BB_HIP_CHECK(bb::launch(kern, grid, block, (size_t)(LDS), s->stream, units, X,          \
                                s->d_udesc, s->chunk_q, s->chunk_r, s->d_wave_slots, rowpart,   \
                                colpart, s->d_stresspart, s->defer_cap_units,                   \
                                s->lds_wave_floats, s->wg_map, s->dense_u0, s->d_stress_slot)); \
    } while (0)
#define BB_LAUNCH2(NTV, OPV, DEF, LDS)                                                          \
    do {                                                                                        \
        if ((sizeof(T) == 4 || W) && s->wpb == 8)                                               \
            BB_LAUNCH3(NTV, OPV, DEF, LDS, ((sizeof(T) == 4 || W) ? 8 : 4));                    \
        else                                                                                    \
            BB_LAUNCH3(NTV, OPV, DEF, LDS, 4);                                                  \
    } while (0)
#define BB_LAUNCH(NTV, OPV)                                                                     \
    do {                                                                                        \
        if ((sizeof(T) == 4 || W) && s->defer_cap_units > 0)                                    \
            BB_LAUNCH2(NTV, OPV, (sizeof(T) == 4 || W), s->defer_lds_bytes);                    \
        else                                                                                    \
            BB_LAUNCH2(NTV, OPV, false, s->defer_lds_bytes);                                    \
    } while (0)
    if (op == kOpMatvec2) {
        if (s->nontemporal) BB_LAUNCH(true, kOpMatvec2); else BB_LAUNCH(false, kOpMatvec2);
    } else {
        if (s->nontemporal) BB_LAUNCH(true, kOpStress); else BB_LAUNCH(false, kOpStress);
    }
#undef BB_LAUNCH
#undef BB_LAUNCH2
#undef BB_LAUNCH3
    return BB_OK;
}

// What the reduce multiplies the summed partials by: 2 for the gradient (SPEC 2.3: the sweep
// leaves half of it), 1 for a plain sum (the matvec of the spectral start) -- and only a
// gradient takes the per-bin step factors (fill_reduce_params).
constexpr double kScaleGradient = 2.0, kScalePlainSum = 1.0;

template <typename T>
void fill_reduce_params(bb_solver *s, ReduceParams<T> &p, int mode, double lr, double *stress_out,
                        double scale);

// The peer exchange in one launch (reduce_exchange_kernel): the caller has bumped peer_seq.
template <typename T, bool W>
int launch_exchange_t(bb_solver *s, double lr, double *stress_out, double scale) {
    ReduceParams<T> p;
    fill_reduce_params<T>(s, p, kReducePeer, lr, stress_out, scale);
    const int64_t es = sizeof(T);
    const unsigned segs = 3 * Lay<T, W>::VW / kRedWG;
    const dim3 grid((unsigned)s->L.n_blocks, segs);
    const PeerTableX<T> *xt = (const PeerTableX<T> *)s->d_peer_table_x + (s->peer_seq & 1);
    const char *base = (const char *)s->peer_arena;
    const T *arena = (const T *)(base + (int64_t)(s->peer_seq & 1) * s->world * s->peer_slot_elems * es);
    const unsigned long long *my_poison = (const unsigned long long *)(base + peer_xpoison_offset(s));
    if (s->red_slices == 4)
        BB_HIP_CHECK(bb::launch(reduce_exchange_kernel<T, W, 4>, grid, dim3(128 * 4), 0, s->stream, p,
                                (const int64_t *)s->d_red_lists, s->red_stride, xt, arena, my_poison,
                                s->peer_slot_elems, s->d_peer_state, s->peer_limit_ticks));
    else
        BB_HIP_CHECK(bb::launch(reduce_exchange_kernel<T, W, 8>, grid, dim3(128 * 8), 0, s->stream, p,
                                (const int64_t *)s->d_red_lists, s->red_stride, xt, arena, my_poison,
                                s->peer_slot_elems, s->d_peer_state, s->peer_limit_ticks));
    return BB_OK;
}

template <typename T>
void fill_reduce_params(bb_solver *s, ReduceParams<T> &p, int mode, double lr, double *stress_out,
                        double scale) {
    p.part = (const T *)s->d_part;
    p.stresspart = s->d_stresspart;
    p.X = s->sum_target ? (T *)s->sum_target : (T *)s->d_X;
    p.V = (T *)s->d_V;
    p.mu = s->sum_target ? T(0) : (T)s->momentum;
    p.scale = (T)scale;
    p.exch = (T *)s->d_exch;
    p.part_out = (T *)s->d_part + s->part2_off;
    p.stress_out = stress_out;
    p.n_pad = s->L.n_pad;
    p.n_waves = s->n_waves;
    p.lr = s->sum_target ? T(-1) : (T)lr;
    p.peer = nullptr;
    p.peer_counter = s->d_peer_counter;
    p.peer_state = s->d_peer_state;
    p.seq = 0;
    p.n_peers = 0;
    if (mode == kReducePeer) {
        // the caller has bumped peer_seq: iteration k (1-based) uses parity k & 1
        p.peer = (const PeerTable<T> *)s->d_peer_table + (s->peer_seq & 1);
        p.seq = s->peer_seq;
        p.n_peers = s->world;
    }
    p.blk_ptr = nullptr;
    p.blk_chunk = nullptr;
    p.mode = mode;
    p.stress_slot = s->d_stress_slot;
    p.n_slots = s->n_slots;
    // (a plain sum -- the matvec, the spectral start's products -- is never scaled)
    p.bin_scale = scale == kScalePlainSum ? nullptr : (const T *)s->d_bin_scale;
    p.map_ptr = s->d_map_ptr;
    p.map_idx = s->d_map_idx;
    p.n_maps = s->n_maps;
    p.peer_mask = s->d_peer_mask;
    p.rank = s->rank;
}

template <typename T, bool W>
int launch_reduce_t(bb_solver *s, int mode, double lr, double *stress_out, double scale) {
    ReduceParams<T> p;
    fill_reduce_params<T>(s, p, mode, lr, stress_out, scale);
    const unsigned segs = 3 * Lay<T, W>::VW / kRedWG;   // workgroups per block of 3*vw elements
    if (s->red_slices > 0) {
        // one launch: every list whole, 128 elements x 4 or 8 slices per workgroup
        const dim3 grid = mode == kReduceStressOnly ? dim3((unsigned)s->n_maps, 1)
                                                    : dim3((unsigned)s->L.n_blocks, segs);
        if (s->red_slices == 4)
            BB_HIP_CHECK(bb::launch(reduce_sliced_kernel<T, W, 4>, grid, dim3(128 * 4), 0, s->stream,
                                    p, (const int64_t *)s->d_red_lists, s->red_stride));
        else
            BB_HIP_CHECK(bb::launch(reduce_sliced_kernel<T, W, 8>, grid, dim3(128 * 8), 0, s->stream,
                                    p, (const int64_t *)s->d_red_lists, s->red_stride));
        return BB_OK;
    }
    if (mode != kReduceStressOnly && s->n_slices > 0) {
        p.blk_ptr = s->d_s1_ptr;
        p.blk_chunk = s->d_s1_chunk;
        p.mode = kReducePartial;
        BB_HIP_CHECK(bb::launch(reduce_kernel<T, W>, dim3((unsigned)s->n_slices, segs), dim3(kRedWG),
                                0, s->stream, p));
    }
    p.blk_ptr = s->d_blk_ptr;
    p.blk_chunk = s->d_blk_chunk;
    p.mode = mode;
    const dim3 grid = mode == kReduceStressOnly ? dim3((unsigned)s->n_maps, 1)
                                                : dim3((unsigned)s->L.n_blocks, segs);
    BB_HIP_CHECK(bb::launch(reduce_kernel<T, W>, grid, dim3(kRedWG), 0, s->stream, p));
    return BB_OK;
}

int launch_grad(bb_solver *s, int op = kOpStress, const void *x_in = nullptr) {
    if (!x_in) x_in = s->d_X;
    return BB_BY_LAYOUT(s, launch_grad_t, s, op, x_in);
}
int launch_reduce(bb_solver *s, int mode, double lr, double *stress_out,
                  double scale = kScaleGradient) {
    return BB_BY_LAYOUT(s, launch_reduce_t, s, mode, lr, stress_out, scale);
}
int launch_exchange(bb_solver *s, double lr, double *stress_out, double scale = kScaleGradient) {
    return BB_BY_LAYOUT(s, launch_exchange_t, s, lr, stress_out, scale);
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

// f64 <-> solver dtype conversions of n elements on the solver's stream.
hipError_t widen(bb_solver *s, const void *in_T, double *out_f64, int64_t n) {
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    return s->dtype == BB_F32
               ? bb::launch(T_to_f64_kernel<float>, grid, block, 0, s->stream, (const float *)in_T,
                            out_f64, n)
               : bb::launch(T_to_f64_kernel<double>, grid, block, 0, s->stream,
                            (const double *)in_T, out_f64, n);
}
hipError_t narrow(bb_solver *s, const double *in_f64, void *out_T, int64_t n) {
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    return s->dtype == BB_F32
               ? bb::launch(f64_to_T_kernel<float>, grid, block, 0, s->stream, in_f64,
                            (float *)out_T, n)
               : bb::launch(f64_to_T_kernel<double>, grid, block, 0, s->stream, in_f64,
                            (double *)out_T, n);
}

// host = the (n_sub, n_sub) matrix of the bins [off, off + n_sub) of the solver (the whole
// map: off = 0, n_sub = n_bins); only units inside that block are touched
template <typename T, bool W>
int set_wish_dense_t(bb_solver *s, const double *host, int64_t ld, int kind, double alpha,
                     int64_t off, int64_t n_sub) {
    constexpr int VW = Lay<T, W>::VW;
    const int64_t upt = s->L.units_per_tile, n = off + n_sub;
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
        if (j0 < off || j0 >= n || i_start < off) { ++ul; continue; }      // another map's unit
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
            e = hipMemcpy2DAsync(stage, VW * sizeof(double), host + (i_start - off) * ld + (j0 - off),
                                 (size_t)ld * sizeof(double), (size_t)cols_valid * sizeof(double),
                                 (size_t)rows_valid, hipMemcpyHostToDevice, s->stream);
        if (e == hipSuccess) {
            e = bb::launch(convert_units_kernel<T, W>, dim3((unsigned)(ue - ul)), dim3(256), 0,
                           s->stream, stage, (T *)s->d_units, s->d_udesc, ul, i_start, n, kind,
                           -1.0 / alpha);
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

// n_pad factors (float64 on the host) -> d_bin_scale in the solver's type.
int upload_bin_scale(bb_solver *s, const double *padded) {
    const int64_t n = s->L.n_pad, es = bb::elem_size(s->dtype);
    std::vector<char> host((size_t)(n * es));
    for (int64_t i = 0; i < n; ++i) {
        if (s->dtype == BB_F32) ((float *)host.data())[i] = (float)padded[i];
        else ((double *)host.data())[i] = padded[i];
    }
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    if (!s->d_bin_scale) BB_TRY(dev_alloc((char **)&s->d_bin_scale, n * es));
    BB_HIP_CHECK(hipMemcpy(s->d_bin_scale, host.data(), host.size(), hipMemcpyHostToDevice));
    return BB_OK;
}

// Row-owner path: rebuild both triangles from the freshly packed units.
int refresh_full(bb_solver *s) {
    if (!s->row_owner) return BB_OK;
    const int64_t es = bb::elem_size(s->dtype);
    BB_HIP_CHECK(hipMemsetAsync(s->d_full, 0, (size_t)(s->L.n_bins * s->full_ld * es), s->stream));
    if (s->n_local > 0) {
#define BB_FULL(TT, WW)                                                                         \
    BB_HIP_CHECK(bb::launch(units_to_full_kernel<TT, WW>, dim3((unsigned)s->n_local), dim3(256), 0, \
                            s->stream, (const TT *)s->d_units, s->d_udesc, (TT *)s->d_full,       \
                            s->full_ld, s->L.n_bins))
        if (s->dtype == BB_F32) BB_FULL(float, true);
        else if (s->wide) BB_FULL(double, true);
        else BB_FULL(double, false);
#undef BB_FULL
    }
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    return BB_OK;
}

// One row-owner launch: iteration `hist_n` (reads d_X, writes d_X2, swaps them) and the
// fold of the previous launch's stress into hist[hist_n - 1] when `fold_prev`.
// update = false: fold only (after the last iteration of a call).
template <typename T>
int launch_row_owner_t(bb_solver *s, double lr, bool fold_prev, bool update) {
    const int par = (int)(s->ro_launches & 1);
    const double *prev = s->d_ro_part + (int64_t)(par ^ 1) * s->ro_blocks;
    double *out = s->d_ro_part + (int64_t)par * s->ro_blocks;
    double *hist_prev = fold_prev ? s->d_stress_hist + (s->hist_n - 1) : nullptr;
    // fold only: a grid of one workgroup, which is then the "last" = the fold workgroup
    const unsigned grid = update ? (unsigned)s->ro_blocks + 1u : 1u;
#define BB_ROW(WPRV)                                                                             \
    BB_HIP_CHECK(bb::launch(row_owner_kernel<T, WPRV>, dim3(grid), dim3(256), 0, s->stream,          \
                            (const T *)s->d_full, s->full_ld, (int)s->L.n_bins, (const T *)s->d_X,  \
                            (T *)s->d_X2, (T *)s->d_V, (T)lr, (T)s->momentum, prev, s->ro_blocks,   \
                            hist_prev, out, (const T *)s->d_bin_scale))
    if (s->ro_wpr == 4) BB_ROW(4);
    else if (s->ro_wpr == 2) BB_ROW(2);
    else BB_ROW(1);
#undef BB_ROW
    if (update) {
        std::swap(s->d_X, s->d_X2);
        s->ro_launches++;
    }
    return BB_OK;
}
int launch_row_owner(bb_solver *s, double lr, bool fold_prev, bool update) {
    return s->dtype == BB_F32 ? launch_row_owner_t<float>(s, lr, fold_prev, update)
                              : launch_row_owner_t<double>(s, lr, fold_prev, update);
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
    uint64_t pci;            // PCI domain << 32 | bus << 16 | device: which GPU, whatever its index
};
static_assert(sizeof(PeerHandle) <= BB_PEER_HANDLE_BYTES, "handle blob too small");
static_assert(offsetof(PeerTable<float>, flag) == kMaxPeers * sizeof(void *) &&
                  offsetof(PeerTable<double>, flag) == kMaxPeers * sizeof(void *),
              "peer_receive_kernel takes the flag pointers as the second half of a PeerTable");
constexpr uint64_t kPeerMagic = 0x6262706565723031ull;  // "bbpeer01"

int64_t peer_flags_offset(const bb_solver *s) {
    return 2 * (int64_t)s->world * s->peer_slot_elems * bb::elem_size(s->dtype);
}
// the one-launch exchange (reduce_exchange_kernel): one poison word per source rank behind
// the per-rank flags of the two-launch form (its data words say by themselves whether they
// have arrived)
int64_t peer_xpoison_offset(const bb_solver *s) {
    return peer_flags_offset(s) + bb::round_up((int64_t)s->world * 64, 256);
}
int64_t peer_arena_size(const bb_solver *s) {
    return peer_xpoison_offset(s) + bb::round_up((int64_t)s->world * 64, 256);
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
    PeerTableX<T> tx[2];
    memset(tx, 0, sizeof(tx));
    for (int par = 0; par < 2; ++par)
        for (int q = 0; q < s->world; ++q) {
            char *base = (char *)s->peer_mapped[q];
            tx[par].dst[q] = tab[par].dst[q];
            tx[par].poison[q] = (unsigned long long *)(base + peer_xpoison_offset(s)) + 8 * s->rank;
        }
    BB_TRY(dev_alloc((char **)&s->d_peer_table_x, (int64_t)sizeof(tx)));
    BB_HIP_CHECK(hipMemcpy(s->d_peer_table_x, tx, sizeof(tx), hipMemcpyHostToDevice));
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
    s->row_owner = s->row_owner_built = world == 1 && n_bins <= row_owner_max();
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
    if (rc == BB_OK && tile_I == nullptr && s->u_end < ((int64_t)1 << 31)) {
        const char *e = getenv("BB_ARITH_DESC");
        if (!(e && atoi(e) == 0)) s->dense_u0 = (int)s->u_begin;
    }
    if (rc == BB_OK) {
        hipError_t e = bb::acquire_stream(device, &s->stream);
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
    if (!s->stream_stuck && (s->stream || !s->own_stream)) hipStreamSynchronize(s->stream);
    if (s->comm) {
        bb::Rccl::Comm c = s->comm;
        const bool cached = s->comm_cached, suspect = s->comm_suspect;
        comm_release(s, /*destroy=*/false);  // a cached communicator goes back to the cache ...
        if (suspect) {                       // ... unless a collective on it failed: out of the
            if (bb::rccl().CommAbort) bb::rccl().CommAbort(c);   // cache (comm_release) and ended
        } else if (!cached) {
            bb::rccl().CommDestroy(c);
        }
    }
    for (void *m : s->peer_opened) hipIpcCloseMemHandle(m);
    hipFree(s->peer_arena);
    hipFree(s->d_peer_table);
    hipFree(s->d_peer_table_x);
    hipFree(s->d_peer_state);
    hipFree(s->d_peer_counter);
    hipFree(s->d_peer_mask);
    for (hipEvent_t e : s->ev) hipEventDestroy(e);
    hipFree(s->d_arena);
    hipFree(s->d_bin_scale);
    hipFree(s->d_map_ptr);
    hipFree(s->d_map_idx);
    hipFree(s->d_map_scalar);
    if (s->own_exch) hipFree(s->d_exch);
    hipFree(s->d_mv_in);
    // (a stream abandoned behind a hung collective is not returned to the pool: the pool
    // synchronises what it takes back)
    if (s->own_stream && s->stream && !s->stream_stuck) bb::release_stream(s->device, s->stream);
    delete s;
    // tear-down is best effort (a free can fail when a peer process has already gone
    // away): whatever it left in the thread's error word is consumed here
    (void)hipGetLastError();
    return BB_OK;
}

int bb_solver_set_stream(bb_solver *s, void *hip_stream) {
    BB_REQUIRE(s != nullptr, "bb_solver_set_stream: solver is NULL");
    BB_TRY(bb::enter_device(s->device));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    if (s->own_stream && s->stream) bb::release_stream(s->device, s->stream);
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
    int rc = BB_BY_LAYOUT(s, set_wish_dense_t, s, host, ld, kind, alpha, (int64_t)0, s->L.n_bins);
    if (rc == BB_OK) rc = refresh_full(s);
    if (rc == BB_OK) s->have_wish = true;
    return rc;
}

int bb_solver_set_maps(bb_solver *s, int n_maps, const int64_t *bin_begin, const double *lr_scale) {
    BB_REQUIRE(s != nullptr && bin_begin != nullptr && lr_scale != nullptr,
               "bb_solver_set_maps: NULL argument");
    BB_REQUIRE(n_maps >= 1 && n_maps <= s->L.n_blocks, "bb_solver_set_maps: bad n_maps");
    if (s->world != 1)
        return bb::fail(BB_ERR_STATE, "bb_solver_set_maps: one rank only (on several GPUs give every "
                                      "rank maps of its own)");
    const int64_t vw = s->L.vw, nb = s->L.n_blocks;
    BB_REQUIRE(bin_begin[0] == 0 && bin_begin[n_maps] == s->L.n_bins,
               "bb_solver_set_maps: bin_begin must run from 0 to n_bins");
    for (int m = 0; m < n_maps; ++m)
        BB_REQUIRE(bin_begin[m] % vw == 0 && bin_begin[m + 1] > bin_begin[m],
                   "bb_solver_set_maps: every map starts at a multiple of the tile edge and is not empty");
    BB_TRY(bb::enter_device(s->device));
    std::vector<int> map_of_block((size_t)nb, 0);
    for (int m = 0, b = 0; b < nb; ++b) {
        while (m + 1 < n_maps && (int64_t)b * vw >= bin_begin[m + 1]) ++m;
        map_of_block[(size_t)b] = m;
    }
    for (size_t t = 0; t < s->tile_I.size(); ++t)
        BB_REQUIRE(map_of_block[(size_t)s->tile_I[t]] == map_of_block[(size_t)s->tile_J[t]],
                   "bb_solver_set_maps: a tile of the solver's list joins two maps");
    // per bin: its map's step; per map: its stress partials (wave w's own, index w, belongs
    // to the strip the wave ends in; slot q's, index n_waves + q, to the slot's strip)
    std::vector<double> scale((size_t)s->L.n_pad);
    for (int64_t i = 0; i < s->L.n_pad; ++i) scale[(size_t)i] = lr_scale[map_of_block[(size_t)(i / vw)]];
    std::vector<std::vector<int>> lists((size_t)n_maps);
    for (int w = 0; w < s->n_waves; ++w)
        if (s->wave_last_strip[(size_t)w] >= 0)
            lists[(size_t)map_of_block[(size_t)s->wave_last_strip[(size_t)w]]].push_back(w);
    for (int q = 0; q < s->n_slots; ++q)
        lists[(size_t)map_of_block[(size_t)s->slot_strip[(size_t)q]]].push_back(s->n_waves + q);
    std::vector<int> ptr(1, 0), idx;
    for (auto &l : lists) {
        idx.insert(idx.end(), l.begin(), l.end());
        ptr.push_back((int)idx.size());
    }
    if (idx.empty()) idx.push_back(0);
    hipFree(s->d_bin_scale); hipFree(s->d_map_ptr); hipFree(s->d_map_idx); hipFree(s->d_map_scalar);
    s->d_bin_scale = nullptr; s->d_map_ptr = s->d_map_idx = nullptr; s->d_map_scalar = nullptr;
    s->bin_steps = false;          // (the maps' steps replace a bb_solver_set_bin_steps)
    s->n_maps = 1;                 // (what holds if an allocation below fails: one map, no tables)
    s->map_begin.clear();
    BB_TRY(upload_bin_scale(s, scale.data()));
    BB_TRY(dev_alloc(&s->d_map_ptr, (int64_t)ptr.size()));
    BB_TRY(dev_alloc(&s->d_map_idx, (int64_t)idx.size()));
    BB_TRY(dev_alloc(&s->d_map_scalar, n_maps));
    BB_HIP_CHECK(hipMemcpy(s->d_map_ptr, ptr.data(), ptr.size() * sizeof(int), hipMemcpyHostToDevice));
    BB_HIP_CHECK(hipMemcpy(s->d_map_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    // (a map that is never set carries no constraint)
    BB_HIP_CHECK(hipMemsetAsync(s->d_units, 0, (size_t)std::max<int64_t>(s->n_local, 1) * bb::kUnitBytes,
                                s->stream));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    s->n_maps = n_maps;
    s->map_begin.assign(bin_begin, bin_begin + n_maps + 1);
    s->row_owner = false;          // the sweep: its reduce knows the maps
    s->hist_n = 0;
    return BB_OK;
}

int bb_solver_set_bin_steps(bb_solver *s, const double *scale, int64_t n_bins) {
    BB_REQUIRE(s != nullptr, "bb_solver_set_bin_steps: solver is NULL");
    if (s->n_maps > 1 && scale == nullptr)
        return bb::fail(BB_ERR_STATE, "bb_solver_set_bin_steps: a solver of several maps always has "
                                      "factors (bb_solver_set_maps sets each map's; pass new ones)");
    if (s->grad_pending)
        return bb::fail(BB_ERR_STATE, "bb_solver_set_bin_steps: a bb_solver_grad is pending");
    BB_TRY(bb::enter_device(s->device));
    if (scale == nullptr) {                       // back to one step for all
        BB_HIP_CHECK(hipStreamSynchronize(s->stream));
        hipFree(s->d_bin_scale);
        s->d_bin_scale = nullptr;
        s->bin_steps = false;
        return BB_OK;
    }
    BB_REQUIRE(n_bins == s->L.n_bins, "bb_solver_set_bin_steps: one factor per bin");
    std::vector<double> padded((size_t)s->L.n_pad, 1.0);
    for (int64_t i = 0; i < n_bins; ++i) {
        BB_REQUIRE(std::isfinite(scale[i]) && scale[i] > 0.0,
                   "bb_solver_set_bin_steps: factors must be finite and positive");
        padded[(size_t)i] = scale[i];
    }
    BB_TRY(upload_bin_scale(s, padded.data()));
    s->bin_steps = true;           // (the sweep's reduce and the row-owner kernel both know them)
    return BB_OK;
}

int bb_solver_set_block_steps(bb_solver *s, const double *scale, int64_t n_blocks) {
    BB_REQUIRE(s != nullptr, "bb_solver_set_block_steps: solver is NULL");
    if (scale == nullptr) return bb_solver_set_bin_steps(s, nullptr, 0);
    BB_REQUIRE(n_blocks == s->L.n_blocks, "bb_solver_set_block_steps: one factor per block of the "
                                          "layout (bb_solver_layout: n_blocks)");
    std::vector<double> per_bin((size_t)s->L.n_bins);
    for (int64_t i = 0; i < s->L.n_bins; ++i) per_bin[(size_t)i] = scale[i / s->L.vw];
    return bb_solver_set_bin_steps(s, per_bin.data(), s->L.n_bins);
}

int bb_solver_degrees(bb_solver *s, int64_t *degree, int64_t n_bins) {
    BB_REQUIRE(s != nullptr && degree != nullptr, "bb_solver_degrees: NULL argument");
    BB_REQUIRE(n_bins == s->L.n_bins, "bb_solver_degrees: one count per bin");
    if (!s->have_wish) return bb::fail(BB_ERR_STATE, "bb_solver_degrees: no wish distances set");
    BB_TRY(bb::enter_device(s->device));
    // (the staging buffer of set_coords / get_coords serves: 3 n_pad doubles, nobody else's
    // while this call runs -- every user of it synchronises before it returns)
    int *d_deg = (int *)s->d_f64_tmp;
    BB_HIP_CHECK(hipMemsetAsync(d_deg, 0, (size_t)s->L.n_pad * sizeof(int), s->stream));
    if (s->n_local > 0) {
#define BB_DEG(TT, WW)                                                                            \
    BB_HIP_CHECK(bb::launch(unit_degrees_kernel<TT, WW>,                                           \
                            dim3((unsigned)((s->n_local + kDegUnits - 1) / kDegUnits)), dim3(256), 0, \
                            s->stream, (const TT *)s->d_units, s->d_udesc, s->n_local, s->L.n_bins,  \
                            d_deg))
        if (s->dtype == BB_F32) BB_DEG(float, true);
        else if (s->wide) BB_DEG(double, true);
        else BB_DEG(double, false);
#undef BB_DEG
    }
    std::vector<int> host((size_t)n_bins);
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    BB_HIP_CHECK(hipMemcpy(host.data(), d_deg, (size_t)n_bins * sizeof(int), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n_bins; ++i) degree[i] = host[(size_t)i];
    return BB_OK;
}

int bb_solver_set_wish_from_cm_block(bb_solver *s, const bb_cm *cm, int64_t bin_offset, int kind,
                                     double alpha) {
    BB_REQUIRE(s != nullptr && cm != nullptr, "bb_solver_set_wish_from_cm_block: NULL argument");
    BB_REQUIRE(kind == BB_KIND_WISH || kind == BB_KIND_COUNTS,
               "bb_solver_set_wish_from_cm_block: bad kind");
    BB_REQUIRE(kind == BB_KIND_WISH || alpha > 0.0, "bb_solver_set_wish_from_cm_block: alpha must be > 0");
    const double *m = nullptr;
    int64_t d = 0;
    int dev = -1;
    BB_TRY(bb_cm_device_ptr(cm, &m, &d, &dev));
    BB_REQUIRE(d >= 1 && bin_offset >= 0 && bin_offset % s->L.vw == 0 && bin_offset + d <= s->L.n_bins,
               "bb_solver_set_wish_from_cm_block: the map does not fit the solver at that offset");
    BB_REQUIRE(dev == s->device,
               "bb_solver_set_wish_from_cm_block: the map lives on another device than the solver");
    BB_TRY(bb::enter_device(s->device));
    if (s->n_local > 0) {
#define BB_PACKB(TT, WW)                                                                         \
    BB_HIP_CHECK(bb::launch(pack_units_from_matrix_kernel<TT, WW>, dim3((unsigned)s->n_local),       \
                            dim3(256), 0, s->stream, m, d, (TT *)s->d_units, s->d_udesc,          \
                            bin_offset + d, kind, -1.0 / alpha, bin_offset))
        if (s->dtype == BB_F32) BB_PACKB(float, true);
        else if (s->wide) BB_PACKB(double, true);
        else BB_PACKB(double, false);
#undef BB_PACKB
    }
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    BB_TRY(refresh_full(s));       // (a small one-map solver iterates over the full matrix)
    s->have_wish = true;
    return BB_OK;
}

int bb_solver_set_wish_dense_block(bb_solver *s, const double *host, int64_t ld, int64_t n_sub,
                                   int64_t bin_offset, int kind, double alpha) {
    BB_REQUIRE(s != nullptr && host != nullptr, "bb_solver_set_wish_dense_block: NULL argument");
    BB_REQUIRE(n_sub >= 1 && bin_offset >= 0 && bin_offset + n_sub <= s->L.n_bins && ld >= n_sub,
               "bb_solver_set_wish_dense_block: the block does not fit the solver");
    BB_REQUIRE(bin_offset % s->L.vw == 0,
               "bb_solver_set_wish_dense_block: a block starts at a multiple of the tile edge");
    BB_REQUIRE(kind == BB_KIND_WISH || kind == BB_KIND_COUNTS,
               "bb_solver_set_wish_dense_block: bad kind");
    BB_REQUIRE(kind == BB_KIND_WISH || alpha > 0.0, "bb_solver_set_wish_dense_block: alpha must be > 0");
    BB_TRY(bb::enter_device(s->device));
    int rc = BB_BY_LAYOUT(s, set_wish_dense_t, s, host, ld, kind, alpha, bin_offset, n_sub);
    if (rc == BB_OK) rc = refresh_full(s);
    if (rc == BB_OK) s->have_wish = true;
    return rc;
}

int bb_solver_set_wish_from_cm_block(bb_solver *s, const bb_cm *cm, int64_t bin_offset, int kind,
                                     double alpha);
int bb_solver_set_wish_from_cm(bb_solver *s, const bb_cm *cm, int kind, double alpha) {
    BB_REQUIRE(s != nullptr && cm != nullptr, "bb_solver_set_wish_from_cm: NULL argument");
    if (s->n_maps > 1)
        return bb::fail(BB_ERR_STATE, "bb_solver_set_wish_from_cm: a solver of several maps takes "
                                      "them one by one (bb_solver_set_wish_from_cm_block)");
    BB_REQUIRE(kind == BB_KIND_WISH || kind == BB_KIND_COUNTS,
               "bb_solver_set_wish_from_cm: bad kind");
    BB_REQUIRE(kind == BB_KIND_WISH || alpha > 0.0, "bb_solver_set_wish_from_cm: alpha must be > 0");
    const double *m = nullptr;
    int64_t d = 0;
    int dev = -1;
    BB_TRY(bb_cm_device_ptr(cm, &m, &d, &dev));
    BB_REQUIRE(d == s->L.n_bins,
               "bb_solver_set_wish_from_cm: the map's edge differs from the solver's n_bins");
    BB_REQUIRE(dev == s->device,
               "bb_solver_set_wish_from_cm: the map lives on another device than the solver");
    BB_TRY(bb::enter_device(s->device));
    if (s->n_local > 0) {
#define BB_PACK(TT, WW)                                                                          \
    BB_HIP_CHECK(bb::launch(pack_units_from_matrix_kernel<TT, WW>, dim3((unsigned)s->n_local),       \
                            dim3(256), 0, s->stream, m, d, (TT *)s->d_units, s->d_udesc,          \
                            s->L.n_bins, kind, -1.0 / alpha, (int64_t)0))
        if (s->dtype == BB_F32) BB_PACK(float, true);
        else if (s->wide) BB_PACK(double, true);
        else BB_PACK(double, false);
#undef BB_PACK
    }
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    BB_TRY(refresh_full(s));
    s->have_wish = true;
    return BB_OK;
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
            const EntrySrc src = {d_rows, d_cols, d_vals, nullptr, 0, 0, 1.0};
#define BB_SCATTER(TT, WW, PH)                                                                  \
    bb::launch(scatter_entries_kernel<TT, WW>, dim3(grid), dim3(256), 0, s->stream, src, m, d_map, \
               nb, s->L.n_bins, s->u_begin, s->u_end, (TT *)s->d_units, kind, -1.0 / alpha, d_kr,  \
               d_ke, d_bad, PH)
            // clear / find the last entry per cell / store it (later chunks simply
            // repeat this on top of earlier ones, so "last wins" holds across chunks)
            for (int ph = 0; ph < 3 && e == hipSuccess; ++ph) {
                if (s->dtype == BB_F32) e = BB_SCATTER(float, true, ph);
                else if (s->wide) e = BB_SCATTER(double, true, ph);
                else e = BB_SCATTER(double, false, ph);
            }
#undef BB_SCATTER
            // the staging buffers are reused by the next chunk: stream order makes that safe
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        if (e == hipSuccess) e = hipMemcpy(&host_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost);
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
    BB_TRY(refresh_full(s));
    s->have_wish = true;
    return BB_OK;
}

// ---- Rao-format triples resident on the device (fit_triples without host binning) --------
struct bb_triples {
    int device = 0;
    int64_t n = 0, st = 3, sc = 1;      // element (t, c) at d[t * st + c * sc]
    double resolution = 1.0;
    double *d = nullptr;
};

int bb_triples_create(bb_triples **out, const double *triples, int64_t n, int32_t resolution,
                      int32_t row_major, int device) {
    BB_REQUIRE(out != nullptr, "bb_triples_create: out is NULL");
    *out = nullptr;
    BB_REQUIRE(n >= 0 && (n == 0 || triples != nullptr), "bb_triples_create: NULL triples");
    BB_REQUIRE(resolution > 0, "bb_triples_create: resolution must be positive");
    BB_TRY(bb::use_device(device));
    bb_triples *t = new (std::nothrow) bb_triples();
    if (!t) return bb::fail(BB_ERR_NOMEM, "bb_triples_create: out of host memory");
    t->device = device;
    t->n = n;
    t->resolution = (double)resolution;
    if (row_major) { t->st = 3; t->sc = 1; } else { t->st = 1; t->sc = n; }
    int rc = dev_alloc(&t->d, 3 * n);
    if (rc == BB_OK && n > 0) {
        const hipError_t e = hipMemcpy(t->d, triples, (size_t)n * 24, hipMemcpyHostToDevice);
        if (e != hipSuccess)
            rc = bb::fail(BB_ERR_HIP, std::string("bb_triples_create: ") + hipGetErrorString(e));
    }
    if (rc != BB_OK) {
        hipFree(t->d);
        delete t;
        return rc;
    }
    *out = t;
    return BB_OK;
}

int bb_triples_destroy(bb_triples *t) {
    if (!t) return BB_OK;
    hipSetDevice(t->device);
    hipFree(t->d);
    delete t;
    (void)hipGetLastError();
    return BB_OK;
}

int bb_triples_tiles(const bb_triples *t, int64_t n_bins, int dtype, uint8_t *present,
                     int64_t n_blocks) {
    BB_REQUIRE(t != nullptr && present != nullptr, "bb_triples_tiles: NULL argument");
    bb_layout_info L;
    BB_TRY(bb_layout_dense_info(n_bins, dtype, &L));
    BB_REQUIRE(n_blocks == L.n_blocks, "bb_triples_tiles: present must hold n_blocks^2 bytes of "
                                       "the layout of (n_bins, dtype)");
    BB_TRY(bb::enter_device(t->device));
    bb::DevBuf bp, bb_;
    if (bp.alloc((size_t)(n_blocks * n_blocks)) != hipSuccess || bb_.alloc(sizeof(int)) != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, "bb_triples_tiles: out of device memory");
    BB_HIP_CHECK(hipMemset(bp.p, 0, (size_t)(n_blocks * n_blocks)));
    BB_HIP_CHECK(hipMemset(bb_.p, 0, sizeof(int)));
    constexpr int64_t kChunk = (int64_t)1 << 30;
    for (int64_t k0 = 0; k0 < t->n; k0 += kChunk) {
        const int64_t m = std::min(kChunk, t->n - k0);
        const EntrySrc src = {nullptr, nullptr, nullptr, t->d + k0 * (t->st == 3 ? 3 : 1), t->st, t->sc,
                              t->resolution};
        BB_HIP_CHECK(bb::launch(entries_tiles_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0,
                                (hipStream_t) nullptr, src, m, L.vw, n_blocks, n_bins,
                                (unsigned char *)bp.p, (int *)bb_.p));
    }
    int bad = 0;
    BB_HIP_CHECK(hipMemcpy(&bad, bb_.p, sizeof(int), hipMemcpyDeviceToHost));
    BB_HIP_CHECK(hipMemcpy(present, bp.p, (size_t)(n_blocks * n_blocks), hipMemcpyDeviceToHost));
    if (bad) return bb::fail(BB_ERR_INVALID, "bb_triples_tiles: an index is outside [0, n_bins)");
    return BB_OK;
}

int bb_solver_set_wish_triples(bb_solver *s, const bb_triples *t, int kind, double alpha,
                               const double *KRnorm, const double *KRexpected) {
    BB_REQUIRE(s != nullptr && t != nullptr, "bb_solver_set_wish_triples: NULL argument");
    BB_REQUIRE(kind == BB_KIND_WISH || kind == BB_KIND_COUNTS, "bb_solver_set_wish_triples: bad kind");
    BB_REQUIRE(kind == BB_KIND_WISH || alpha > 0.0, "bb_solver_set_wish_triples: alpha must be > 0");
    BB_REQUIRE((KRnorm == nullptr) == (KRexpected == nullptr),
               "bb_solver_set_wish_triples: KRnorm and KRexpected go together");
    BB_REQUIRE(t->device == s->device,
               "bb_solver_set_wish_triples: the triples live on another device than the solver");
    BB_TRY(bb::enter_device(s->device));
    const int64_t nb = s->L.n_blocks;
    std::vector<int32_t> tilemap((size_t)(nb * nb), -1);
    for (size_t q = 0; q < s->tile_I.size(); ++q)
        tilemap[(size_t)s->tile_I[q] * nb + s->tile_J[q]] = (int32_t)q;
    bb::DevBuf bmap, bbad, bkr, bke;
    if (bmap.alloc((size_t)(nb * nb) * 4) != hipSuccess || bbad.alloc(sizeof(int)) != hipSuccess ||
        (KRnorm && (bkr.alloc((size_t)s->L.n_bins * 8) != hipSuccess ||
                    bke.alloc((size_t)s->L.n_bins * 8) != hipSuccess)))
        return bb::fail(BB_ERR_NOMEM, "bb_solver_set_wish_triples: out of device memory");
    hipStream_t st = s->stream;
    BB_HIP_CHECK(hipMemcpyAsync(bmap.p, tilemap.data(), tilemap.size() * 4, hipMemcpyHostToDevice, st));
    if (KRnorm) {
        BB_HIP_CHECK(hipMemcpyAsync(bkr.p, KRnorm, (size_t)s->L.n_bins * 8, hipMemcpyHostToDevice, st));
        BB_HIP_CHECK(hipMemcpyAsync(bke.p, KRexpected, (size_t)s->L.n_bins * 8, hipMemcpyHostToDevice, st));
    }
    BB_HIP_CHECK(hipMemsetAsync(bbad.p, 0, sizeof(int), st));
    BB_HIP_CHECK(hipMemsetAsync(s->d_units, 0, (size_t)std::max<int64_t>(s->n_local, 1) * bb::kUnitBytes, st));
    // chunks of 2^30 entries: an fp32 cell holds the index of the entry that wins it in 32 bits;
    // later chunks run on top of earlier ones, so "the last entry wins" holds across them
    constexpr int64_t kChunk = (int64_t)1 << 30;
    for (int64_t k0 = 0; k0 < t->n; k0 += kChunk) {
        const int64_t m = std::min(kChunk, t->n - k0);
        const EntrySrc src = {nullptr, nullptr, nullptr, t->d + k0 * (t->st == 3 ? 3 : 1), t->st, t->sc,
                              t->resolution};
        const unsigned grid = (unsigned)((m + 255) / 256);
        for (int ph = 0; ph < 3; ++ph) {
#define BB_SCATTER_T(TT, WW)                                                                       \
    BB_HIP_CHECK(bb::launch(scatter_entries_kernel<TT, WW>, dim3(grid), dim3(256), 0, st, src, m,   \
                            (const int32_t *)bmap.p, nb, s->L.n_bins, s->u_begin, s->u_end,        \
                            (TT *)s->d_units, kind, -1.0 / alpha, (const double *)bkr.p,            \
                            (const double *)bke.p, (int *)bbad.p, ph))
            if (s->dtype == BB_F32) BB_SCATTER_T(float, true);
            else if (s->wide) BB_SCATTER_T(double, true);
            else BB_SCATTER_T(double, false);
#undef BB_SCATTER_T
        }
    }
    BB_HIP_CHECK(hipStreamSynchronize(st));
    int bad = 0;
    BB_HIP_CHECK(hipMemcpy(&bad, bbad.p, sizeof(int), hipMemcpyDeviceToHost));
    if (bad == 1)
        return bb::fail(BB_ERR_INVALID, "bb_solver_set_wish_triples: an index is outside [0, n_bins)");
    if (bad == 2)
        return bb::fail(BB_ERR_INVALID, "bb_solver_set_wish_triples: an entry falls in a tile that is "
                                        "not in the solver's tile list");
    BB_TRY(refresh_full(s));
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
#define BB_GEN(TT, WW)                                                                         \
    BB_HIP_CHECK(bb::launch(gen_units_kernel<TT, WW>, dim3((unsigned)s->n_local), dim3(256), 0,    \
                            s->stream, s->d_f64_tmp, (TT *)s->d_units, s->d_udesc, s->L.n_bins))
        if (s->dtype == BB_F32) BB_GEN(float, true);
        else if (s->wide) BB_GEN(double, true);
        else BB_GEN(double, false);
#undef BB_GEN
    }
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    BB_TRY(refresh_full(s));
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
    BB_HIP_CHECK(narrow(s, s->d_f64_tmp, s->d_X, n3));
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
    BB_HIP_CHECK(widen(s, s->d_X, s->d_f64_tmp, n3));
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
    if ((s->hist_n + iters) * s->n_maps > s->hist_cap)
        return bb::fail(BB_ERR_STATE, "bb_solver_iterate: stress history full");
    BB_TRY(bb::enter_device(s->device));
    if (s->row_owner) {
        // one launch per iteration; launch k also folds the stress of launch k-1
        for (int64_t k = 0; k < iters; ++k) {
            hipEvent_t *ev = timing_slot(s);
            if (ev) BB_HIP_CHECK(hipEventRecord(ev[0], s->stream));
            BB_TRY(launch_row_owner(s, lr, k > 0, true));
            if (ev) BB_HIP_CHECK(hipEventRecord(ev[1], s->stream));
            if (ev) BB_HIP_CHECK(hipEventRecord(ev[2], s->stream));
            s->hist_n++;
        }
        if (iters > 0) BB_TRY(launch_row_owner(s, lr, true, false));
        return BB_OK;
    }
    for (int64_t k = 0; k < iters; ++k) {
        hipEvent_t *ev = timing_slot(s);
        if (ev) BB_HIP_CHECK(hipEventRecord(ev[0], s->stream));
        BB_TRY(launch_grad(s));
        if (ev) BB_HIP_CHECK(hipEventRecord(ev[1], s->stream));
        BB_TRY(launch_reduce(s, kReduceApply, lr, s->d_stress_hist + s->hist_n * s->n_maps));
        if (ev) BB_HIP_CHECK(hipEventRecord(ev[2], s->stream));
        s->hist_n++;
    }
    return BB_OK;
}

int bb_solver_grad(bb_solver *s) {
    BB_TRY(check_ready(s, "bb_solver_grad"));
    if (s->n_maps > 1)
        return bb::fail(BB_ERR_STATE, "bb_solver_grad: a solver of several maps (bb_solver_set_maps) "
                                      "iterates with bb_solver_iterate only");
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
        BB_HIP_CHECK(bb::launch(apply_kernel<float>, dim3(grid), dim3(256), 0, s->stream,
                                (float *)s->d_X, (float *)s->d_V, (const float *)s->d_exch, n3,
                                (float)lr, (float)s->momentum, s->d_stress_hist + s->hist_n));
    else
        BB_HIP_CHECK(bb::launch(apply_kernel<double>, dim3(grid), dim3(256), 0, s->stream,
                                (double *)s->d_X, (double *)s->d_V, (const double *)s->d_exch, n3,
                                lr, s->momentum, s->d_stress_hist + s->hist_n));
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

// The library's communicators outlive the solvers that use them: ncclCommInitRank costs
// 0.1-1 s and every multi-rank fit() used to pay it (round 2 made a fresh communicator per
// fit and destroyed it with the solver).  One communicator per (device, rank, world) is
// kept for the life of the process; a solver borrows it (bb_solver_comm_attach) and gives
// it back when it is destroyed.  While one solver holds it, another one gets a
// communicator of its own (a communicator must not serve two streams at once).
namespace {
struct CachedComm {
    bb::Rccl::Comm comm = nullptr;
    bool in_use = false;
    // Which ncclCommInitRank made it: a hash of the 128-byte unique id, the same on every rank
    // of that call and on no other.  The key (device, rank, world) alone cannot tell two
    // communicators of different jobs -- or of different generations of one job -- apart; the
    // ranks compare this before anybody attaches (bb_comm_cached_generation, solver.comm_reuse).
    uint64_t generation = 0;
};
uint64_t comm_generation_of(const void *unique_id) {
    uint64_t h = 1469598103934665603ull;             // FNV-1a over the id
    const unsigned char *b = (const unsigned char *)unique_id;
    for (size_t i = 0; i < bb::kUniqueIdBytes; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h ? h : 1;                                // 0 = "none"
}
std::mutex g_comm_mu;
std::map<std::tuple<int, int, int>, CachedComm> g_comm_cache;

void comm_release(bb_solver *s, bool destroy) {
    if (!s->comm) return;
    std::lock_guard<std::mutex> lock(g_comm_mu);
    if (s->comm_cached) {
        auto it = g_comm_cache.find(std::make_tuple(s->device, s->rank, s->world));
        if (it != g_comm_cache.end() && it->second.comm == s->comm) {
            if (destroy || s->comm_suspect)
                g_comm_cache.erase(it);       // a suspect communicator is not handed out again
            else
                it->second.in_use = false;
        }
    }
    s->comm = nullptr;
    s->comm_cached = false;
    s->comm_suspect = false;
}
}  // namespace

int bb_comm_cached_generation(int device, int rank, int world, int *available,
                              uint64_t *generation) {
    BB_REQUIRE(available != nullptr && generation != nullptr,
               "bb_comm_cached_generation: NULL argument");
    std::lock_guard<std::mutex> lock(g_comm_mu);
    auto it = g_comm_cache.find(std::make_tuple(device, rank, world));
    const bool have = it != g_comm_cache.end() && it->second.comm && !it->second.in_use;
    *available = have ? 1 : 0;
    *generation = have ? it->second.generation : 0;
    return BB_OK;
}

int bb_comm_cached(int device, int rank, int world, int *available) {
    BB_REQUIRE(available != nullptr, "bb_comm_cached: available is NULL");
    std::lock_guard<std::mutex> lock(g_comm_mu);
    auto it = g_comm_cache.find(std::make_tuple(device, rank, world));
    *available = (it != g_comm_cache.end() && it->second.comm && !it->second.in_use) ? 1 : 0;
    return BB_OK;
}

int bb_solver_comm_attach(bb_solver *s) {
    BB_REQUIRE(s != nullptr, "bb_solver_comm_attach: solver is NULL");
    if (s->comm) return bb::fail(BB_ERR_STATE, "bb_solver_comm_attach: communicator already made");
    std::lock_guard<std::mutex> lock(g_comm_mu);
    auto it = g_comm_cache.find(std::make_tuple(s->device, s->rank, s->world));
    if (it == g_comm_cache.end() || !it->second.comm || it->second.in_use)
        return bb::fail(BB_ERR_STATE, "bb_solver_comm_attach: no free cached communicator");
    it->second.in_use = true;
    s->comm = it->second.comm;
    s->comm_cached = true;
    return BB_OK;
}

int bb_solver_comm_detach(bb_solver *s) {
    BB_REQUIRE(s != nullptr, "bb_solver_comm_detach: solver is NULL");
    if (s->comm && !s->comm_cached)
        return bb::fail(BB_ERR_STATE, "bb_solver_comm_detach: the communicator is the solver's own");
    comm_release(s, /*destroy=*/false);
    return BB_OK;
}

int bb_comm_cache_clear(void) {
    const bb::Rccl &R = bb::rccl();
    std::lock_guard<std::mutex> lock(g_comm_mu);
    for (auto it = g_comm_cache.begin(); it != g_comm_cache.end();) {
        if (it->second.in_use) { ++it; continue; }
        if (it->second.comm && R.ok) R.CommDestroy(it->second.comm);
        it = g_comm_cache.erase(it);
    }
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
    {
        // into the cache, unless another solver holds this key's entry right now
        std::lock_guard<std::mutex> lock(g_comm_mu);
        CachedComm &c = g_comm_cache[std::make_tuple(s->device, s->rank, s->world)];
        if (!c.in_use) {
            if (c.comm && c.comm != s->comm) R.CommDestroy(c.comm);   // a stale one: replaced
            c.comm = s->comm;
            c.in_use = true;
            c.generation = comm_generation_of(unique_id);
            s->comm_cached = true;
        }
    }
    return BB_OK;
}

int bb_solver_comm_abort(bb_solver *s) {
    BB_REQUIRE(s != nullptr, "bb_solver_comm_abort: solver is NULL");
    if (!s->comm) return BB_OK;
    const bb::Rccl &R = bb::rccl();
    bb::Rccl::Comm c = s->comm;
    comm_release(s, /*destroy=*/true);       // out of the cache: it is not handed out again
    s->grad_pending = false;
    if (!R.CommAbort) {
        // No ncclCommAbort in this librccl.  ncclCommDestroy waits for the communicator's
        // outstanding work -- on a communicator that may sit in a collective a peer never
        // joined that is for ever, which is exactly what this call exists to end.  The
        // communicator is leaked instead and the solver's stream is marked stuck: nothing
        // synchronises on it any more (bb_solver_destroy, the stream pool).
        s->stream_stuck = true;
        return bb::fail(BB_ERR_STATE,
                        "bb_solver_comm_abort: this librccl has no ncclCommAbort; the "
                        "communicator was leaked and the solver's stream abandoned");
    }
    // ncclCommAbort ends the communicator's in-flight kernels, so a stream that is
    // stuck behind a collective a peer never joined drains again
    const int rc = R.CommAbort(c);
    if (rc != bb::Rccl::kSuccess)
        return bb::fail(BB_ERR_HIP, std::string("ncclCommAbort: ") + R.GetErrorString(rc));
    return BB_OK;
}

int bb_solver_comm_world(const bb_solver *s, int *world) {
    BB_REQUIRE(s != nullptr && world != nullptr, "bb_solver_comm_world: NULL argument");
    *world = 0;
    if (!s->comm) return bb::fail(BB_ERR_STATE, "bb_solver_comm_world: no communicator");
    const bb::Rccl &R = bb::rccl();
    if (!R.CommCount) return bb::fail(BB_ERR_HIP, "bb_solver_comm_world: ncclCommCount missing");
    const int rc = R.CommCount(s->comm, world);
    if (rc != bb::Rccl::kSuccess)
        return bb::fail(BB_ERR_HIP, std::string("ncclCommCount: ") + R.GetErrorString(rc));
    return BB_OK;
}

namespace {
int enqueue_allreduce(bb_solver *s) {
    const bb::Rccl &R = bb::rccl();
    const int rc = R.AllReduce(s->d_exch, s->d_exch, (size_t)(3 * s->L.n_pad + 2),
                               s->dtype == BB_F32 ? bb::Rccl::kFloat32 : bb::Rccl::kFloat64,
                               bb::Rccl::kSum, s->comm, s->stream);
    if (rc != bb::Rccl::kSuccess) {
        s->comm_suspect = true;      // its peers may now sit in a collective this rank left
        return bb::fail(BB_ERR_HIP, std::string("ncclAllReduce: ") + R.GetErrorString(rc));
    }
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

}  // extern "C"

namespace {
// The receiving half of the two-launch peer exchange (peer_receive_kernel): wait for every
// rank's flag of exchange `peer_seq`, X <- X + (mu V - lr * sum over the ranks in rank order).
int launch_peer_receive(bb_solver *s, void *X, double lr, double mu, double *stress_out) {
    const int64_t n3 = s->L.n_pad * 3, es = bb::elem_size(s->dtype);
    // Ranks that share a GPU (a rehearsal) share its wave slots too: the receive kernels of ALL
    // of them wait at once, and eight times 256 workgroups of 4 waves are every slot the chip
    // has -- nobody's sweep could start (found by tools/world8_rehearsal.py).  The cap is
    // divided among them; a thread then takes more elements.
    const int64_t cap = std::max<int64_t>(8, (int64_t)kPeerReceiveWGs / (s->peer_ranks_on_gpu > 2 ? s->peer_ranks_on_gpu / 2 : 1));
    const unsigned grid = (unsigned)std::min<int64_t>((n3 + 255) / 256, cap);
    const char *arena = (const char *)s->peer_arena +
                        (int64_t)(s->peer_seq & 1) * s->world * s->peer_slot_elems * es;
    const unsigned long long *flags =
        (const unsigned long long *)((const char *)s->peer_arena + peer_flags_offset(s));
    // the flag pointers are the second half of a PeerTable (same for both parities)
    unsigned long long *const *poison =
        (unsigned long long *const *)((const char *)s->d_peer_table + kMaxPeers * sizeof(void *));
    if (s->dtype == BB_F32)
        BB_HIP_CHECK(bb::launch(peer_receive_kernel<float>, dim3(grid), dim3(256), 0, s->stream,
                                (float *)X, (float *)s->d_V, (const float *)arena, flags, poison,
                                s->world, s->peer_slot_elems, n3, (float)lr, (float)mu, stress_out,
                                s->peer_seq, s->d_peer_state, s->peer_limit_ticks,
                                (const unsigned *)s->d_peer_mask, (int)(3 * s->L.vw)));
    else
        BB_HIP_CHECK(bb::launch(peer_receive_kernel<double>, dim3(grid), dim3(256), 0, s->stream,
                                (double *)X, (double *)s->d_V, (const double *)arena, flags, poison,
                                s->world, s->peer_slot_elems, n3, lr, mu, stress_out, s->peer_seq,
                                s->d_peer_state, s->peer_limit_ticks,
                                (const unsigned *)s->d_peer_mask, (int)(3 * s->L.vw)));
    return BB_OK;
}

// The plain SUM over the ranks of what the last sweep left in the partials (scale 1: a
// matvec, not a gradient), into the exchange buffer [0, 3 n_pad) of every rank, enqueued on
// the solver's stream: the reduce alone on one rank; the peer exchange (either form; the
// ranks' partials are added in rank order, so every rank holds the same bits) when the
// arenas are connected; else the library's RCCL communicator.  The step-taking kernels are
// used as they are -- see bb_solver::sum_target.  d_V is left holding the sum as well.
int exchange_sum(bb_solver *s) {
    const int64_t n3 = s->L.n_pad * 3, es = bb::elem_size(s->dtype);
    if (s->peer_connected) {        // (also a one-rank solver that pushes to itself: a rehearsal)
        BB_HIP_CHECK(hipMemsetAsync(s->d_exch, 0, (size_t)(n3 * es), s->stream));
        s->peer_seq++;
        s->sum_target = s->d_exch;
        int rc;
        if (s->peer_fused && s->red_slices > 0) {
            rc = launch_exchange(s, -1.0, s->d_stress_scalar, kScalePlainSum);
        } else {
            rc = launch_reduce(s, kReducePeer, 0.0, nullptr, kScalePlainSum);
            if (rc == BB_OK) rc = launch_peer_receive(s, s->d_exch, -1.0, 0.0, s->d_stress_scalar);
        }
        s->sum_target = nullptr;
        return rc;
    }
    if (s->comm) {
        BB_TRY(launch_reduce(s, kReduceExchange, 0.0, nullptr, kScalePlainSum));
        return enqueue_allreduce(s);
    }
    if (s->world == 1) return launch_reduce(s, kReduceExchange, 0.0, nullptr, kScalePlainSum);
    return bb::fail(BB_ERR_STATE, "no exchange between the ranks is set up (bb_solver_peer_connect "
                                  "or bb_solver_comm_init / _attach first)");
}
}  // namespace

extern "C" {

int bb_solver_peer_export(bb_solver *s, void *handle_out) {
    BB_REQUIRE(s != nullptr && handle_out != nullptr, "bb_solver_peer_export: NULL argument");
    BB_REQUIRE(s->world <= kMaxPeers, "bb_solver_peer_export: world > 16");
    if (s->peer_arena) return bb::fail(BB_ERR_STATE, "bb_solver_peer_export: already exported");
    BB_TRY(bb::enter_device(s->device));
    const int64_t es = bb::elem_size(s->dtype);
    s->peer_slot_elems = bb::round_up(3 * s->L.n_pad + 2, 256 / es);
    s->peer_arena_bytes = peer_arena_size(s);
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
    // every data word starts out EMPTY (all bits set: reduce_exchange_kernel), flags and poison
    // words at 0
    BB_HIP_CHECK(hipMemset(s->peer_arena, 0xff, (size_t)peer_flags_offset(s)));
    BB_HIP_CHECK(hipMemset((char *)s->peer_arena + peer_flags_offset(s), 0,
                           (size_t)(s->peer_arena_bytes - peer_flags_offset(s))));
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
    {
        int dom = 0, bus = 0, dev = 0;
        if (hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, s->device) != hipSuccess ||
            hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, s->device) != hipSuccess ||
            hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, s->device) != hipSuccess) {
            (void)hipGetLastError();
            dom = bus = dev = 0;       // unknown: ranks then count as sharing one GPU (the safe side)
        }
        h.pci = ((uint64_t)(uint32_t)dom << 32) | ((uint64_t)(bus & 0xffff) << 16) | (uint64_t)(dev & 0xffff);
    }
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
    std::map<uint64_t, int> ranks_on_gpu;      // how many ranks sit on each GPU of the node
    int most_on_one_gpu = 1;
    for (int r = 0; r < s->world; ++r) {
        PeerHandle h;
        memcpy(&h, (const char *)handles + (size_t)r * BB_PEER_HANDLE_BYTES, sizeof(h));
        most_on_one_gpu = std::max(most_on_one_gpu, ++ranks_on_gpu[h.pci]);
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
    {
        // Who sends what: rank q's units lie in tiles (I, J) and carry gradient for the bins of
        // blocks I and J only; for every other block its partial is zero by construction.  Every
        // rank computes the same table from the tile list and the partition, so senders and
        // receivers agree without a word exchanged.  Dense N=50,000 on 8 ranks: rank 0 touches
        // a third of the blocks (28 % fewer bytes on the links over all ranks); the whole genome
        // as blocks: a rank touches the blocks of a few chromosomes (BB_PEER_MASK=0: all send all).
        const char *env = getenv("BB_PEER_MASK");
        hipFree(s->d_peer_mask);
        s->d_peer_mask = nullptr;
        if (!(env && atoi(env) == 0) && s->world <= 32) {
            std::vector<unsigned> mask((size_t)s->L.n_blocks, 0u);
            const int64_t upt = s->L.units_per_tile;
            for (int q = 0; q < s->world; ++q) {
                int64_t ub = 0, ue = 0;
                BB_TRY(bb_layout_rank_units(s->L.n_units, q, s->world, &ub, &ue));
                if (ue <= ub) continue;
                for (int64_t t = ub / upt; t <= (ue - 1) / upt; ++t) {
                    mask[(size_t)s->tile_I[(size_t)t]] |= 1u << q;
                    mask[(size_t)s->tile_J[(size_t)t]] |= 1u << q;
                }
            }
            BB_TRY(dev_alloc(&s->d_peer_mask, s->L.n_blocks));
            BB_HIP_CHECK(hipMemcpy(s->d_peer_mask, mask.data(), mask.size() * sizeof(unsigned),
                                   hipMemcpyHostToDevice));
        }
    }
    BB_TRY(dev_alloc(&s->d_peer_state, 1));
    // the top counter + one per block on a line of its own (reduce_sliced_kernel's two-level
    // check-in), also good for reduce_kernel, which uses the top one only
    const int64_t n_counters = 32 * (s->L.n_blocks + 1);
    BB_TRY(dev_alloc(&s->d_peer_counter, n_counters));
    BB_HIP_CHECK(hipMemset(s->d_peer_state, 0, sizeof(PeerState)));
    BB_HIP_CHECK(hipMemset(s->d_peer_counter, 0, (size_t)n_counters * sizeof(unsigned)));
    // ticks of wall_clock64(): ask the runtime, fall back to gfx9's 100 MHz.  The query
    // is allowed to fail (older runtimes); its error is consumed here, on the spot.
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, s->device) != hipSuccess) {
        (void)hipGetLastError();
        khz = 0;
    }
    s->peer_clock_khz = khz > 0 ? khz : 100000;
    long long ms = 10000;
    if (const char *env = getenv("BB_PEER_TIMEOUT_MS")) {
        const long long v = atoll(env);
        if (v > 0) ms = v;
    }
    s->peer_limit_ticks = ms * s->peer_clock_khz;
    s->peer_seq = 0;
    {
        // The one-launch exchange (reduce_exchange_kernel) has every workgroup wait for its
        // peers' copies of itself: fine with one rank per GPU (workgroups are dispatched in
        // order and push before they wait), not when ranks SHARE a GPU and the waiting
        // workgroups of some can keep the sweep of another off the device (three ranks at
        // N=7,000 fp64 on one GPU: every CU held a waiting workgroup and 16 KB of its LDS, the
        // third rank's sweep wants 150 KB per workgroup, time-out).  Every rank sees every
        // handle (PCI ids), so all of them decide alike: any GPU with two ranks on it -- a
        // rehearsal -- and it is the two-launch form.  BB_PEER_FUSED=0|1 overrides (the same
        // on every rank; small problems on a shared GPU do run in one launch).
        const char *env = getenv("BB_PEER_FUSED");
        s->peer_fused = env ? atoi(env) != 0 : most_on_one_gpu == 1;
        s->peer_ranks_on_gpu = most_on_one_gpu;
        if (s->red_slices <= 0) s->peer_fused = false;
    }
    s->peer_connected = true;
    BB_HIP_CHECK(hipDeviceSynchronize());
    return BB_OK;
}

int bb_solver_peer_set_timeout(bb_solver *s, int64_t milliseconds) {
    BB_REQUIRE(s != nullptr, "bb_solver_peer_set_timeout: solver is NULL");
    BB_REQUIRE(milliseconds > 0 && milliseconds <= 3600000,
               "bb_solver_peer_set_timeout: need 0 < ms <= 3600000");
    if (!s->peer_connected)
        return bb::fail(BB_ERR_STATE, "bb_solver_peer_set_timeout: not connected");
    s->peer_limit_ticks = (long long)milliseconds * s->peer_clock_khz;
    return BB_OK;
}

int bb_solver_peer_set_form(bb_solver *s, int one_launch) {
    BB_REQUIRE(s != nullptr, "bb_solver_peer_set_form: solver is NULL");
    if (!s->peer_connected)
        return bb::fail(BB_ERR_STATE, "bb_solver_peer_set_form: not connected");
    // the one-launch form reads "empty" in every slot word it has not been sent: it can only
    // be chosen while no exchange has run (the two-launch form leaves its partials in the slots)
    if (s->peer_seq != 0)
        return bb::fail(BB_ERR_STATE, "bb_solver_peer_set_form: exchanges have already run");
    s->peer_fused = one_launch != 0;
    return BB_OK;
}

int bb_solver_peer_form(bb_solver *s, int *one_launch) {
    BB_REQUIRE(s != nullptr && one_launch != nullptr, "bb_solver_peer_form: NULL argument");
    if (!s->peer_connected) return bb::fail(BB_ERR_STATE, "bb_solver_peer_form: not connected");
    *one_launch = (s->peer_fused && s->red_slices > 0) ? 1 : 0;
    return BB_OK;
}

int bb_solver_peer_status(bb_solver *s, int *status) {
    BB_REQUIRE(s != nullptr, "bb_solver_peer_status: solver is NULL");
    if (!s->peer_connected) return bb::fail(BB_ERR_STATE, "bb_solver_peer_status: not connected");
    BB_TRY(bb::enter_device(s->device));
    PeerState st;
    memset(&st, 0, sizeof(st));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    BB_HIP_CHECK(hipMemcpy(&st, s->d_peer_state, sizeof(st), hipMemcpyDeviceToHost));
    if (status) *status = st.status;
    if (st.status != 0) {
        const bool one = s->peer_fused && s->red_slices > 0;
        return bb::fail(BB_ERR_STATE,
                        std::string("peer exchange: a rank did not deliver its partial within the "
                                    "time limit (BB_PEER_TIMEOUT_MS), or reported its own failure; ") +
                            (one ? "the last exchange this rank completed whole was "
                                 : "this rank's coordinates were left at its last completed step (") +
                            std::to_string(st.verdict) + " of " + std::to_string(s->peer_seq) +
                            (one ? " (the one after it may be applied in part: the coordinates of "
                                   "this solver are not a result)"
                                 : " exchanges)") +
                            " and every peer has been told to stop");
    }
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
    for (int64_t k = 0; k < iters; ++k) {
        hipEvent_t *ev = timing_slot(s);
        if (ev) BB_HIP_CHECK(hipEventRecord(ev[0], s->stream));
        BB_TRY(launch_grad(s));
        if (ev) BB_HIP_CHECK(hipEventRecord(ev[1], s->stream));
        s->peer_seq++;
        if (s->peer_fused && s->red_slices > 0) {
            // one launch: reduce, push, wait, sum, update (reduce_exchange_kernel)
            BB_TRY(launch_exchange(s, lr, s->d_stress_hist + s->hist_n));
            if (ev) BB_HIP_CHECK(hipEventRecord(ev[2], s->stream));
            s->hist_n++;
            continue;
        }
        BB_TRY(launch_reduce(s, kReducePeer, 0.0, nullptr));
        if (ev) BB_HIP_CHECK(hipEventRecord(ev[2], s->stream));
        BB_TRY(launch_peer_receive(s, s->d_X, lr, s->momentum, s->d_stress_hist + s->hist_n));
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
    hipError_t e = widen(s, s->d_exch, tmp, n);
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
    if (e == hipSuccess) e = narrow(s, tmp, s->d_exch, n);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    hipFree(tmp);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_solver_write_exchange: ") + hipGetErrorString(e));
    return BB_OK;
}

int bb_solver_matvec_sq(bb_solver *s, const double *x, double *y) {
    BB_REQUIRE(s != nullptr && x != nullptr && y != nullptr, "bb_solver_matvec_sq: NULL argument");
    if (!s->have_wish) return bb::fail(BB_ERR_STATE, "bb_solver_matvec_sq: no wish distances set");
    if (s->n_maps > 1)
        return bb::fail(BB_ERR_STATE, "bb_solver_matvec_sq: not for a solver of several maps");
    if (s->grad_pending)
        return bb::fail(BB_ERR_STATE, "bb_solver_matvec_sq: a bb_solver_grad is pending");
    BB_TRY(bb::enter_device(s->device));
    const int64_t n3 = s->L.n_pad * 3, es = bb::elem_size(s->dtype);
    // (a spectral start calls this ~40 times: the buffer is allocated once and kept)
    if (!s->d_mv_in) BB_TRY(dev_alloc((char **)&s->d_mv_in, n3 * es));
    void *d_in = s->d_mv_in;
    hipError_t e = hipMemsetAsync(s->d_f64_tmp, 0, (size_t)n3 * 8, s->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(s->d_f64_tmp, x, (size_t)s->L.n_bins * 24, hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) e = narrow(s, s->d_f64_tmp, d_in, n3);
    int rc = BB_OK;
    if (e != hipSuccess) rc = bb::fail(BB_ERR_HIP, std::string("bb_solver_matvec_sq: ") + hipGetErrorString(e));
    if (rc == BB_OK) rc = launch_grad(s, kOpMatvec2, d_in);
    if (rc == BB_OK) rc = launch_reduce(s, kReduceExchange, 0.0, nullptr, kScalePlainSum);
    if (rc == BB_OK) {
        e = widen(s, s->d_exch, s->d_f64_tmp, n3);
        if (e == hipSuccess)
            e = hipMemcpyAsync(y, s->d_f64_tmp, (size_t)s->L.n_bins * 24, hipMemcpyDeviceToHost,
                               s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        if (e != hipSuccess)
            rc = bb::fail(BB_ERR_HIP, std::string("bb_solver_matvec_sq: ") + hipGetErrorString(e));
    }
    if (hipStreamSynchronize(s->stream) != hipSuccess && rc == BB_OK)
        rc = bb::fail(BB_ERR_HIP, "bb_solver_matvec_sq: the stream did not drain");
    return rc;
}

int bb_solver_stress(bb_solver *s, double *stress) {
    BB_TRY(check_ready(s, "bb_solver_stress"));
    BB_REQUIRE(stress != nullptr, "bb_solver_stress: stress is NULL");
    BB_TRY(bb::enter_device(s->device));
    BB_TRY(launch_grad(s));
    if (s->n_maps > 1) {                       // the sum over the maps
        std::vector<double> per((size_t)s->n_maps);
        BB_TRY(launch_reduce(s, kReduceStressOnly, 0.0, s->d_map_scalar));
        BB_HIP_CHECK(hipStreamSynchronize(s->stream));
        BB_HIP_CHECK(hipMemcpy(per.data(), s->d_map_scalar, per.size() * sizeof(double),
                               hipMemcpyDeviceToHost));
        *stress = 0.0;
        for (double v : per) *stress += v;
        return BB_OK;
    }
    BB_TRY(launch_reduce(s, kReduceStressOnly, 0.0, s->d_stress_scalar));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    BB_HIP_CHECK(hipMemcpy(stress, s->d_stress_scalar, sizeof(double), hipMemcpyDeviceToHost));
    return BB_OK;
}

int bb_solver_stress_maps(bb_solver *s, double *stress, int n_maps) {
    BB_TRY(check_ready(s, "bb_solver_stress_maps"));
    BB_REQUIRE(stress != nullptr && n_maps == s->n_maps, "bb_solver_stress_maps: bad argument");
    BB_TRY(bb::enter_device(s->device));
    BB_TRY(launch_grad(s));
    double *out = s->n_maps > 1 ? s->d_map_scalar : s->d_stress_scalar;
    BB_TRY(launch_reduce(s, kReduceStressOnly, 0.0, out));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    BB_HIP_CHECK(hipMemcpy(stress, out, (size_t)s->n_maps * sizeof(double), hipMemcpyDeviceToHost));
    return BB_OK;
}

int bb_solver_get_stress_history(bb_solver *s, double *out, int64_t cap, int64_t *n) {
    BB_REQUIRE(s != nullptr && n != nullptr, "bb_solver_get_stress_history: NULL argument");
    BB_TRY(bb::enter_device(s->device));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    // (several maps: iteration-major, n_maps values per iteration)
    *n = s->hist_n * s->n_maps;
    const int64_t m = std::min(cap, s->hist_n * s->n_maps);
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

int bb_solver_sync_timeout(bb_solver *s, int64_t milliseconds) {
    BB_REQUIRE(s != nullptr, "bb_solver_sync_timeout: solver is NULL");
    BB_REQUIRE(milliseconds >= 0, "bb_solver_sync_timeout: milliseconds < 0");
    BB_TRY(bb::enter_device(s->device));
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        const hipError_t e = hipStreamQuery(s->stream);
        if (e == hipSuccess) return BB_OK;
        if (e != hipErrorNotReady)
            return bb::fail(BB_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(e));
        (void)hipGetLastError();   // "not ready" is an answer, not a failure
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        const int64_t ms = (t.tv_sec - t0.tv_sec) * 1000 + (t.tv_nsec - t0.tv_nsec) / 1000000;
        if (ms >= milliseconds)
            return bb::fail(BB_ERR_STATE, "bb_solver_sync_timeout: the stream did not drain within " +
                                              std::to_string(milliseconds) + " ms");
        usleep(50);
    }
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

int bb_solver_measure_event_gap(bb_solver *s, int pairs, double *ms_avg) {
    BB_REQUIRE(s != nullptr && ms_avg != nullptr, "bb_solver_measure_event_gap: NULL argument");
    BB_REQUIRE(pairs >= 1 && pairs <= 256, "bb_solver_measure_event_gap: bad pairs");
    BB_TRY(check_ready(s, "bb_solver_measure_event_gap"));
    BB_TRY(bb::enter_device(s->device));
    std::vector<hipEvent_t> ev((size_t)2 * pairs, nullptr);
    hipError_t e = hipSuccess;
    for (auto &x : ev)
        if (e == hipSuccess) e = hipEventCreate(&x);
    for (int k = 0; k < pairs && e == hipSuccess; ++k) {
        // behind a sweep launch, as the timed intervals are (the sweep only writes partials)
        if (!s->row_owner && launch_grad(s) != BB_OK) e = hipErrorUnknown;
        if (e == hipSuccess) e = hipEventRecord(ev[(size_t)2 * k], s->stream);
        if (e == hipSuccess) e = hipEventRecord(ev[(size_t)2 * k + 1], s->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    double t = 0.0;
    for (int k = 0; k < pairs && e == hipSuccess; ++k) {
        float a = 0.f;
        e = hipEventElapsedTime(&a, ev[(size_t)2 * k], ev[(size_t)2 * k + 1]);
        t += a;
    }
    for (auto x : ev)
        if (x) (void)hipEventDestroy(x);
    if (e != hipSuccess)
        return bb::fail(BB_ERR_HIP, std::string("bb_solver_measure_event_gap: ") + hipGetErrorString(e));
    *ms_avg = t / pairs;
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
        return bb::launch(s->nontemporal ? stream_read_kernel<true> : stream_read_kernel<false>,
                          dim3(s->n_waves / 4), dim3(256), 0, s->stream,
                          (const float4 *)s->d_units, s->chunk_q, s->chunk_r,
                          (float *)s->d_f64_tmp);
    };
    BB_HIP_CHECK(launch());  // warm-up
    BB_HIP_CHECK(hipEventRecord(e0, s->stream));
    for (int k = 0; k < launches; ++k) BB_HIP_CHECK(launch());
    BB_HIP_CHECK(hipEventRecord(e1, s->stream));
    BB_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    BB_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *ms_avg = ms / launches;
    return BB_OK;
}

#if defined(BB_WAVE_TRACE) || defined(BB_UNIT_TRACE)
// Diagnostic build only: `times` sweep launches back to back with nothing in between
// (is a launch's slow first unit a cold instruction cache?).
BB_API int bb_solver_debug_grad_repeat(bb_solver *s, int times) {
    BB_TRY(check_ready(s, "bb_solver_debug_grad_repeat"));
    BB_TRY(bb::enter_device(s->device));
    for (int k = 0; k < times; ++k) BB_TRY(launch_grad(s));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    return BB_OK;
}
// Diagnostic build only (-DBB_OVERLAP_PROBE rides on the trace builds): an UPPER BOUND of what
// overlapping the reduce of iteration k with sweep k + 1 could save (lever (a)).
// The two run as a software pipeline on two streams -- sweep k waits for reduce k - 1 (one
// iteration stale: the coordinates it reads are not the ones the solver would use, timing
// only), reduce k waits for sweep k -- so every iteration has the two cross-queue
// dependencies the real thing would have and NONE of its in-kernel hand-off.  If this is not
// faster than the plain loop, nothing built on it can be.
BB_API int bb_solver_debug_overlap_bound(bb_solver *s, int iters, double lr, double *us_plain,
                                         double *us_pipelined) {
    BB_TRY(check_ready(s, "bb_solver_debug_overlap_bound"));
    BB_REQUIRE(us_plain != nullptr && us_pipelined != nullptr && iters >= 2 && iters <= 100000,
               "bb_solver_debug_overlap_bound: bad argument");
    BB_REQUIRE(s->world == 1 && !s->row_owner, "bb_solver_debug_overlap_bound: one rank, sweep path");
    BB_TRY(bb::enter_device(s->device));
    hipStream_t main_st = s->stream, side = nullptr;
    BB_HIP_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t e_sweep[2], e_red[2];
    for (int q = 0; q < 2; ++q) {
        BB_HIP_CHECK(hipEventCreateWithFlags(&e_sweep[q], hipEventDisableTiming));
        BB_HIP_CHECK(hipEventCreateWithFlags(&e_red[q], hipEventDisableTiming));
    }
    auto now = []() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; };
    int rc = BB_OK;
    for (int phase = 0; phase < 2 && rc == BB_OK; ++phase) {
        for (int pass = 0; pass < 2 && rc == BB_OK; ++pass) {        // pass 0 warms up
            const int n = pass == 0 ? std::max(20, iters / 4) : iters;
            const double t0 = now();
            for (int k = 0; k < n && rc == BB_OK; ++k) {
                if (phase == 0) {
                    rc = launch_grad(s);
                    if (rc == BB_OK) rc = launch_reduce(s, kReduceApply, lr, s->d_stress_scalar);
                } else {
                    if (k > 0) (void)hipStreamWaitEvent(main_st, e_red[(k - 1) & 1], 0);
                    rc = launch_grad(s);
                    (void)hipEventRecord(e_sweep[k & 1], main_st);
                    (void)hipStreamWaitEvent(side, e_sweep[k & 1], 0);
                    s->stream = side;
                    if (rc == BB_OK) rc = launch_reduce(s, kReduceApply, lr, s->d_stress_scalar);
                    s->stream = main_st;
                    (void)hipEventRecord(e_red[k & 1], side);
                }
            }
            (void)hipStreamSynchronize(side);
            (void)hipStreamSynchronize(main_st);
            if (pass == 1) *(phase == 0 ? us_plain : us_pipelined) = (now() - t0) / n;
        }
    }
    for (int q = 0; q < 2; ++q) { hipEventDestroy(e_sweep[q]); hipEventDestroy(e_red[q]); }
    hipStreamDestroy(side);
    return rc;
}
// Diagnostic build only: the stamps of the last stress_grad_kernel launch,
// 8 x uint64 per wave {start, first unit done, last unit consumed, end, xcc<<32 | hw_id,
// first load landed, window landed, coordinates landed}.
BB_API int bb_solver_debug_wave_trace(bb_solver *s, unsigned long long *out, int64_t cap,
                                      int64_t *n_waves) {
    BB_REQUIRE(s != nullptr && n_waves != nullptr, "bb_solver_debug_wave_trace: NULL argument");
    BB_TRY(bb::enter_device(s->device));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    *n_waves = s->n_waves;
    if (out && cap >= 8 * (int64_t)s->n_waves)
        BB_HIP_CHECK(hipMemcpy(out, s->d_stresspart + s->n_waves, (size_t)s->n_waves * 64,
                               hipMemcpyDeviceToHost));
    return BB_OK;
}
// -DBB_UNIT_TRACE: kUnitTraceSlots stamps per wave of the last sweep launch: slot k = the top
// of the wave's unit k (10-ns ticks), slot n_units = the end of its last unit, 0 = unused.
BB_API int bb_solver_debug_unit_trace(bb_solver *s, unsigned long long *out, int64_t cap,
                                      int64_t *n_waves, int64_t *slots) {
    BB_REQUIRE(s != nullptr && n_waves != nullptr && slots != nullptr,
               "bb_solver_debug_unit_trace: NULL argument");
    BB_TRY(bb::enter_device(s->device));
    BB_HIP_CHECK(hipStreamSynchronize(s->stream));
    *n_waves = s->n_waves;
    *slots = abl::kUnitTrace ? abl::kUnitTraceSlots : 0;
    if (out && abl::kUnitTrace && cap >= *slots * (int64_t)s->n_waves)
        BB_HIP_CHECK(hipMemcpy(out, s->d_stresspart + (int64_t)s->n_waves * 9,
                               (size_t)s->n_waves * abl::kUnitTraceSlots * 8, hipMemcpyDeviceToHost));
    return BB_OK;
}
#endif

int bb_solver_iteration_path(const bb_solver *s, int *row_owner, int *waves_per_row) {
    BB_REQUIRE(s != nullptr, "bb_solver_iteration_path: solver is NULL");
    if (row_owner) *row_owner = s->row_owner ? 1 : 0;
    if (waves_per_row) *waves_per_row = s->row_owner ? s->ro_wpr : 0;
    return BB_OK;
}

int bb_solver_traffic(const bb_solver *s, int64_t *unit_bytes, int64_t *pairs_dense) {
    BB_REQUIRE(s != nullptr, "bb_solver_traffic: solver is NULL");
    if (unit_bytes) *unit_bytes = s->n_local * bb::kUnitBytes;
    if (pairs_dense) *pairs_dense = s->L.n_bins * (s->L.n_bins - 1) / 2;
    return BB_OK;
}

}  // extern "C"

// ---- spectral start, device resident ---------------------------------------------
namespace {

// small dense helpers on the host (3 x 3, row-major)
void jacobi3(double *a, double *z) {   // a symmetric 3 x 3 -> eigenvalues on its diagonal, vectors in z
    for (int q = 0; q < 9; ++q) z[q] = (q % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 50; ++sweep) {
        const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
        if (off < 1e-300) break;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = a[p * 3 + q];
                if (apq == 0.0) continue;
                const double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = a[k * 3 + p], akq = a[k * 3 + q];
                    a[k * 3 + p] = c * akp - sn * akq;
                    a[k * 3 + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = a[p * 3 + k], aqk = a[q * 3 + k];
                    a[p * 3 + k] = c * apk - sn * aqk;
                    a[q * 3 + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double zkp = z[k * 3 + p], zkq = z[k * 3 + q];
                    z[k * 3 + p] = c * zkp - sn * zkq;
                    z[k * 3 + q] = sn * zkp + c * zkq;
                }
            }
    }
}

template <typename T>
int spectral_init_t(bb_solver *s, int n_iter, const double *v0, double tol, int *iters_done,
                    double *residual_out) {
    const int64_t n = s->L.n_bins, n_pad = s->L.n_pad, n3 = n_pad * 3;
    if (!s->d_mv_in) BB_TRY(dev_alloc((char **)&s->d_mv_in, n3 * (int64_t)sizeof(T)));
    // V, Z: (n_pad,3) doubles; dS: 12 sums (the two host steps at the end); dP0 / dP1: the
    // per-workgroup partial sums a pass leaves for the next one (two buffers: a pass reads its
    // producer's while it writes its own); dF: rank-loss flag
    bb::DevBuf bV, bZ, bS, bP, bF;
    if (bV.alloc((size_t)n3 * 8) != hipSuccess || bZ.alloc((size_t)n3 * 8) != hipSuccess ||
        bS.alloc(24 * 8) != hipSuccess || bP.alloc((size_t)2 * kSpMaxGroups * 12 * 8) != hipSuccess ||
        bF.alloc(sizeof(int)) != hipSuccess)
        return bb::fail(BB_ERR_NOMEM, "bb_solver_spectral_init: out of device memory");
    double *dV = (double *)bV.p, *dZ = (double *)bZ.p, *dS = (double *)bS.p;
    double *dP0 = (double *)bP.p, *dP1 = dP0 + kSpMaxGroups * 12;
    int *dF = (int *)bF.p;
    hipStream_t st = s->stream;
    const dim3 gvec((unsigned)((n_pad + kSpWG - 1) / kSpWG)), bwg(kSpWG);
    const int groups = (int)std::min<int64_t>(kSpMaxGroups, (n_pad + kSpWG - 1) / kSpWG);
    const dim3 ggrp((unsigned)groups);
    const Affine3 ident = {{0, 0, 0}, {1, 0, 0, 0, 1, 0, 0, 0, 1}, 1.0};
    double sums[12];
    auto fetch_sums = [&]() -> int {
        BB_HIP_CHECK(hipStreamSynchronize(st));
        BB_HIP_CHECK(hipMemcpy(sums, dS, sizeof(sums), hipMemcpyDeviceToHost));
        return BB_OK;
    };
    // dZ (any basis of the subspace, its 12 sums in dP1) -> dV orthonormal by Cholesky-QR, twice
    // (the second pass removes what the first leaves at cond(Z)^2 * eps), and the sweep's
    // right-hand sides d_mv_in = (T)(V - mean V): two kernels, the 3 x 3 steps in their prologues
    auto orthonormalise = [&]() -> int {
        BB_HIP_CHECK(bb::launch(sp_affine_stats_kernel<double>, ggrp, bwg, 0, st, (const double *)dZ, dV,
                                n, n_pad, (const double *)dP1, groups, (int)kSpChol, 1.0, dF, dP0));
        BB_HIP_CHECK(bb::launch(sp_affine_centre_kernel<T>, gvec, bwg, 0, st, (const double *)dV, dV,
                                (T *)s->d_mv_in, n, n_pad, (const double *)dP0, groups, dF));
        return BB_OK;
    };
    // dZ = -1/2 J (D o D) J V  (J = I - 11'/n) from the centred V in d_mv_in: sweep, sum over
    // the ranks, centre; the 12 sums of dZ are left in dP1
    auto apply_B = [&]() -> int {
        BB_TRY(launch_grad(s, kOpMatvec2, s->d_mv_in));
        BB_TRY(exchange_sum(s));
        BB_HIP_CHECK(bb::launch(sp_stats_kernel<T>, ggrp, bwg, 0, st, (const T *)s->d_exch, n, n_pad, dP0));
        BB_HIP_CHECK(bb::launch(sp_affine_stats_kernel<T>, ggrp, bwg, 0, st, (const T *)s->d_exch, dZ, n,
                                n_pad, (const double *)dP0, groups, (int)kSpMean, -0.5, dF, dP1));
        return BB_OK;
    };
    BB_HIP_CHECK(hipMemsetAsync(dF, 0, sizeof(int), st));
    BB_HIP_CHECK(hipMemsetAsync(s->d_V, 0, (size_t)n3 * sizeof(T), st));   // exchange_sum reads it
    BB_HIP_CHECK(hipMemsetAsync(dZ, 0, (size_t)n3 * 8, st));
    BB_HIP_CHECK(hipMemcpyAsync(dZ, v0, (size_t)n * 24, hipMemcpyHostToDevice, st));
    BB_HIP_CHECK(bb::launch(sp_stats_kernel<double>, ggrp, bwg, 0, st, (const double *)dZ, n, n_pad, dP1));
    BB_TRY(orthonormalise());
    // tol > 0: after every product but the first (a random V cannot have converged), the
    // relative distance of Z = B V from span(V), ||Z - V (V^T Z)||_F / ||Z||_F -- from the
    // Gram matrix V^T Z and the sums of Z^T Z the product left in dP1: V is orthonormal, so
    // the squared distance is ||Z||_F^2 - ||V^T Z||_F^2 (the difference of two sums resolves
    // a ratio down to about 1e-7).  Below tol the loop ends; the product just made is the
    // Rayleigh-Ritz step's.  Costs two small kernels and a read of 24 doubles per product.  Every
    // rank holds the same V and Z bit for bit (the summed products are identical on all
    // ranks), so all ranks leave at the same product.
    double zsums[12];
    double residual = -1.0;
    int done = 0;
    bool have_product = false;
    for (int it = 0; it < n_iter; ++it) {
        BB_TRY(apply_B());
        if (tol > 0.0 && it > 0) {
            BB_HIP_CHECK(bb::launch(gram3_kernel<double, double>, dim3(1), dim3(1024), 0, st,
                                    (const double *)dV, (const double *)dZ, n, dS));
            BB_HIP_CHECK(bb::launch(sp_fold_partials_kernel, dim3(1), dim3(64), 0, st,
                                    (const double *)dP1, groups, dS + 12));
            BB_TRY(fetch_sums());
            BB_HIP_CHECK(hipMemcpy(zsums, dS + 12, sizeof(zsums), hipMemcpyDeviceToHost));
            double gg = 0.0;
            const double zz = zsums[0] + zsums[4] + zsums[8];
            for (int q = 0; q < 9; ++q) gg += sums[q] * sums[q];
            residual = zz > 0.0 ? sqrt(std::max(0.0, zz - gg) / zz) : 0.0;
            if (residual < tol) { have_product = true; break; }
        }
        BB_TRY(orthonormalise());
        done = it + 1;
    }
    if (iters_done) *iters_done = done;
    if (residual_out) *residual_out = residual;
    // Rayleigh-Ritz on span(V): M = sym(V^T B V), X0 = V E sqrt(max(lambda, 0)).  From here on
    // the host takes part: three reads of 12 doubles for the whole start.
    if (!have_product) {
        BB_TRY(apply_B());
        BB_HIP_CHECK(bb::launch(gram3_kernel<double, double>, dim3(1), dim3(1024), 0, st,
                                (const double *)dV, (const double *)dZ, n, dS));
        BB_TRY(fetch_sums());
    }
    int lost = 0;
    BB_HIP_CHECK(hipMemcpy(&lost, dF, sizeof(int), hipMemcpyDeviceToHost));
    if (lost)
        return bb::fail(BB_ERR_STATE, "bb_solver_spectral_init: the iterate lost rank "
                                      "(fewer than 3 independent directions in the map)");
    double m[9], z[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m[i * 3 + j] = 0.5 * (sums[i * 3 + j] + sums[j * 3 + i]);
    jacobi3(m, z);
    int order[3] = {0, 1, 2};
    std::sort(order, order + 3, [&](int a, int b) { return m[a * 4] > m[b * 4]; });
    // An eigenvector's sign is arbitrary: every Ritz vector V z_c is turned to the side of
    // the start's first column p (p . V z_c >= 0), as solver.py's spectral_init does, so one
    // seed gives one start whichever of the two computed it.
    BB_HIP_CHECK(hipMemsetAsync(dZ, 0, (size_t)n3 * 8, st));
    BB_HIP_CHECK(hipMemcpyAsync(dZ, v0, (size_t)n * 24, hipMemcpyHostToDevice, st));
    BB_HIP_CHECK(bb::launch(gram3_kernel<double, double>, dim3(1), dim3(1024), 0, st, (const double *)dV,
                            (const double *)dZ, n, dS));
    BB_TRY(fetch_sums());                                  // sums[r * 3 + 0] = V[:, r] . p
    Affine3 f = ident;
    for (int c = 0; c < 3; ++c) {
        const double lam = m[order[c] * 4] > 0.0 ? m[order[c] * 4] : 0.0;
        double side = 0.0;
        for (int r = 0; r < 3; ++r) side += z[r * 3 + order[c]] * sums[r * 3];
        const double sg = side < 0.0 ? -1.0 : 1.0;
        for (int r = 0; r < 3; ++r) f.m[r * 3 + c] = sg * z[r * 3 + order[c]] * sqrt(lam);
    }
    BB_HIP_CHECK(bb::launch(affine3_kernel<double, T>, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0,
                            st, (const double *)dV, (T *)s->d_X, n, n_pad, f));
    BB_HIP_CHECK(hipMemsetAsync(s->d_V, 0, (size_t)n3 * sizeof(T), st));
    BB_HIP_CHECK(hipStreamSynchronize(st));
    return BB_OK;
}

}  // namespace

extern "C" int bb_solver_spectral_init_tol(bb_solver *s, int n_iter, double tol, const double *v0,
                                           int *iters_done, double *residual);
extern "C" int bb_solver_spectral_init(bb_solver *s, int n_iter, const double *v0) {
    return bb_solver_spectral_init_tol(s, n_iter, 0.0, v0, nullptr, nullptr);
}
extern "C" int bb_solver_spectral_init_tol(bb_solver *s, int n_iter, double tol, const double *v0,
                                           int *iters_done, double *residual) {
    BB_REQUIRE(s != nullptr && v0 != nullptr, "bb_solver_spectral_init: NULL argument");
    BB_REQUIRE(n_iter >= 0 && n_iter <= 100000, "bb_solver_spectral_init: bad n_iter");
    BB_REQUIRE(tol >= 0.0 && tol < 1.0, "bb_solver_spectral_init: need 0 <= tol < 1");
    if (!s->have_wish) return bb::fail(BB_ERR_STATE, "bb_solver_spectral_init: no wish distances set");
    if (s->world != 1 && !s->peer_connected && !s->comm)
        return bb::fail(BB_ERR_STATE, "bb_solver_spectral_init: with world > 1 the ranks need their "
                                      "exchange first (bb_solver_peer_connect, or bb_solver_comm_init "
                                      "/ _attach); without one the caller sums bb_solver_matvec_sq "
                                      "over the ranks");
    if (s->grad_pending)
        return bb::fail(BB_ERR_STATE, "bb_solver_spectral_init: a bb_solver_grad is pending");
    if (s->n_maps > 1)
        return bb::fail(BB_ERR_STATE, "bb_solver_spectral_init: not for a solver of several maps "
                                      "(start each map with a solver of its own)");
    BB_REQUIRE(s->L.n_bins >= 3, "bb_solver_spectral_init: needs at least 3 bins");
    BB_TRY(bb::enter_device(s->device));
    const int rc = s->dtype == BB_F32 ? spectral_init_t<float>(s, n_iter, v0, tol, iters_done, residual)
                                      : spectral_init_t<double>(s, n_iter, v0, tol, iters_done, residual);
    if (rc != BB_OK) return rc;
    s->have_coords = true;
    s->hist_n = 0;
    s->grad_pending = false;
    return BB_OK;
}
