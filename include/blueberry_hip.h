/*
 * blueberry_hip.h -- C-ABI of libblueberry_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for the contact-matrix -> 3D-coordinates hot path of
 * jmschrei/blueberry.  Plain C: pointers, sizes, int status codes.  No torch,
 * numpy or C++ types cross this line.  The Python host (blueberry_amd/) binds
 * it with ctypes; INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * The reference has NO FFI / plugin interface: its "interface"
 * is Python names star-exported from blueberry/__init__.py:38-43 and Cython
 * `cpdef` functions taking numpy arrays.  Every entry point below therefore
 * cites the reference *function or loop* it replaces; entry points of the
 * 3D-structure solver cite docs/SPEC.md instead, because the reference
 * contains no solver.
 *
 * Conventions
 *   - every function returns BB_OK (0) or a BB_ERR_* code; bb_last_error()
 *     returns a thread-local message for the last failure on this thread;
 *   - host pointers are borrowed for the duration of the call only; device
 *     memory is owned by opaque handles with explicit create/destroy;
 *   - one handle = one device = one host thread at a time; the library never
 *     calls the oracle or any CPU fallback: without a usable GPU every compute
 *     entry point fails with BB_ERR_HIP.
 */
#ifndef BLUEBERRY_HIP_H
#define BLUEBERRY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BB_API __attribute__((visibility("default")))

#define BB_VERSION 100 /* 0.1.0 */

enum {
    BB_OK = 0,
    BB_ERR_INVALID = 1, /* bad argument (null pointer, negative size, ...)   */
    BB_ERR_HIP = 2,     /* a HIP runtime call failed / no usable device      */
    BB_ERR_STATE = 3,   /* call sequence error (e.g. iterate before set_wish) */
    BB_ERR_NOMEM = 4    /* host or device allocation failed                  */
};

enum { BB_F32 = 0, BB_F64 = 1 };              /* arithmetic type of the solver */
enum { BB_KIND_WISH = 0, BB_KIND_COUNTS = 1 }; /* meaning of a dense input      */

BB_API int bb_version(void);
BB_API const char *bb_last_error(void);
/* Number of visible HIP devices; BB_ERR_HIP (count = 0) when there is none. */
BB_API int bb_device_count(int *count);

/* ---------------------------------------------------------------------- */
/* K1  count_band_regions   (reference: blueberry/blueberry.pyx:77-91)      */
/* ---------------------------------------------------------------------- */

/* t = #{(i, j) : 0 <= j < i < n, low <= regions[i] - regions[j] <= high}.
 * `regions` is a host vector of n float64 (what the reference reads through
 * `<double*> regions_ndarray.data`, pyx:80); low/high are C ints as at pyx:82
 * (LOW_FITHIC_CUTOFF / HIGH_FITHIC_CUTOFF, blueberry/utils.py:25-26).  The
 * count is exact (fp64 subtract, integer sum). */
BB_API int bb_band_count(const double *regions, int64_t n, int32_t low, int32_t high,
                         int device, int64_t *count);

/* The same sum restricted to rows i in [i_begin, i_end): one rank's share of
 * a row-sharded count (the shares are summed with an integer all-reduce). */
BB_API int bb_band_count_rows(const double *regions, int64_t n, int32_t low, int32_t high,
                              int64_t i_begin, int64_t i_end, int device, int64_t *count);

/* ---------------------------------------------------------------------- */
/* Device layout of the wish-distance matrix (docs/SPEC.md 3).  Host-only   */
/* arithmetic: callable without a GPU.                                      */
/* ---------------------------------------------------------------------- */

typedef struct bb_layout_info {
    int64_t n_bins;         /* N                                              */
    int64_t n_pad;          /* N rounded up to a multiple of vw               */
    int64_t vw;             /* tile edge = strip width: 512; fp64 <= 4096 bins: 128 */
    int64_t rows_per_unit;  /* rows per 8-KiB unit: 4 (fp32); fp64: 2, or 8 when vw = 128 */
    int64_t units_per_tile; /* vw / rows_per_unit                             */
    int64_t n_blocks;       /* n_pad / vw                                     */
    int64_t n_tiles;        /* upper-triangular tiles incl. the diagonal      */
    int64_t n_units;        /* n_tiles * units_per_tile                       */
} bb_layout_info;

BB_API int bb_layout_dense_info(int64_t n_bins, int dtype, bb_layout_info *info);
/* Tile list of the dense upper triangle in device order (column-strip major:
 * J ascending, then I ascending, I <= J).  cap >= info.n_tiles. */
BB_API int bb_layout_dense_tiles(int64_t n_bins, int dtype, int32_t *tile_I, int32_t *tile_J,
                                 int64_t cap);
/* Contiguous share of n_units units owned by `rank` of `world`. */
BB_API int bb_layout_rank_units(int64_t n_units, int rank, int world, int64_t *u_begin,
                                int64_t *u_end);

/* ---------------------------------------------------------------------- */
/* S0  3D-structure solver  (docs/SPEC.md; absent from the reference)       */
/* ---------------------------------------------------------------------- */

typedef struct bb_solver bb_solver;

/* Create a solver for n_bins bins on `device`, playing `rank` of `world`
 * ranks (world = 1: the whole matrix).  tile_I/tile_J (n_tiles entries, device
 * order, I <= J) select the tiles that exist -- pairs outside them carry no
 * constraint (blocked-sparse input); pass NULL / 0 for the dense upper
 * triangle.  The solver runs on its own HIP stream unless bb_solver_set_stream
 * replaces it. */
BB_API int bb_solver_create(bb_solver **out, int64_t n_bins, int dtype, int device, int rank,
                            int world, const int32_t *tile_I, const int32_t *tile_J,
                            int64_t n_tiles);
BB_API int bb_solver_destroy(bb_solver *s);

/* Run on the caller's stream (a hipStream_t; NULL = the device's null stream),
 * e.g. torch's current stream so that an external all-reduce is ordered with
 * the kernels. */
BB_API int bb_solver_set_stream(bb_solver *s, void *hip_stream);
BB_API int bb_solver_layout(const bb_solver *s, bb_layout_info *info, int64_t *u_begin,
                            int64_t *u_end);

/* Dense (n_bins, n_bins) float64 host matrix, leading dimension ld elements;
 * only elements [i][j] with i < j are read.  kind = BB_KIND_WISH: entries are
 * wish distances (<= 0 / non-finite = no constraint).  kind = BB_KIND_COUNTS:
 * entries are contact counts, converted on the device with
 * delta = c^(-1/alpha) (SPEC 2.1).  This is the matrix a ContactMap holds
 * (reference: blueberry/datatypes.pyx:78-86 `matrix`, float64, C-contiguous).
 * Each rank uploads only its own units. */
BB_API int bb_solver_set_wish_dense(bb_solver *s, const double *host, int64_t ld, int kind,
                                    double alpha);
/* SEVERAL MAPS IN ONE SOLVER.  The reference's ContactMap is per chromosome
 * (blueberry/datatypes.pyx:88), so the common job is 23 maps of 1,000-5,000 bins, each of
 * which alone is launch-bound (5-20 us per iteration whatever its size).  Laid end to end --
 * map m owns the bins [bin_begin[m], bin_begin[m+1]), every map starting at a multiple of the
 * tile edge, the solver created with the tile list of the maps' own dense triangles (no tile
 * joins two maps) -- they are ONE blocked-sparse problem: one sweep and one reduce per
 * iteration for all of them.
 *   bb_solver_set_maps            declare the maps (world = 1); map m steps with
 *                                 lr * lr_scale[m] (bb_solver_iterate's lr times it: pass
 *                                 lr = 1 and lr_scale[m] = 1 / (2 n_m) for the SMACOF step),
 *                                 and the stress is kept PER MAP: bb_solver_get_stress_history
 *                                 then returns n_maps values per iteration (iteration-major)
 *   bb_solver_set_wish_dense_block  map m's (n_sub, n_sub) host matrix into its bins
 *                                 [bin_offset, bin_offset + n_sub); as bb_solver_set_wish_dense.
 *                                 A map ends where its last tile ends: the pairs of the tiles
 *                                 the block touches that lie outside the block are cleared
 *                                 (no constraint); other tiles keep what they hold
 *   bb_solver_stress_maps         per-map stress of the current coordinates
 * Coordinates travel as one (n_bins, 3) array (rows between the maps are padding: 0). */
BB_API int bb_solver_set_maps(bb_solver *s, int n_maps, const int64_t *bin_begin,
                              const double *lr_scale);
BB_API int bb_solver_set_wish_dense_block(bb_solver *s, const double *host, int64_t ld,
                                          int64_t n_sub, int64_t bin_offset, int kind, double alpha);
BB_API int bb_solver_stress_maps(bb_solver *s, double *stress, int n_maps);
/* A STEP PER BIN: bin i moves by lr * scale[i] * g_i.  SPEC 2.4's step 1 / (2 N) is the
 * Guttman transform of a COMPLETE map; in an incomplete one the bins do not all have N - 1
 * partners -- the whole genome at 10 kb as blocks: 24,926 for a bin of chr1, 4,813 for one of
 * chr21; inside a real chromosome's block most long-range pairs have no contact at all -- and
 * the one step the largest degree allows is many times too short for the rest.  With
 * scale[i] = (D + 1) / (degree[i] + 1), D the largest degree, and lr = 1 / (2 (D + 1)) every
 * bin takes the step 1 / (2 (degree[i] + 1)) its own number of partners allows (SPEC 2.4.1:
 * still a descent step; on a genome-like tile list 1e-3 of the start's stress in a third of
 * the iterations).
 *   bb_solver_degrees          per bin, how many of THIS rank's stored pairs constrain it
 *                              (delta > 0): one pass over the resident units; world > 1: the
 *                              caller sums the counts over the ranks
 *   bb_solver_set_bin_steps    the factors, one per bin (finite, > 0); NULL: one step for all
 *   bb_solver_set_block_steps  the same from one factor per block of the layout
 *                              (bb_solver_layout: n_blocks blocks of vw bins) -- what a tile
 *                              list alone gives, without a pass over the data
 * The gradient of bin i is scaled where it leaves the reduce -- into the update, the
 * exchange buffer (bb_solver_grad then holds scale * g) or the peers' arenas -- so it works
 * on any number of ranks with every exchange; every rank passes the same factors (maps of at
 * most 4,096 bins on one rank keep their one launch per iteration: its kernel scales too).
 * On a solver of several maps the factors REPLACE the per-map steps of bb_solver_set_maps
 * (pass lr = 1 to bb_solver_iterate and scale[i] = the step of bin i, e.g.
 * 1 / (2 (degree[i] + 1)); clearing them there is BB_ERR_STATE).  BB_ERR_STATE while a
 * bb_solver_grad is pending.  No counterpart in the reference (it has no solver;
 * SURVEY.md 0). */
BB_API int bb_solver_degrees(bb_solver *s, int64_t *degree, int64_t n_bins);
BB_API int bb_solver_set_bin_steps(bb_solver *s, const double *scale, int64_t n_bins);
BB_API int bb_solver_set_block_steps(bb_solver *s, const double *scale, int64_t n_blocks);
/* Blocked-sparse input: nnz entries (rows[k], cols[k], vals[k]) of the symmetric
 * matrix, either triangle; an unordered pair that occurs more than once keeps its
 * LAST entry, as in the reference's scatter loop (blueberry/datatypes.pyx:110-116)
 * and in bb_contactmap_scatter.  Every entry must fall in a
 * tile of the solver's tile list; all other pairs carry no constraint.  This
 * is the form of the reference's own input files -- sparse (pos_i, pos_j,
 * count) triples (blueberry/datatypes.pyx:31-38, :100-102) -- without the
 * dense (n_bins+1)^2 host matrix, for genome-wide 10 kb maps (BASELINE config 5).
 * KRnorm / KRexpected (n_bins doubles each, or both NULL): each value is first
 * divided by KRnorm[i] * KRnorm[j] * KRexpected[j - i] (i < j), the element-wise
 * form of ContactMap.normalize (blueberry/datatypes.pyx:166-169), followed by the
 * reference's nan_to_num (pyx:171): a NaN quotient is 0 = "no constraint", an overflowing
 * one the largest double (whose wish distance, 1.8e-103, is a constraint in fp64 and
 * below the wish floor in fp32) -- exactly what the dense ContactMap path gives. */
BB_API int bb_solver_set_wish_sparse(bb_solver *s, const int64_t *rows, const int64_t *cols,
                                     const double *vals, int64_t nnz, int kind, double alpha,
                                     const double *KRnorm, const double *KRexpected);
/* The same input WITHOUT host-side binning: the (n, 3) float64 array [pos_i, pos_j, count] of a
 * Rao-format file exactly as ContactMap.__init__ holds it (blueberry/datatypes.pyx:100-102),
 * copied to the device once.  row_major != 0: C-ordered (n, 3) rows; 0: the reference's
 * column-major array (pyx:111-113 read it that way).  numpy.nan_to_num (pyx:102) and
 * bin = (int)(pos / resolution) (pyx:111-112) are applied by the kernels as they read.
 *   bb_triples_tiles            which tiles of the (n_bins, dtype) layout the triples name:
 *                               present[I * n_blocks + J] = 1, I <= J (n_blocks^2 bytes) -- the
 *                               tile list to create the solver with
 *   bb_solver_set_wish_triples  as bb_solver_set_wish_sparse (last entry of a pair wins,
 *                               KRnorm / KRexpected optional), from the resident triples */
typedef struct bb_triples bb_triples;
BB_API int bb_triples_create(bb_triples **out, const double *triples, int64_t n, int32_t resolution,
                             int32_t row_major, int device);
BB_API int bb_triples_destroy(bb_triples *t);
BB_API int bb_triples_tiles(const bb_triples *t, int64_t n_bins, int dtype, uint8_t *present,
                            int64_t n_blocks);
BB_API int bb_solver_set_wish_triples(bb_solver *s, const bb_triples *t, int kind, double alpha,
                                      const double *KRnorm, const double *KRexpected);
/* Synthetic input generated on the device: delta_ij = |x*_i - x*_j| for the
 * (n_bins,3) float64 host coordinates `xstar`, so that
 * N = 50k needs no 20 GB host matrix. */
BB_API int bb_solver_set_wish_from_coords(bb_solver *s, const double *xstar);

BB_API int bb_solver_set_coords(bb_solver *s, const double *xyz); /* (n_bins,3) f64 host */
BB_API int bb_solver_get_coords(bb_solver *s, double *xyz);

/* Heavy-ball momentum (SPEC 2.4): V <- mu V - lr g, X <- X + V, V = 0 after
 * bb_solver_set_coords.  mu = 0 (the default) is the plain gradient step. */
BB_API int bb_solver_set_momentum(bb_solver *s, double mu);

/* world = 1: `iters` iterations of { stress + gradient, update }, enqueued
 * back to back; stress history is kept on the device (2^20 entries: more iterations than
 * that since the last bb_solver_set_coords are refused with BB_ERR_STATE before anything is
 * enqueued -- read the history out and set the coordinates again to go on). */
BB_API int bb_solver_iterate(bb_solver *s, int64_t iters, double lr);

/* world > 1 (also valid for world = 1): one iteration in two halves around
 * the caller's all-reduce(sum) of the exchange buffer.
 *   bb_solver_grad : this rank's partial gradient and stress -> exchange buffer
 *   bb_solver_apply: X <- X - lr * exchange[0:3*n_pad]; records the stress    */
BB_API int bb_solver_grad(bb_solver *s);
BB_API int bb_solver_apply(bb_solver *s, double lr);
/* The exchange buffer: 3*n_pad + 2 elements of the solver's dtype,
 * [ g (n_pad,3) | stress hi | stress lo ].  By default owned by the solver;
 * a caller that needs to hand it to a collective library may supply its own
 * device allocation of that size. */
BB_API int bb_solver_exchange_size(const bb_solver *s, int64_t *n_elems);
BB_API int bb_solver_get_exchange_buffer(bb_solver *s, void **dev_ptr);
BB_API int bb_solver_set_exchange_buffer(bb_solver *s, void *dev_ptr);
/* Direct RCCL path: the library loads librccl at run time and owns a
 * communicator, so the whole multi-rank iteration is enqueued from C on the
 * solver's own stream -- no framework in the loop.
 *   bb_comm_unique_id      one rank makes the 128-byte id (ncclGetUniqueId) ...
 *   bb_solver_comm_init    ... every rank passes it in (ncclCommInitRank; collective)
 *   bb_solver_allreduce    in-place sum of the exchange buffer, between grad and apply
 *   bb_solver_iterate_dist `iters` x { grad, allreduce, apply }
 * The id travels by whatever the caller has (MPI_Bcast, torch.distributed, a file). */
BB_API int bb_comm_unique_id(void *id_out_128_bytes);
BB_API int bb_solver_comm_init(bb_solver *s, const void *unique_id_128_bytes);
BB_API int bb_solver_allreduce(bb_solver *s);
BB_API int bb_solver_iterate_dist(bb_solver *s, int64_t iters, double lr);
/* Number of ranks RCCL reports for the library's communicator (ncclCommCount): what the
 * bench line quotes as the world size that really took part in the collective. */
BB_API int bb_solver_comm_world(const bb_solver *s, int *world);
/* Give the communicator up (ncclCommAbort): ends a collective that a peer never
 * joined, so the solver's stream drains again.  The solver falls back to
 * bb_solver_grad / caller's all-reduce / bb_solver_apply. */
BB_API int bb_solver_comm_abort(bb_solver *s);
/* The communicator a bb_solver_comm_init makes is kept by the library for the life of the
 * process, one per (device, rank, world); a solver borrows it and hands it back when it is
 * destroyed.  Later solvers of the same job skip ncclCommInitRank:
 *   bb_comm_cached          *available = 1 if a free communicator for this key is held
 *   bb_solver_comm_attach   borrow it (every rank of the job must take the same path: agree
 *                           on `available` first -- the Python wrapper all-reduces a MIN)
 *   bb_solver_comm_detach   hand a borrowed one back without using it
 *   bb_comm_cache_clear     destroy the communicators no solver holds (ncclCommDestroy) */
BB_API int bb_comm_cached(int device, int rank, int world, int *available);
/* The same with the cached communicator's GENERATION: a hash of the unique id its
 * ncclCommInitRank was given -- equal on the ranks that made it together, different for any
 * other communicator of the same (device, rank, world), e.g. one left over from an earlier
 * process group or made while another rank's solver still held the previous one.  The ranks
 * must hold the SAME generation before any of them attaches (the Python wrapper all-gathers
 * it); 0 when nothing is available.  A communicator whose collective failed to enqueue, or
 * that bb_solver_comm_abort gave up, never returns to the cache. */
BB_API int bb_comm_cached_generation(int device, int rank, int world, int *available,
                                     uint64_t *generation);
BB_API int bb_solver_comm_attach(bb_solver *s);
BB_API int bb_solver_comm_detach(bb_solver *s);
BB_API int bb_comm_cache_clear(void);

/* Peer exchange: a one-shot all-reduce over xGMI written into the solver's own
 * kernels -- no collective library in the loop.  Every rank owns a receive
 * arena (uncached device memory, 2 parities x world slots of the exchange
 * vector + one sequence flag per source rank) that the other ranks map through
 * HIP IPC.  The last reduce launch of an iteration stores this rank's partial
 * gradient straight into its slot on every peer and then raises its flag there;
 * the update kernel waits (bounded) for all `world` flags, sums the slots in
 * RANK ORDER -- bit-identical coordinates on every rank -- and applies the step.
 *   bb_solver_peer_export   allocate the arena, write its handle (BB_PEER_HANDLE_BYTES)
 *   bb_solver_peer_connect  map the arenas of all ranks (handles in rank order;
 *                           collective: every rank must have exported)
 *   bb_solver_iterate_peer  `iters` x { grad, reduce+push, wait+sum+apply }
 *   bb_solver_peer_status   0 = healthy; 1 = a wait ran into the time limit
 *                           (BB_PEER_TIMEOUT_MS, default 10000) or a peer reported
 *                           its own failure: this call reports BB_ERR_STATE
 *   bb_solver_peer_set_timeout  change that limit (a short one for a trial run)
 *   bb_solver_peer_form     1 = the one-launch form below, 0 = the two launches above
 * One-launch form (the default with one rank per GPU): the unit of exchange is what one
 * workgroup of the reduce sums anyway, 128 gradient elements of one block.  Each thread that
 * holds a sum stores it into its rank's slot on every peer, reads the same element of every
 * rank's slot in its own arena until all have ARRIVED, adds the `world` partials in rank
 * order, updates its coordinate and marks the words it read as empty again: { grad, exchange }
 * per iteration, no flags and no fences -- an empty slot word holds all-ones (a NaN that no
 * arithmetic produces; a sum that should ever carry those bits is sent as the canonical NaN),
 * a 4- or 8-byte store is seen whole or not at all, and nothing else is published with it.
 * Nobody waits for any workgroup but the owners of the same elements.  Every
 * workgroup pushes before it waits and workgroups start in index order, so the lowest
 * unfinished one can always finish -- if each rank has its GPU to itself.  bb_solver_peer_connect
 * sees from the handles (PCI ids) whether ranks SHARE a GPU (a rehearsal) and then keeps the
 * two-launch form: waiting workgroups of one rank can keep another rank's sweep off the
 * device.  BB_PEER_FUSED=0|1 overrides, identically on every rank.
 * Failure.  Two-launch form: decided ONCE per iteration, by one wave, for the whole update:
 * X is advanced by a complete step or not at all.  One-launch form: every wave decides
 * for its 64 elements, so a rank that dies in the middle of an exchange can leave its
 * peers with part of a step applied; when nothing of an exchange arrives (a rank that is
 * late, stalled or gone) no workgroup applies anything.  Either way the failed rank stops
 * pushing and leaves a poison word on every peer, so their waits fail at once as well -- no
 * rank goes on consuming partials of coordinates that no longer move -- the status is
 * sticky, and the coordinates of a solver whose status is not 0 are not a result.
 * Who sends what.  Rank q's units lie in tiles (I, J) and carry gradient for the bins of
 * blocks I and J only; its partial for every other block is zero by construction.  Every rank
 * derives the same table (block -> ranks that touch it) from the tile list and the partition
 * at bb_solver_peer_connect, and in both forms a rank pushes a block only if it is in the
 * block's set, and nobody waits for, reads or adds the slot of a rank that is not: not one
 * bit of the result changes (the zeros were added as zeros), but a dense map on 8 ranks sends
 * 72 % of the bytes (rank 0 touches the first third of the blocks), the whole genome as
 * blocks 15 % (a rank's share touches the blocks of a few chromosomes).  BB_PEER_MASK=0:
 * every rank sends every block.
 * The handles travel by whatever the caller has (MPI_Allgather, torch.distributed,
 * a file).  All ranks must be on one node with peer access between their GPUs. */
#define BB_PEER_HANDLE_BYTES 128
BB_API int bb_solver_peer_export(bb_solver *s, void *handle_out);
BB_API int bb_solver_peer_connect(bb_solver *s, const void *handles_rank_order);
BB_API int bb_solver_iterate_peer(bb_solver *s, int64_t iters, double lr);
BB_API int bb_solver_peer_status(bb_solver *s, int *status);
BB_API int bb_solver_peer_set_timeout(bb_solver *s, int64_t milliseconds);
BB_API int bb_solver_peer_form(bb_solver *s, int *one_launch);
/* Choose the form after bb_solver_peer_connect and before the first exchange (identically on
 * every rank): 0 = the two launches, 1 = the one launch.  The one-launch form rests on
 * properties of remote stores over xGMI that no one-GPU test can exercise; the Python host
 * therefore keeps the two-launch form for a fit() that was not preceded by a trial against
 * RCCL (BB_COMM=peer without BB_COMM_TRIAL / BB_PEER_FUSED), and takes the one-launch form
 * where a trial has compared its coordinates with RCCL's (bench.py). */
BB_API int bb_solver_peer_set_form(bb_solver *s, int one_launch);

/* Host-staged access to the exchange buffer, widened to float64, for callers
 * whose collective runs on host memory (MPI, gloo): read after bb_solver_grad,
 * sum over ranks, write back, then bb_solver_apply.  n = bb_solver_exchange_size. */
BB_API int bb_solver_read_exchange(bb_solver *s, double *host, int64_t n);
BB_API int bb_solver_write_exchange(bb_solver *s, const double *host, int64_t n);

/* Y = (D o D) X for this rank's resident units: y_i = sum_j delta_ij^2 x_j over
 * the symmetric matrix of squared wish distances (absent pairs count 0), for
 * three right-hand sides at once; x, y are (n_bins,3) float64 on the host.  One
 * sweep of the same kernel and layout as the gradient.  With world > 1 the
 * caller sums y over the ranks.  It is the operator of classical-MDS / spectral
 * initialisation -- the role SURVEY.md 8(f)-2 gives the reference's
 * ContactMap.eigenvector (blueberry/datatypes.pyx:216-235). */
BB_API int bb_solver_matvec_sq(bb_solver *s, const double *x, double *y);

/* Classical-MDS start computed and LEFT on the device: `n_iter` block power iterations on
 * B = -1/2 J (D o D) J over the resident units (one sweep each, as bb_solver_matvec_sq) from
 * the (n_bins,3) host start `v0`, Cholesky-QR between them, a 3 x 3 Rayleigh-Ritz step,
 * X0 = V sqrt(Lambda) written straight into the solver's coordinates (history and velocity
 * reset, as bb_solver_set_coords).  The N x 3 algebra of the loop -- centring, Gram matrices,
 * the 3 x 3 Cholesky factors and the maps they define -- runs in kernels that hand their
 * results to one another in device memory: no host round trip per product, three reads of
 * 12 doubles for the whole start.  world > 1: collective; every rank passes the same v0, the
 * per-rank products are summed over the ranks on the device through the exchange the solver
 * already has (bb_solver_peer_connect, or bb_solver_comm_init / _attach; BB_ERR_STATE without
 * one), and every rank ends with the same coordinates.  BB_ERR_STATE "lost rank" when the
 * iterate has fewer than 3 independent directions (coordinates untouched).  The role
 * SURVEY.md 8(f)-2 gives ContactMap.eigenvector as the solver's initialisation
 * (blueberry/datatypes.pyx:216-235). */
BB_API int bb_solver_spectral_init(bb_solver *s, int n_iter, const double *v0);

/* The same start with a stopping rule: `n_iter` is the most products the loop makes; after
 * every product but the first the relative distance of Z = B V from span(V),
 * ||Z - V (V^T Z)||_F / ||Z||_F, is read back (24 doubles) and the loop ends
 * once it is below `tol` (0 = never: exactly bb_solver_spectral_init; the difference of sums
 * it is formed from resolves it down to about 1e-7, the fp32 sweep to about 1e-6).  A complete
 * noise-free map has rank 3 and ends after two products instead of `n_iter`; a noisy or
 * incomplete one runs until the third and fourth eigenvalue have separated that far.
 * `iters_done` (optional) = orthonormalised products made; `residual` (optional) = the last
 * distance read, -1 if none was.  world > 1: the ranks hold identical V and Z, so all of them
 * leave at the same product. */
BB_API int bb_solver_spectral_init_tol(bb_solver *s, int n_iter, double tol, const double *v0,
                                       int *iters_done, double *residual);

/* Stress of the current coordinates (one gradient pass, no update). */
BB_API int bb_solver_stress(bb_solver *s, double *stress);
/* Copies the stress history (one value per completed iteration since the
 * last bb_solver_set_coords) to the host; synchronises the stream. */
BB_API int bb_solver_get_stress_history(bb_solver *s, double *out, int64_t cap, int64_t *n);
BB_API int bb_solver_sync(bb_solver *s);
/* The same with a bound: BB_ERR_STATE if the stream has not drained after
 * `milliseconds` (polls hipStreamQuery; the work stays enqueued). */
BB_API int bb_solver_sync_timeout(bb_solver *s, int64_t milliseconds);

/* HIP-event timing of the dominant kernel (stress+gradient) and of the
 * reduce/update kernel on the solver's stream.  Averages are over the timed
 * launches since timing was (re-)enabled.  enabled = 1: every iteration is
 * timed; enabled = k > 1: every k-th one (three event records cost about 10 us
 * of stream time per timed iteration on MI355X, which matters once a step is
 * ~0.1 ms); 0: off. */
BB_API int bb_solver_set_timing(bb_solver *s, int enabled);
BB_API int bb_solver_get_timing(bb_solver *s, double *grad_ms_avg, double *reduce_ms_avg,
                                int64_t *launches);
/* The same events, start of one timed iteration to start of the next: the whole
 * device-side step including the exchange, the update and the gaps between
 * launches (average over consecutive timed iterations; 0 with fewer than two). */
BB_API int bb_solver_get_step_timing(bb_solver *s, double *step_ms_avg);
/* What a pair of HIP events costs by itself on the solver's stream: the average time
 * between two events recorded back to back behind a sweep launch (`pairs` of them): an
 * upper bound of what an event-timed interval contains besides its kernel (bench.py
 * reports it beside its event-timed figures; it subtracts nothing). */
BB_API int bb_solver_measure_event_gap(bb_solver *s, int pairs, double *ms_avg);
/* Measurement aid: average duration of a kernel that only READS this rank's
 * resident units (same grid, same per-wave chunks, same 8-row window, no
 * arithmetic) -- the practical HBM ceiling for the access pattern, to put
 * beside the spec peak in the roofline. */
BB_API int bb_solver_measure_stream_read(bb_solver *s, int launches, double *ms_avg);
/* Which kernel bb_solver_iterate runs: row_owner = 1 for the one-launch-per-iteration
 * path of small one-rank maps (both triangles resident; waves_per_row waves per bin),
 * 0 for the unit sweep + reduce.  Either pointer may be NULL. */
BB_API int bb_solver_iteration_path(const bb_solver *s, int *row_owner, int *waves_per_row);
/* Bytes of wish-distance data the stress+gradient kernel streams per launch on
 * this rank (resident units * unit bytes), and pairs evaluated. */
BB_API int bb_solver_traffic(const bb_solver *s, int64_t *unit_bytes, int64_t *pairs_dense);

/* ---------------------------------------------------------------------- */
/* A2/A3/A4  the ContactMap stage, resident on the device                   */
/*        (reference: blueberry/datatypes.pyx:97-116, :161-171, :122-141)   */
/* ---------------------------------------------------------------------- */

/* One (d, d) float64 matrix, d = n_bins + 1, that stays in HBM across
 * scatter -> normalize -> filter -> solver, as `ContactMap.matrix` stays in host
 * memory across the same calls in the reference (datatypes.pyx:97-120, :161-171,
 * :140-141).  No matrix-sized host transfer happens unless the caller asks for
 * one (bb_cm_upload / bb_cm_download).  Every operation is bit-exact fp64. */
typedef struct bb_cm bb_cm;
BB_API int bb_cm_create(bb_cm **out, int64_t d, int device);   /* zero-filled (pyx:99) */
BB_API int bb_cm_destroy(bb_cm *cm);
BB_API int bb_cm_dim(const bb_cm *cm, int64_t *d);
/* The resident matrix itself (row-major, leading dimension d) for callers that share
 * the HIP context; borrowed, valid until bb_cm_destroy (bb_cm_filter changes d). */
BB_API int bb_cm_device_ptr(const bb_cm *cm, const double **dev_matrix, int64_t *d, int *device);
BB_API int bb_cm_upload(bb_cm *cm, const double *matrix, int64_t ld);   /* host (d,d) -> device */
BB_API int bb_cm_download(bb_cm *cm, double *matrix, int64_t ld);       /* device -> host (d,d) */
/* ContactMap.__init__'s loop (pyx:110-116): zero the matrix, then for every triple
 * (column-major `triples`, as bb_contactmap_scatter) set [j][k] and [k][j]; later
 * triples overwrite earlier ones.  Needs no scratch the size of the matrix. */
BB_API int bb_cm_scatter(bb_cm *cm, const double *triples, int64_t n, int32_t resolution);
/* The same with two conveniences for a caller that holds C-ordered rows and wants
 * `regions` (pyx:120, numpy.union1d of the two position columns) without sorting 2n
 * doubles: row_major != 0 reads `triples` as (n, 3) rows instead of the reference's
 * column-major (3, n); `present` (d bytes) receives 1 for every bin some position falls
 * in and *on_grid 1 if every position equals bin * resolution exactly -- then regions
 * are the present bins times the resolution, bit for bit.  present / on_grid may both
 * be NULL. */
BB_API int bb_cm_scatter_ex(bb_cm *cm, const double *triples, int64_t n, int32_t resolution,
                            int32_t row_major, uint8_t *present, int32_t *on_grid);
/* ContactMap.normalize (pyx:161-171), in place: m[j][j+i] /= KRnorm[j]*KRnorm[j+i]*
 * KRexpected[i], mirrored, then nan_to_num over the whole matrix.  d must be n_bins+1. */
BB_API int bb_cm_normalize(bb_cm *cm, int64_t n_bins, const double *KRnorm,
                           const double *KRexpected);
/* Column marginals `matrix.sum(axis=0)` (pyx:140), d doubles to the host: rows are
 * added in order, which is bit for bit what numpy computes. */
BB_API int bb_cm_marginals(bb_cm *cm, double *sums);
/* ContactMap.filter (pyx:140-141): keep the rows and columns whose marginal is
 * > threshold (a NaN marginal is dropped, as `NaN > t` is false in numpy).  The
 * resident matrix becomes (d_new, d_new), compacted in place (same device pointer, leading
 * dimension d_new; the allocation keeps its size); keep_out (d bytes, may be NULL)
 * receives the 0/1 mask over the OLD indices. */
BB_API int bb_cm_filter(bb_cm *cm, double threshold, int64_t *d_new, uint8_t *keep_out);
/* y = M x over the resident matrix (x, y: d doubles on the host): one HBM sweep. */
BB_API int bb_cm_symv(bb_cm *cm, const double *x, double *y);
/* ContactMap.eigenvector (pyx:216-235: scipy.sparse.linalg.eigsh(matrix, k=1), i.e.
 * ARPACK's eigenpair of LARGEST MAGNITUDE): restarted Lanczos with full
 * re-orthogonalisation over the resident matrix, one HBM sweep per matrix-vector
 * product.  Stops when |M v - lambda v| <= tol * |lambda| or after max_matvecs products.
 * vec: d doubles, unit length, largest-magnitude component positive (ARPACK's sign is
 * arbitrary).  eigenvalue / matvecs_used / residual may be NULL. */
BB_API int bb_cm_eigenvector(bb_cm *cm, double *vec, double *eigenvalue, double tol,
                             int64_t max_matvecs, int64_t *matvecs_used, double *residual);
/* ContactMap.correlation (pyx:173-188): matrix <- numpy.corrcoef(matrix), in place on the
 * resident matrix: rows centred, Gram matrix on the fp64 matrix cores
 * (v_mfma_f64_16x16x4_f64; the one dense contraction on this path), scaled and clipped
 * in numpy's order of operations.  Equal to numpy to rounding (BLAS sums in another
 * order).  tflops (may be NULL): rate of the Gram kernel alone, by HIP events. */
BB_API int bb_cm_correlation(bb_cm *cm, double *tflops);
/* bb_cm_correlation's temporaries (centred rows + Gram matrix, about 2x the matrix) live in one
 * grow-only scratch per device, shared by every map on it; this frees it (it is made again by
 * the next call that needs it). */
BB_API int bb_cm_release_scratch(int device);
/* Hand the resident matrix to a solver of n_bins = d bins on the same device, device
 * to device (same meaning of kind / alpha as bb_solver_set_wish_dense). */
BB_API int bb_solver_set_wish_from_cm(bb_solver *s, const bb_cm *cm, int kind, double alpha);
/* The same into the bins [bin_offset, bin_offset + d) of a solver of several maps
 * (bb_solver_set_maps), d = the map's edge. */
BB_API int bb_solver_set_wish_from_cm_block(bb_solver *s, const bb_cm *cm, int64_t bin_offset,
                                            int kind, double alpha);

/* The same two loops around a HOST matrix (round-1 entry points; upload, kernel,
 * download): */

/* matrix (d,d) float64 host, d = n_bins+1, zero-filled on entry.  `triples`
 * is the (n,3) array exactly as the reference's pointer arithmetic reads it:
 * column-major (pos_i[0..n), pos_j[0..n), count[0..n)).  bin = pos/resolution
 * truncated; later triples overwrite earlier ones. */
BB_API int bb_contactmap_scatter(const double *triples, int64_t n, int32_t resolution,
                                 double *matrix, int64_t d, int device);
/* In place on the host matrix: m[j][j+i] /= KRnorm[j]*KRnorm[j+i]*KRexp[i],
 * mirrored, then numpy.nan_to_num over the whole matrix.  bit-exact fp64. */
BB_API int bb_contactmap_normalize(double *matrix, int64_t n_bins, const double *KRnorm,
                                   const double *KRexpected, int device);

/* ---------------------------------------------------------------------- */
/* The remaining numeric Cython helpers of blueberry.pyx                    */
/* ---------------------------------------------------------------------- */

/* q[i] = max_{k<=i} min(p[k] * n / (k+1), 1) for d sorted p-values and n tests
 * (reference: blueberry/blueberry.pyx:40-75; a parallel prefix-max, exact). */
BB_API int bb_benjamini_hochberg(const double *p_values, int64_t d, int64_t n, double *q_values,
                                 int device);
/* yp5i (n5,n5) float32, in place: yp5i[i][j] = max(yp5i[i][j], 5x5 block of yp1
 * (n1,n1)) for i, j < n5-1 (reference: blueberry/blueberry.pyx:93-104). */
BB_API int bb_downsample(const float *yp1, int64_t n1, float *yp5i, int64_t n5, int device);

#ifdef __cplusplus
}
#endif
#endif /* BLUEBERRY_HIP_H */
