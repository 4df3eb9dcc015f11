"""StructureSolver -- contact matrix -> 3D coordinates on MI355X.

Host side of the hot path BASELINE.json names.  The reference has no solver
(SURVEY.md section 0); the estimator shape follows the only estimator the
reference has, `FitHiC(hyper-parameters).fit_transform(data)`
(`blueberry/fithic.py:76-108`), and the input is the dense float64 matrix a
`ContactMap` holds (`blueberry/datatypes.pyx:78-86`).  The algorithm is
specified in docs/SPEC.md and runs entirely in libblueberry_hip.so
(include/blueberry_hip.h); this module only validates arguments, owns the
handle and, for world_size > 1, drives one all-reduce per iteration through
torch.distributed (backend "nccl" = RCCL over xGMI).
"""
import os
import sys

import numpy

from . import _lib

_DTYPES = {"float32": _lib.BB_F32, "float64": _lib.BB_F64}
_KINDS = {"wish": _lib.BB_KIND_WISH, "counts": _lib.BB_KIND_COUNTS}


class RankDeficient(RuntimeError):
    """The block power iteration of the device-resident spectral start lost rank: the map has
    fewer than three independent directions (an empty or unconstrained map, fewer than 4 bins).
    `fit()` then takes the host-driven start; every other library error propagates."""


class DeviceTriples(object):
    """The (n, 3) array [pos_i, pos_j, count] of a Rao-format file -- what
    `ContactMap.__init__` reads (`blueberry/datatypes.pyx:100-102`) -- copied to the device
    once (`bb_triples_*`): nan_to_num, binning and the scatter into the solver's tiles all
    happen there.  C-ordered rows and the reference's column-major array are read in place."""
    is_triples = True

    def __init__(self, triples, resolution, device):
        t = numpy.asarray(triples, dtype=numpy.float64)
        if t.ndim != 2 or t.shape[1] != 3:
            raise ValueError("triples must have shape (n, 3)")
        if t.flags.c_contiguous:
            buf, row_major = t, 1
        elif t.flags.f_contiguous:
            buf, row_major = t.T, 0
        else:
            buf, row_major = numpy.ascontiguousarray(t), 1
        self._lib = _lib.load()
        self._h = _lib.c_void_p()
        self.n, self.device = int(t.shape[0]), int(device)
        _lib.check(self._lib.bb_triples_create(self._h, _lib.as_f64_ptr(buf), self.n,
                                               int(resolution), row_major, self.device),
                   "bb_triples_create")

    def tiles(self, n_bins, dtype):
        """(tile_I, tile_J), device order, of the tiles the triples name."""
        import ctypes
        nb = layout_info(n_bins, dtype)["n_blocks"]
        present = numpy.zeros((nb, nb), dtype=numpy.uint8)
        _lib.check(self._lib.bb_triples_tiles(self._h, int(n_bins), _DTYPES[dtype],
                                              present.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                              nb), "bb_triples_tiles")
        tj, ti = numpy.nonzero(present.T)            # J ascending, then I
        return ti.astype(numpy.int32), tj.astype(numpy.int32)

    def close(self):
        if self._h:
            self._lib.bb_triples_destroy(self._h)
            self._h = _lib.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipEngine(object):
    """One rank's device state: thin, 1:1 over the bb_solver_* C-ABI."""

    def __init__(self, n_bins, dtype, rank=0, world=1, device=0, tiles=None):
        self._lib = _lib.load()
        self._h = _lib.c_void_p()
        self.n_bins, self.dtype, self.rank, self.world, self.device = (
            int(n_bins), dtype, int(rank), int(world), int(device))
        if tiles is None:
            ti = tj = None
            nt = 0
        else:
            ti = numpy.ascontiguousarray(tiles[0], dtype=numpy.int32)
            tj = numpy.ascontiguousarray(tiles[1], dtype=numpy.int32)
            if ti.shape != tj.shape or ti.ndim != 1:
                raise ValueError("tiles must be a pair of equal-length 1-D index arrays")
            nt = ti.shape[0]
        _lib.check(self._lib.bb_solver_create(
            self._h, self.n_bins, _DTYPES[dtype], self.device, self.rank, self.world,
            None if ti is None else ti.ctypes.data_as(_lib.p_i32),
            None if tj is None else tj.ctypes.data_as(_lib.p_i32), nt), "bb_solver_create")
        self._exch = None
        self._comm_state = None     # "peer" | "rccl" | "torch" | "host" once chosen
        self._peer = None           # outcome of peer_setup()
        self._peer_error = ""
        self._comm_trial = None     # timings of select_exchange's trial, if one ran

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if self._h:
            self._lib.bb_solver_destroy(self._h)
            self._h = _lib.c_void_p()
            self._exch = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- layout -----------------------------------------------------------
    def layout(self):
        info = _lib.LayoutInfo()
        ub, ue = _lib.c_i64(), _lib.c_i64()
        _lib.check(self._lib.bb_solver_layout(self._h, info, ub, ue), "bb_solver_layout")
        d = info.as_dict()
        d["u_begin"], d["u_end"] = int(ub.value), int(ue.value)
        return d

    # -- inputs -----------------------------------------------------------
    def set_wish_dense(self, matrix, kind, alpha):
        m = _check_square(matrix, self.n_bins)
        _lib.check(self._lib.bb_solver_set_wish_dense(
            self._h, _lib.as_f64_ptr(m), m.strides[0] // 8, _KINDS[kind], float(alpha)),
            "bb_solver_set_wish_dense")

    def set_wish_from_cm(self, dev_matrix, kind, alpha):
        """Device to device from a resident ContactMap matrix (datatypes._DeviceMatrix)."""
        _lib.check(self._lib.bb_solver_set_wish_from_cm(self._h, dev_matrix._h, _KINDS[kind],
                                                        float(alpha)),
                   "bb_solver_set_wish_from_cm")

    def set_wish_sparse(self, rows, cols, vals, kind, alpha, KRnorm=None, KRexpected=None):
        r = numpy.ascontiguousarray(rows, dtype=numpy.int64)
        c = numpy.ascontiguousarray(cols, dtype=numpy.int64)
        v = numpy.ascontiguousarray(vals, dtype=numpy.float64)
        if not (r.ndim == c.ndim == v.ndim == 1 and r.shape == c.shape == v.shape):
            raise ValueError("rows, cols, vals must be 1-D arrays of equal length")
        kr = ke = None
        if KRnorm is not None or KRexpected is not None:
            if KRnorm is None or KRexpected is None:
                raise ValueError("KRnorm and KRexpected go together")
            kr, ke = _pad_vector(KRnorm, self.n_bins), _pad_vector(KRexpected, self.n_bins)
        _lib.check(self._lib.bb_solver_set_wish_sparse(
            self._h, r.ctypes.data_as(_lib.p_i64), c.ctypes.data_as(_lib.p_i64),
            _lib.as_f64_ptr(v), r.shape[0], _KINDS[kind], float(alpha),
            None if kr is None else _lib.as_f64_ptr(kr),
            None if ke is None else _lib.as_f64_ptr(ke)), "bb_solver_set_wish_sparse")

    def set_wish_triples(self, dev_triples, kind, alpha, KRnorm=None, KRexpected=None):
        """From triples resident on the device (`DeviceTriples`): binned, KR / O-E
        normalised and converted there."""
        kr = ke = None
        if KRnorm is not None or KRexpected is not None:
            if KRnorm is None or KRexpected is None:
                raise ValueError("KRnorm and KRexpected go together")
            kr, ke = _pad_vector(KRnorm, self.n_bins), _pad_vector(KRexpected, self.n_bins)
        _lib.check(self._lib.bb_solver_set_wish_triples(
            self._h, dev_triples._h, _KINDS[kind], float(alpha),
            None if kr is None else _lib.as_f64_ptr(kr),
            None if ke is None else _lib.as_f64_ptr(ke)), "bb_solver_set_wish_triples")

    # -- several maps in one solver (bb_solver_set_maps) -----------------------
    def set_maps(self, bin_begin, lr_scale):
        """Declare the maps laid end to end in this solver: map m owns the bins
        [bin_begin[m], bin_begin[m + 1]) and steps with lr * lr_scale[m]."""
        b = numpy.ascontiguousarray(bin_begin, dtype=numpy.int64)
        sc = numpy.ascontiguousarray(lr_scale, dtype=numpy.float64)
        if b.ndim != 1 or sc.ndim != 1 or b.shape[0] != sc.shape[0] + 1:
            raise ValueError("bin_begin needs one entry more than lr_scale")
        _lib.check(self._lib.bb_solver_set_maps(self._h, sc.shape[0], b.ctypes.data_as(_lib.p_i64),
                                                _lib.as_f64_ptr(sc)), "bb_solver_set_maps")
        self.n_maps = int(sc.shape[0])

    def set_wish_dense_block(self, matrix, bin_offset, kind, alpha):
        m = _check_square(matrix, numpy.asarray(matrix).shape[0])
        _lib.check(self._lib.bb_solver_set_wish_dense_block(
            self._h, _lib.as_f64_ptr(m), m.strides[0] // 8, m.shape[0], int(bin_offset),
            _KINDS[kind], float(alpha)), "bb_solver_set_wish_dense_block")

    def set_wish_from_cm_block(self, dev_matrix, bin_offset, kind, alpha):
        _lib.check(self._lib.bb_solver_set_wish_from_cm_block(
            self._h, dev_matrix._h, int(bin_offset), _KINDS[kind], float(alpha)),
            "bb_solver_set_wish_from_cm_block")

    def set_block_steps(self, scale):
        """A step per block of the layout (`bb_solver_set_block_steps`): bin i moves by
        lr * scale[i // vw] * g_i; None = one step for all again."""
        if scale is None:
            _lib.check(self._lib.bb_solver_set_block_steps(self._h, None, 0),
                       "bb_solver_set_block_steps")
            return
        sc = numpy.ascontiguousarray(scale, dtype=numpy.float64)
        if sc.ndim != 1:
            raise ValueError("scale must be one factor per block")
        _lib.check(self._lib.bb_solver_set_block_steps(self._h, _lib.as_f64_ptr(sc), sc.shape[0]),
                   "bb_solver_set_block_steps")

    def set_bin_steps(self, scale):
        """A step per bin (`bb_solver_set_bin_steps`): bin i moves by lr * scale[i] * g_i;
        None = one step for all again."""
        if scale is None:
            _lib.check(self._lib.bb_solver_set_bin_steps(self._h, None, 0), "bb_solver_set_bin_steps")
            return
        sc = numpy.ascontiguousarray(scale, dtype=numpy.float64)
        if sc.ndim != 1:
            raise ValueError("scale must be one factor per bin")
        _lib.check(self._lib.bb_solver_set_bin_steps(self._h, _lib.as_f64_ptr(sc), sc.shape[0]),
                   "bb_solver_set_bin_steps")

    def degrees(self):
        """Per bin, the number of this rank's stored pairs that constrain it (delta > 0):
        `bb_solver_degrees`, one pass over the resident units."""
        out = numpy.zeros(self.n_bins, dtype=numpy.int64)
        _lib.check(self._lib.bb_solver_degrees(self._h, out.ctypes.data_as(_lib.p_i64), out.shape[0]),
                   "bb_solver_degrees")
        return out

    def stress_maps(self):
        out = numpy.empty(getattr(self, "n_maps", 1), dtype=numpy.float64)
        _lib.check(self._lib.bb_solver_stress_maps(self._h, _lib.as_f64_ptr(out), out.shape[0]),
                   "bb_solver_stress_maps")
        return out

    def set_wish_from_coords(self, xstar):
        x = _check_coords(xstar, self.n_bins)
        _lib.check(self._lib.bb_solver_set_wish_from_coords(self._h, _lib.as_f64_ptr(x)),
                   "bb_solver_set_wish_from_coords")

    def set_coords(self, x0):
        x = _check_coords(x0, self.n_bins)
        _lib.check(self._lib.bb_solver_set_coords(self._h, _lib.as_f64_ptr(x)),
                   "bb_solver_set_coords")

    def get_coords(self):
        out = numpy.empty((self.n_bins, 3), dtype=numpy.float64)
        _lib.check(self._lib.bb_solver_get_coords(self._h, _lib.as_f64_ptr(out)),
                   "bb_solver_get_coords")
        return out

    # -- iterations -------------------------------------------------------
    def set_momentum(self, mu):
        _lib.check(self._lib.bb_solver_set_momentum(self._h, float(mu)), "bb_solver_set_momentum")

    def iterate(self, iters, lr):
        _lib.check(self._lib.bb_solver_iterate(self._h, int(iters), float(lr)),
                   "bb_solver_iterate")

    def grad(self):
        _lib.check(self._lib.bb_solver_grad(self._h), "bb_solver_grad")

    def apply(self, lr):
        _lib.check(self._lib.bb_solver_apply(self._h, float(lr)), "bb_solver_apply")

    def matvec_sq(self, x):
        """(D o D) @ x for this rank's units; x is (n_bins, 3)."""
        x = _check_coords(x, self.n_bins)
        y = numpy.empty_like(x)
        _lib.check(self._lib.bb_solver_matvec_sq(self._h, _lib.as_f64_ptr(x), _lib.as_f64_ptr(y)),
                   "bb_solver_matvec_sq")
        return y

    def spectral_init_device(self, n_iter, v0, tol=0.0):
        """Classical-MDS start computed and left on the device: `bb_solver_spectral_init_tol`.
        v0: (n_bins, 3) start of the block power iteration (the same on every rank).  With
        several ranks it is collective and needs their exchange set up first (peer arenas or
        the library's communicator): the per-rank products are summed on the device, nothing
        of size N crosses PCIe.  tol > 0: n_iter is the most products made; the loop ends once
        B V lies within tol (relative) of span(V).  Returns (products orthonormalised, last
        distance read or -1).  Raises RankDeficient when the iterate lost rank."""
        v0 = _check_coords(v0, self.n_bins)
        done, res = _lib.c_int(), _lib.c_dbl()
        rc = self._lib.bb_solver_spectral_init_tol(self._h, int(n_iter), float(tol),
                                                   _lib.as_f64_ptr(v0), done, res)
        if rc == _lib.BB_ERR_STATE and "lost rank" in _lib.last_error():
            raise RankDeficient(_lib.last_error())
        _lib.check(rc, "bb_solver_spectral_init_tol")
        return int(done.value), float(res.value)

    def stress(self):
        out = _lib.c_dbl()
        _lib.check(self._lib.bb_solver_stress(self._h, out), "bb_solver_stress")
        return float(out.value)

    def stress_history(self):
        n = _lib.c_i64()
        _lib.check(self._lib.bb_solver_get_stress_history(self._h, None, 0, n),
                   "bb_solver_get_stress_history")
        out = numpy.empty(int(n.value), dtype=numpy.float64)
        if out.size:
            _lib.check(self._lib.bb_solver_get_stress_history(
                self._h, _lib.as_f64_ptr(out), out.size, n), "bb_solver_get_stress_history")
        return out

    def sync(self):
        _lib.check(self._lib.bb_solver_sync(self._h), "bb_solver_sync")

    def exchange_size(self):
        n = _lib.c_i64()
        _lib.check(self._lib.bb_solver_exchange_size(self._h, n), "bb_solver_exchange_size")
        return int(n.value)

    def read_exchange(self):
        """Exchange buffer [g (n_pad,3) | stress hi | lo] as float64 on the host."""
        out = numpy.empty(self.exchange_size(), dtype=numpy.float64)
        _lib.check(self._lib.bb_solver_read_exchange(self._h, _lib.as_f64_ptr(out), out.size),
                   "bb_solver_read_exchange")
        return out

    def write_exchange(self, host):
        host = numpy.ascontiguousarray(host, dtype=numpy.float64)
        _lib.check(self._lib.bb_solver_write_exchange(self._h, _lib.as_f64_ptr(host), host.size),
                   "bb_solver_write_exchange")

    # -- the all-reduce boundary (world > 1) --------------------------------
    def comm_setup(self):
        """Make the library's own RCCL communicator for this job: rank 0 draws the
        128-byte id, torch.distributed only carries it to the other ranks, every
        rank joins (`ncclCommInitRank`).  Collective.  Returns False -- on every
        rank alike -- when RCCL cannot be used, so that the caller can fall back
        to the torch.distributed all-reduce."""
        import ctypes
        import torch.distributed as dist
        box = [None]
        if self.rank == 0:
            buf = ctypes.create_string_buffer(128)
            if self._lib.bb_comm_unique_id(buf) == _lib.BB_OK:
                box[0] = buf.raw
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            return False
        ok = self._lib.bb_solver_comm_init(self._h, box[0]) == _lib.BB_OK
        # all ranks must take the same path: agree on the outcome
        import torch
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32,
                            device=torch.device("cuda", self.device))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    # the library's communicator cache (bb_comm_cached / bb_solver_comm_attach / _detach):
    # `comm_reuse` below drives these three
    def _comm_cached(self):
        import ctypes
        have = ctypes.c_int(0)
        self._lib.bb_comm_cached(self.device, self.rank, self.world, ctypes.byref(have))
        return bool(have.value)

    def _comm_generation(self):
        """Generation of the free cached communicator for this engine's key -- a hash of the
        unique id it was made with, equal on the ranks that made it together -- or 0."""
        import ctypes
        have, gen = ctypes.c_int(0), ctypes.c_uint64(0)
        self._lib.bb_comm_cached_generation(self.device, self.rank, self.world,
                                            ctypes.byref(have), ctypes.byref(gen))
        return int(gen.value) if have.value else 0

    def _comm_attach(self):
        return self._lib.bb_solver_comm_attach(self._h) == _lib.BB_OK

    def _comm_detach(self):
        self._lib.bb_solver_comm_detach(self._h)

    def iterate_dist(self, iters, lr):
        """`iters` x { grad, RCCL all-reduce, apply }, all enqueued by one C call."""
        _lib.check(self._lib.bb_solver_iterate_dist(self._h, int(iters), float(lr)),
                   "bb_solver_iterate_dist")

    def peer_setup(self):
        """Connect the peer exchange (one-shot all-reduce inside the solver's own
        kernels, include/blueberry_hip.h): every rank exports its receive arena,
        torch.distributed only carries the 128-byte handles, every rank maps all
        arenas.  Collective.  Returns False -- on every rank alike -- when any
        rank could not export or map, so that the caller can fall back to RCCL."""
        import ctypes
        import torch.distributed as dist
        if self._peer is not None:
            return self._peer
        buf = ctypes.create_string_buffer(_lib.BB_PEER_HANDLE_BYTES)
        mine = buf.raw if self._lib.bb_solver_peer_export(self._h, buf) == _lib.BB_OK else None
        handles = [None] * self.world
        dist.all_gather_object(handles, mine)
        ok = all(h is not None for h in handles)
        if ok:
            ok = self._lib.bb_solver_peer_connect(self._h, b"".join(handles)) == _lib.BB_OK
        if not ok:
            self._peer_error = _lib.last_error()
        oks = [None] * self.world
        dist.all_gather_object(oks, bool(ok))
        self._peer = all(oks)
        return self._peer

    def iterate_peer(self, iters, lr):
        """`iters` x { grad, reduce + push to every peer, wait + rank-ordered sum +
        update }, all enqueued by one C call."""
        _lib.check(self._lib.bb_solver_iterate_peer(self._h, int(iters), float(lr)),
                   "bb_solver_iterate_peer")

    def sync_timeout(self, milliseconds):
        """sync() that raises RuntimeError if the stream has not drained in time."""
        _lib.check(self._lib.bb_solver_sync_timeout(self._h, int(milliseconds)),
                   "bb_solver_sync_timeout")

    def comm_abort(self):
        _lib.check(self._lib.bb_solver_comm_abort(self._h), "bb_solver_comm_abort")

    def peer_form(self):
        """'one launch' (reduce, push, wait, sum and update in one kernel) or 'two launches'
        (include/blueberry_hip.h, peer exchange)."""
        one = _lib.c_int()
        _lib.check(self._lib.bb_solver_peer_form(self._h, one), "bb_solver_peer_form")
        return "one launch" if one.value else "two launches"

    def peer_set_form(self, one_launch):
        """Choose the exchange's form before its first use (the same on every rank)."""
        _lib.check(self._lib.bb_solver_peer_set_form(self._h, 1 if one_launch else 0),
                   "bb_solver_peer_set_form")

    def peer_set_timeout(self, milliseconds):
        _lib.check(self._lib.bb_solver_peer_set_timeout(self._h, int(milliseconds)),
                   "bb_solver_peer_set_timeout")

    def comm_world(self):
        """Ranks RCCL reports for the library's communicator, or None without one."""
        n = _lib.c_int()
        if self._lib.bb_solver_comm_world(self._h, n) != _lib.BB_OK:
            return None
        return int(n.value)

    def peer_status(self):
        """Synchronise and raise if a peer wait ran into its time limit."""
        st = _lib.c_int()
        _lib.check(self._lib.bb_solver_peer_status(self._h, st), "bb_solver_peer_status")
        return int(st.value)

    def exchange_tensor(self):
        """A torch tensor aliasing the exchange buffer [g (n_pad,3) | hi | lo].

        torch allocates it (plumbing: it is what torch.distributed can reduce)
        and the solver is told to write its partial gradient there; the solver
        is also moved onto torch's current stream so that the collective and
        the kernels are ordered without host synchronisation."""
        if self._exch is None:
            import torch
            rts = _lib.hip_runtimes_loaded()
            if len(rts) > 1:
                raise RuntimeError(
                    "two HIP runtimes are loaded (%s): import torch BEFORE the first "
                    "blueberry_amd compute call in a distributed job, so that "
                    "libblueberry_hip.so binds to the runtime torch uses" % ", ".join(rts))
            n = _lib.c_i64()
            _lib.check(self._lib.bb_solver_exchange_size(self._h, n), "bb_solver_exchange_size")
            tdt = torch.float32 if self.dtype == "float32" else torch.float64
            dev = torch.device("cuda", self.device)
            self._exch = torch.zeros(int(n.value), dtype=tdt, device=dev)
            stream = torch.cuda.current_stream(dev)
            _lib.check(self._lib.bb_solver_set_stream(self._h, _lib.c_void_p(stream.cuda_stream)),
                       "bb_solver_set_stream")
            _lib.check(self._lib.bb_solver_set_exchange_buffer(
                self._h, _lib.c_void_p(self._exch.data_ptr())), "bb_solver_set_exchange_buffer")
        return self._exch

    # -- measurement ------------------------------------------------------
    def set_timing(self, enabled):
        """True / 1: HIP events around every iteration; k > 1: every k-th; False: off."""
        _lib.check(self._lib.bb_solver_set_timing(self._h, int(enabled)),
                   "bb_solver_set_timing")

    def timing(self):
        g, r, n = _lib.c_dbl(), _lib.c_dbl(), _lib.c_i64()
        _lib.check(self._lib.bb_solver_get_timing(self._h, g, r, n), "bb_solver_get_timing")
        st = _lib.c_dbl()
        _lib.check(self._lib.bb_solver_get_step_timing(self._h, st), "bb_solver_get_step_timing")
        return {"grad_ms": float(g.value), "reduce_ms": float(r.value), "launches": int(n.value),
                "step_ms": float(st.value)}

    def event_gap_ms(self, pairs=16):
        """Average ms between two HIP events recorded back to back behind a sweep launch:
        what an event-timed interval contains besides its kernel (measurement aid)."""
        ms = _lib.c_dbl()
        _lib.check(self._lib.bb_solver_measure_event_gap(self._h, int(pairs), ms),
                   "bb_solver_measure_event_gap")
        return float(ms.value)

    def stream_read_ms(self, launches=10):
        """Average ms of a read-only sweep over the resident units (measurement aid)."""
        ms = _lib.c_dbl()
        _lib.check(self._lib.bb_solver_measure_stream_read(self._h, int(launches), ms),
                   "bb_solver_measure_stream_read")
        return float(ms.value)

    def iteration_path(self):
        """'row_owner' (small one-rank maps: one launch per iteration) or 'units'."""
        ro, wpr = _lib.c_int(), _lib.c_int()
        _lib.check(self._lib.bb_solver_iteration_path(self._h, ro, wpr), "bb_solver_iteration_path")
        return ("row_owner", int(wpr.value)) if ro.value else ("units", 0)

    def traffic(self):
        b, p = _lib.c_i64(), _lib.c_i64()
        _lib.check(self._lib.bb_solver_traffic(self._h, b, p), "bb_solver_traffic")
        return {"unit_bytes": int(b.value), "pairs_dense": int(p.value)}


def _check_square(matrix, n_bins):
    m = numpy.asarray(matrix)
    if m.ndim != 2 or m.shape[0] != m.shape[1]:
        raise ValueError("contact matrix must be square, got shape %r" % (m.shape,))
    if m.shape[0] != n_bins:
        raise ValueError("contact matrix has %d bins, solver was created for %d"
                         % (m.shape[0], n_bins))
    if m.dtype != numpy.float64 or m.strides[1] != 8 or m.strides[0] % 8 or m.strides[0] < 8 * n_bins:
        m = numpy.ascontiguousarray(m, dtype=numpy.float64)
    return m


def _pad_vector(v, n):
    """KR vectors have n_bins entries, the matrix n_bins+1 bins: pad with NaN
    (a NaN divisor = no constraint, as for unmappable bins in Rao's files)."""
    v = numpy.asarray(v, dtype=numpy.float64).ravel()
    if v.shape[0] > n:
        raise ValueError("KR vector longer than the number of bins")
    out = numpy.full(n, numpy.nan)
    out[:v.shape[0]] = v
    return out


def _check_coords(x, n_bins):
    x = numpy.ascontiguousarray(x, dtype=numpy.float64)
    if x.shape != (n_bins, 3):
        raise ValueError("coordinates must have shape (%d, 3), got %r" % (n_bins, x.shape))
    if not numpy.all(numpy.isfinite(x)):
        raise ValueError("coordinates must be finite")
    return x


def layout_info(n_bins, dtype):
    """bb_layout_dense_info as a dict (host-only arithmetic: no GPU needed)."""
    info = _lib.LayoutInfo()
    _lib.check(_lib.load().bb_layout_dense_info(int(n_bins), _DTYPES[dtype], info),
               "bb_layout_dense_info")
    return info.as_dict()


def tiles_from_entries(n_bins, rows, cols, dtype):
    """The (tile_I, tile_J) list -- device order: J ascending, then I -- of the
    tiles that hold at least one entry (rows[k], cols[k]); either triangle."""
    vw = layout_info(n_bins, dtype)["vw"]     # 512, or 128 for small fp64 problems
    r = numpy.asarray(rows, dtype=numpy.int64)
    c = numpy.asarray(cols, dtype=numpy.int64)
    if r.size and (min(r.min(), c.min()) < 0 or max(r.max(), c.max()) >= n_bins):
        raise ValueError("an index is outside [0, n_bins)")
    lo, hi = numpy.minimum(r, c) // vw, numpy.maximum(r, c) // vw
    nb = (int(n_bins) + vw - 1) // vw
    key = numpy.unique(hi * nb + lo)
    return (key % nb).astype(numpy.int32), (key // nb).astype(numpy.int32)


def tiles_from_blocks(n_bins, boundaries, band_bins, dtype):
    """Tile list of a block-sparse genome-wide map (BASELINE config 5; SURVEY.md 8(d):
    a tile is kept "when genomic separation <= band or the tile has any c > 0"): every
    tile that holds a pair of bins of ONE block of `boundaries` -- a chromosome's own
    contacts, where a Hi-C map is populated -- plus every tile that holds a pair of bins
    at most `band_bins` apart whatever the block.  `boundaries` = ascending bin offsets
    [0, ..., n_bins] (blueberry_amd.utils.genome_boundaries).  Replaces the dense
    `(n_bins+1)**2` float64 matrix of the reference (`blueberry/datatypes.pyx:99`: 720 GB
    at 10 kb) for whole-genome maps.  Returns (tile_I, tile_J) in device order and the
    number of stored pairs i < j < n_bins inside those tiles."""
    vw = layout_info(n_bins, dtype)["vw"]
    n = int(n_bins)
    b = numpy.asarray(boundaries, dtype=numpy.int64)
    if b.ndim != 1 or b.size < 2 or b[0] != 0 or b[-1] != n or (numpy.diff(b) < 0).any():
        raise ValueError("boundaries must ascend from 0 to n_bins")
    nb = (n + vw - 1) // vw
    first = numpy.arange(nb, dtype=numpy.int64) * vw
    last = numpy.minimum(first + vw, n) - 1
    blk_lo = numpy.searchsorted(b, first, side="right") - 1     # block of a tile's first bin
    blk_hi = numpy.searchsorted(b, last, side="right") - 1      # ... and of its last bin
    J, I = numpy.meshgrid(numpy.arange(nb), numpy.arange(nb))
    upper = I <= J
    same_block = blk_hi[I] >= blk_lo[J]           # I <= J and blocks are contiguous runs
    near = (J - I - 1) * vw + 1 <= int(band_bins)   # closest pair of bins of the two tiles
    sel = upper & (same_block | near)
    ti, tj = I[sel], J[sel]
    order = numpy.lexsort((ti, tj))
    ti, tj = ti[order].astype(numpy.int32), tj[order].astype(numpy.int32)
    rows = numpy.minimum(vw, n - ti.astype(numpy.int64) * vw)
    cols = numpy.minimum(vw, n - tj.astype(numpy.int64) * vw)
    pairs = int(numpy.where(ti == tj, rows * (rows - 1) // 2, rows * cols).sum())
    return (ti, tj), pairs


def block_degrees(n_bins, tiles, dtype):
    """Per block of the layout: an upper bound on the number of stored partners of its bins
    under a tile list (tiles in the block's row and column, times the tile edge)."""
    vw = layout_info(n_bins, dtype)["vw"]
    nb = (int(n_bins) + vw - 1) // vw
    ti, tj = numpy.asarray(tiles[0]), numpy.asarray(tiles[1])
    deg = numpy.bincount(ti, minlength=nb) + numpy.bincount(tj, minlength=nb)
    deg -= numpy.bincount(ti[ti == tj], minlength=nb)             # a diagonal tile counts once
    return numpy.minimum(int(n_bins), deg * vw).astype(numpy.int64)


def max_degree(n_bins, tiles, dtype):
    """Upper bound on the number of stored partners of any bin under a tile list: the step
    1 / (2 * max_degree) is the one SPEC 2.4 gives a dense map of that many bins."""
    return int(block_degrees(n_bins, tiles, dtype).max())


def block_step_factors(n_bins, tiles, dtype):
    """(lr, scale) of SPEC 2.4.1 for a tile list: lr = 1 / (2 max_degree) and, per block,
    scale[b] = max_degree / degree[b] -- every block takes the step 1 / (2 degree[b]) its own
    number of stored partners allows (a block without tiles keeps the factor 1)."""
    deg = block_degrees(n_bins, tiles, dtype)
    top = int(deg.max()) if deg.size and deg.max() > 0 else int(n_bins)
    return 1.0 / (2.0 * top), numpy.where(deg > 0, top / numpy.maximum(deg, 1), 1.0)


def degree_step_factors(degree):
    """(lr, scale) of SPEC 2.4.1 from the bins' degrees (number of constraining pairs):
    lr = 1 / (2 (D + 1)), D the largest degree, and scale[i] = (D + 1) / (degree[i] + 1), so
    that bin i steps by 1 / (2 (degree[i] + 1)) -- for a complete map of n bins exactly SPEC
    2.4's 1 / (2 n).  scale is None when all bins have the same degree."""
    deg = numpy.asarray(degree, dtype=numpy.int64)
    top = int(deg.max()) if deg.size else 0
    lr = 1.0 / (2.0 * (top + 1))
    if deg.size == 0 or int(deg.min()) == top:
        return lr, None
    return lr, (top + 1.0) / (deg + 1.0)


def _sum_over_ranks(counts, eng, world):
    """Element-wise sum of an int64 host array over the ranks (world 1: the array)."""
    if world <= 1:
        return counts
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(numpy.ascontiguousarray(counts, dtype=numpy.int64))
    if dist.get_backend() == "nccl":
        t = t.to(torch.device("cuda", getattr(eng, "device", 0)))
        dist.all_reduce(t)
        return t.cpu().numpy()
    dist.all_reduce(t)
    return t.numpy()


def allreduce_exchange(t):
    """Sum a device-resident exchange tensor over all ranks, in place: one
    all-reduce of 3*n_pad+2 elements (backend nccl = RCCL, over xGMI)."""
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM)


def allreduce_exchange_host(eng):
    """The same sum for a host-memory collective (backend gloo: CPU rehearsals,
    tests, 2 ranks sharing one GPU): read the buffer through the C-ABI, reduce
    on the host in float64, write it back.  Transport only -- the kernels on
    either side are the same ones the RCCL path runs."""
    import torch
    import torch.distributed as dist
    host = torch.from_numpy(eng.read_exchange())
    dist.all_reduce(host, op=dist.ReduceOp.SUM)
    eng.write_exchange(host.numpy())


def _dist_state(distributed):
    """(rank, world) of the running torch.distributed job, or (0, 1)."""
    if distributed is False:
        return 0, 1
    if distributed is None:
        # auto: a process that runs a torch.distributed job has imported it already;
        # importing torch here just to find that out costs the first fit() 0.7 s
        dist = sys.modules.get("torch.distributed")
        if dist is None:
            return 0, 1
    else:
        import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    if distributed:
        raise RuntimeError("distributed=True but torch.distributed is not initialised")
    return 0, 1


class StructureSolver(object):
    """Infer 3D bin coordinates from a Hi-C contact matrix (metric MDS).

    Minimises  S(X) = sum_{i<j, c_ij>0} (|x_i - x_j| - delta_ij)^2  with
    delta_ij = c_ij^(-1/alpha), by `n_iter` gradient steps  X <- X - lr * grad S
    (docs/SPEC.md).  With lr = 1/(2 N) and a complete matrix each step equals a
    SMACOF / Guttman-transform step, so the stress never increases.

    Parameters
    ----------
    n_iter : int
        Number of iterations (fixed; there is no convergence test on the device).
    lr : float or 'auto'
        Step size; 'auto' = 1 / (2 * n_bins).
    dtype : 'float32' or 'float64'
        Arithmetic type on the GPU.
    alpha : float
        Count-to-distance exponent, delta = c^(-1/alpha).
    kind : 'counts' or 'wish'
        Whether the input matrix holds contact counts or wish distances.
    seed : int
        Seed of the default initial coordinates (numpy default_rng standard normal).
    tol : float or None
        None: exactly n_iter steps.  Otherwise stop as soon as the relative stress
        decrease of one step is <= tol, checked every `check_every` steps (n_iter is
        then the maximum); `n_iter_` tells how many were run.
    init : 'random' or 'spectral'
        Start used when `fit()` gets no `init=` array: seeded standard normal, or
        classical MDS computed on the device (`spectral_init`).
    degree_steps : bool
        A step per bin from the map itself (SPEC 2.4.1): bin i steps by 1 / (2 (deg_i + 1)),
        deg_i = the number of pairs that constrain it (counted on the device, summed over
        the ranks), instead of one step for all (`lr='auto'`: 1 / (2 N), the step of a
        COMPLETE map).  For incomplete maps -- blocked-sparse input, a whole genome whose
        chromosomes differ in size, real maps whose long-range pairs have no contact -- that
        is each bin's own Guttman-like step: a genome-like map converges in about a third
        of the iterations.  A float `lr` is then the step of the bin with the most partners.
        A complete map: no effect.
    spectral_iter, spectral_tol : int, float
        The spectral start's block power iteration makes at most `spectral_iter` products
        and ends once B V lies within `spectral_tol` (relative) of span(V); 0 = always
        `spectral_iter` products.  `spectral_iterations_` tells how many were made.
    momentum : float in [0, 1)
        Heavy-ball coefficient mu: V <- mu V - lr g, X <- X + V.  0 = plain steps.
    device : int or None
        HIP device index; None = LOCAL_RANK (distributed) or 0.
    distributed : bool or None
        None: shard over the ranks of an initialised torch.distributed job if
        there is one.  Every rank passes the same matrix and gets the same result.

    Attributes
    ----------
    structure_ : numpy.ndarray, shape (n_bins, 3), float64
    stress_ : numpy.ndarray, shape (n_iter,) -- stress BEFORE each step
    n_bins_, lr_ : the problem size and the step actually used
    exchange_ : how the ranks summed their partial gradients ('rccl', 'peer', 'torch',
        'host'), None on one rank.  Results are reproducible bit for bit for a given
        transport; fit() never picks one by timing (that is bench.py's trial).

    Failure on several ranks.  A rank that fails mid-fit raises, and takes the library's
    communicator out of the cache (its peers may still sit in a collective on it).  With
    the peer exchange ('peer', BB_COMM=peer or a trial) in its one-launch form every wave
    decides for its own 64 coordinates, so a rank that dies IN THE MIDDLE of an exchange can
    leave its peers with part of a step applied: their fit() then raises too
    (`peer_status`), and the coordinates of a solver that raised are not a result.  Nothing
    of a rank that is late or gone ever arrives, so nothing is applied; the two-launch form
    (ranks sharing a GPU, BB_PEER_FUSED=0) applies a step whole or not at all.
    """

    def __init__(self, n_iter=100, lr="auto", dtype="float32", alpha=3.0, kind="counts",
                 seed=0, device=None, distributed=None, engine=None, momentum=0.0,
                 init="random", tol=None, check_every=10, spectral_iter=40, spectral_tol=1e-3,
                 degree_steps=False):
        if dtype not in _DTYPES:
            raise ValueError("dtype must be 'float32' or 'float64'")
        if kind not in _KINDS:
            raise ValueError("kind must be 'counts' or 'wish'")
        if int(n_iter) < 0:
            raise ValueError("n_iter must be >= 0")
        if int(n_iter) > (1 << 20):
            raise ValueError("n_iter is limited to 2**20: the per-iteration stress history "
                             "lives on the device (include/blueberry_hip.h, bb_solver_iterate)")
        if not (lr == "auto" or float(lr) > 0):
            raise ValueError("lr must be positive or 'auto'")
        if not float(alpha) > 0:
            raise ValueError("alpha must be positive")
        if not 0.0 <= float(momentum) < 1.0:
            raise ValueError("momentum must be in [0, 1)")
        self.momentum = float(momentum)
        if tol is not None and not float(tol) > 0:
            raise ValueError("tol must be positive or None")
        if int(check_every) < 1:
            raise ValueError("check_every must be >= 1")
        self.tol, self.check_every = (None if tol is None else float(tol)), int(check_every)
        if init not in ("random", "spectral"):
            raise ValueError("init must be 'random' or 'spectral' (or pass init= to fit())")
        self.init = init
        if int(spectral_iter) < 0 or not 0.0 <= float(spectral_tol) < 1.0:
            raise ValueError("need spectral_iter >= 0 and 0 <= spectral_tol < 1")
        self.spectral_iter, self.spectral_tol = int(spectral_iter), float(spectral_tol)
        self.degree_steps = bool(degree_steps)
        self.n_iter, self.lr, self.dtype, self.alpha, self.kind, self.seed = (
            int(n_iter), lr, dtype, float(alpha), kind, int(seed))
        self.device, self.distributed = device, distributed
        # Engine factory: the HIP engine unless a test injects another one to
        # rehearse the multi-rank orchestration without a GPU.
        self._engine_factory = engine if engine is not None else HipEngine

    def _pick_device(self, world):
        if self.device is not None:
            return int(self.device)
        return int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0

    def fit(self, X, init=None):
        """Solve for the structure of `X`: a ContactMap, a square ndarray, or a
        scipy.sparse matrix (symmetric; either triangle is enough; of several
        COO entries for one pair the last is kept) -- the sparse form never builds
        the dense matrix."""
        if getattr(X, "is_resident", False) and hasattr(self._engine_factory, "set_wish_from_cm"):
            # a ContactMap whose matrix lives in HBM: packed device to device
            n = X.shape[0]
            return self._fit_impl(X, n, init, None, None)
        matrix = getattr(X, "matrix", X)
        sparse = hasattr(matrix, "tocoo")          # any scipy.sparse matrix
        if sparse:
            matrix = matrix.tocoo()
        else:
            matrix = numpy.asarray(matrix)
        if matrix.ndim != 2 or matrix.shape[0] != matrix.shape[1]:
            raise ValueError("contact matrix must be square, got shape %r" % (matrix.shape,))
        n = matrix.shape[0]
        return self._fit_impl(matrix, n, init, None, None)

    def _fit_impl(self, matrix, n, init, KRnorm, KRexpected):
        resident = getattr(matrix, "is_resident", False)
        triples = getattr(matrix, "is_triples", False)
        sparse = hasattr(matrix, "row") and not resident
        if n < 2:
            raise ValueError("need at least 2 bins (the contact map is empty)" if n == 0 else
                             "need at least 2 bins")
        rank, world = _dist_state(self.distributed)
        lr = 1.0 / (2.0 * n) if self.lr == "auto" else float(self.lr)
        if init is None and self.init == "random":
            init = numpy.random.default_rng(self.seed).standard_normal((n, 3))

        tiles = None
        if sparse:
            # blocked-sparse: only the tiles that hold an entry exist on the device
            keep = matrix.row != matrix.col
            rows, cols, vals = matrix.row[keep], matrix.col[keep], matrix.data[keep]
            tiles = tiles_from_entries(n, rows, cols, self.dtype)
        elif triples:
            tiles = matrix.tiles(n, self.dtype)
        eng = self._engine_factory(n, self.dtype, rank=rank, world=world,
                                   device=self._pick_device(world), tiles=tiles)
        try:
            if resident:
                dev = matrix._resident()
                if dev.device == eng.device:
                    eng.set_wish_from_cm(dev, self.kind, self.alpha)
                else:
                    # the map lives on another GPU than this rank's solver (a ContactMap made
                    # with an explicit device): through the host once, like a plain matrix
                    eng.set_wish_dense(matrix.to_host(), self.kind, self.alpha)
            elif triples:
                eng.set_wish_triples(matrix, self.kind, self.alpha, KRnorm, KRexpected)
            elif sparse:
                eng.set_wish_sparse(rows, cols, vals, self.kind, self.alpha, KRnorm, KRexpected)
            else:
                eng.set_wish_dense(matrix, self.kind, self.alpha)
            if self.degree_steps:
                lr_top, scale = degree_step_factors(_sum_over_ranks(eng.degrees(), eng, world))
                if self.lr == "auto":
                    lr = lr_top
                if scale is not None:
                    eng.set_bin_steps(scale)
            on_device = False
            if init is None:
                # 'spectral'.  The whole block power iteration stays on the device -- on one
                # rank, and on several once they have their exchange (peer arenas or the
                # library's communicator: the per-rank products are then summed where they
                # are).  Its Cholesky-QR needs an iterate of full column rank; maps it cannot
                # take -- fewer than 4 bins (centring leaves at most 2 directions), an empty
                # or unconstrained map -- and transports that sum on the host (gloo, torch)
                # go through the host-driven form below, so the same input runs everywhere.
                v0 = numpy.random.default_rng(self.seed).standard_normal((n, 3))
                if world > 1:
                    eng.set_coords(v0)             # (a trial in select_exchange wants a start)
                    select_exchange(eng, lr)
                device_form = hasattr(eng, "spectral_init_device") and n >= 4 and (
                    world == 1 or getattr(eng, "_comm_state", None) in ("peer", "rccl"))
                if device_form:
                    try:
                        info = eng.spectral_init_device(self.spectral_iter, v0, tol=self.spectral_tol)
                        self.spectral_iterations_ = info[0] if info else self.spectral_iter
                        on_device = True
                    except RankDeficient:
                        pass
                    if world > 1:
                        if getattr(eng, "_comm_state", None) == "peer":
                            eng.peer_status()
                        on_device = _all_ranks(on_device)
            if not on_device:
                if init is None:                   # 'spectral', host-driven
                    init, self.spectral_iterations_ = spectral_init(
                        eng, n, world, n_iter=self.spectral_iter, seed=self.seed,
                        tol=self.spectral_tol, return_iterations=True)
                eng.set_coords(init)
            if self.momentum:
                eng.set_momentum(self.momentum)
            if world > 1:
                select_exchange(eng, lr)
            if self.tol is None:
                run_iterations(eng, self.n_iter, lr, world)
            else:
                # early stop: every `check_every` steps the stress history is read
                # back (one sync) and the loop ends once the relative decrease per
                # step falls below tol.  Every rank sees the same all-reduced
                # stress, so all ranks stop at the same iteration.
                done = 0
                while done < self.n_iter:
                    k = min(self.check_every, self.n_iter - done)
                    run_iterations(eng, k, lr, world)
                    done += k
                    h = eng.stress_history()
                    if h.size >= 2 and h[-2] > 0 and \
                            abs(h[-2] - h[-1]) <= self.tol * h[-2]:
                        break
            if getattr(eng, "_comm_state", None) == "peer":
                eng.peer_status()              # raises if a peer wait ran into its time limit
            self.exchange_ = getattr(eng, "_comm_state", None)
            self.structure_ = eng.get_coords()
            self.stress_ = eng.stress_history()
        except BaseException:
            # this rank leaves a multi-rank job in the middle: its peers may sit in a collective
            # on the library's communicator, which must then not go back into the cache for the
            # next fit() to borrow (close() would return it as free)
            if getattr(eng, "_comm_state", None) == "rccl" and hasattr(eng, "comm_abort"):
                try:
                    eng.comm_abort()
                except Exception:                  # noqa: BLE001 -- the first error is the one to report
                    pass
            raise
        finally:
            eng.close()
        self.n_bins_, self.lr_, self.n_iter_ = n, lr, int(self.stress_.shape[0])
        return self

    def fit_triples(self, triples, resolution, n_bins, KRnorm=None, KRexpected=None, init=None):
        """Solve straight from a Rao-format sparse file's content, never building
        the dense matrix: `triples` is the (n, 3) array [pos_i, pos_j, count] that
        `ContactMap.__init__` reads (reference `blueberry/datatypes.pyx:100-102`),
        bins are `int(pos / resolution)` (pyx:111-112), the matrix has
        `n_bins + 1` bins (pyx:97), and with KRnorm / KRexpected each count is
        balanced and O/E-normalised on the device as `ContactMap.normalize`
        would (pyx:166-169).  A bin pair that occurs more than once keeps its last
        count, as in the reference's scatter (pyx:115-116)."""
        n = int(n_bins) + 1
        if KRnorm is not None and (numpy.any(numpy.asarray(KRnorm) == 0.0)
                                   or numpy.any(numpy.asarray(KRexpected)[:n_bins] == 0.0)):
            raise ZeroDivisionError("float division")      # as ContactMap.normalize
        if hasattr(self._engine_factory, "set_wish_triples"):
            # nan_to_num (pyx:102), the binning and the tile occupancy on the device: the host
            # never makes a pass over the triples (round 3: two isfinite passes, two divide +
            # astype passes and a scipy COO over 240 MB at chr1@10kb)
            _, world = _dist_state(self.distributed)
            dev = DeviceTriples(triples, resolution, self._pick_device(world))
            try:
                return self._fit_impl(dev, n, init, KRnorm, KRexpected)
            finally:
                dev.close()
        # engines without a device (tests): bin on the host, as round 3 did
        from .datatypes import _nan_to_num
        t = _nan_to_num(triples)          # pyx:102 (no copy when every value is finite)
        if t.ndim != 2 or t.shape[1] != 3:
            raise ValueError("triples must have shape (n, 3)")
        rows = (t[:, 0] / resolution).astype(numpy.int64)
        cols = (t[:, 1] / resolution).astype(numpy.int64)
        import scipy.sparse
        keep = rows != cols
        sp = scipy.sparse.coo_matrix((t[keep, 2], (rows[keep], cols[keep])), shape=(n, n))
        return self._fit_impl(sp, n, init, KRnorm, KRexpected)

    def fit_many(self, maps, inits=None):
        """Solve SEVERAL maps at once on one GPU -- e.g. the 23 per-chromosome ContactMaps
        of a genome (the reference's ContactMap is per chromosome, `blueberry/
        datatypes.pyx:88`), each of which alone is launch-bound (5-20 us per iteration
        whatever its size).  The maps are laid end to end in one blocked-sparse solver
        (`bb_solver_set_maps`): one sweep and one reduce launch per iteration serve all of
        them, every map with its own step (`lr='auto'`: 1 / (2 n_m)) and its own stress
        history.  Same iteration as `fit()` map by map; results agree with the single fits
        to rounding (1e-5 fp32 / 1e-12 fp64: the partial sums are cut differently), not bit
        for bit.  Inside a torch.distributed job the maps are dealt to the ranks by size,
        every rank solves its own in one solver on its GPU, and every rank gets all results
        (`ranks_of_maps_` tells who solved what): no exchange during the iterations.

        maps: sequence of ContactMaps (resident ones are packed device to device), square
        ndarrays or anything `numpy.asarray` takes.  inits: None, or one (n_m, 3) start per
        map (None entries: the seeded default of `fit()`).
        Sets `structures_` (list of (n_m, 3) float64), `stresses_` (list of per-iteration
        arrays), `n_bins_many_`, `lrs_`, `n_iter_`; returns self."""
        if not hasattr(self._engine_factory, "set_maps"):
            raise TypeError("fit_many needs an engine that holds several maps (HipEngine)")
        maps = list(maps)
        if not maps:
            raise ValueError("fit_many needs at least one map")
        if inits is None:
            inits = [None] * len(maps)
        if len(inits) != len(maps):
            raise ValueError("inits needs one entry per map")
        rank, world = _dist_state(self.distributed)
        if world > 1:
            # Several GPUs: the maps are independent problems, so every rank solves maps of its
            # own (dealt by size, largest first, to the rank with the least pairs so far) and
            # the results are gathered -- no exchange during the iterations at all.
            import torch.distributed as dist
            sizes = [int(getattr(X, "shape", numpy.shape(getattr(X, "matrix", X)))[0]) for X in maps]
            load, mine = [0] * world, [[] for _ in range(world)]
            for m in sorted(range(len(maps)), key=lambda q: (-sizes[q], q)):
                r = min(range(world), key=lambda q: (load[q], q))
                load[r] += sizes[m] * sizes[m]
                mine[r].append(m)
            part = None
            if mine[rank]:
                local = StructureSolver(n_iter=self.n_iter, lr=self.lr, dtype=self.dtype,
                                        alpha=self.alpha, kind=self.kind, seed=self.seed,
                                        device=self._pick_device(world), distributed=False,
                                        engine=self._engine_factory, momentum=self.momentum,
                                        init=self.init, tol=self.tol, check_every=self.check_every,
                                        spectral_iter=self.spectral_iter,
                                        spectral_tol=self.spectral_tol,
                                        degree_steps=self.degree_steps)
                local._fit_many_local([maps[m] for m in mine[rank]], [inits[m] for m in mine[rank]])
                part = (mine[rank], local.structures_, local.stresses_, local.lrs_)
            parts = [None] * world
            dist.all_gather_object(parts, part)
            n = len(maps)
            self.structures_, self.stresses_, self.lrs_ = [None] * n, [None] * n, [None] * n
            for p in parts:
                if p is not None:
                    for k, m in enumerate(p[0]):
                        self.structures_[m], self.stresses_[m], self.lrs_[m] = p[1][k], p[2][k], p[3][k]
            self.n_bins_many_ = sizes
            self.n_iter_ = max(int(h.shape[0]) for h in self.stresses_)
            self.ranks_of_maps_ = [next(r for r in range(world) if m in mine[r]) for m in range(n)]
            return self
        return self._fit_many_local(maps, inits)

    def _fit_many_local(self, maps, inits):
        """fit_many on this rank's GPU (see fit_many)."""
        sizes, srcs = [], []
        for X in maps:
            if getattr(X, "is_resident", False):
                sizes.append(int(X.shape[0]))
                srcs.append(X)
            else:
                m = numpy.asarray(getattr(X, "matrix", X))
                if m.ndim != 2 or m.shape[0] != m.shape[1]:
                    raise ValueError("contact matrix must be square, got shape %r" % (m.shape,))
                sizes.append(int(m.shape[0]))
                srcs.append(m)
        if min(sizes) < 2:
            raise ValueError("every map needs at least 2 bins")
        # the tile edge of the joint layout: 128 only for a small fp64 problem
        def layout(vw):
            off = [0]
            for n in sizes[:-1]:
                off.append(off[-1] + -(-n // vw) * vw)
            return off, off[-1] + sizes[-1]
        off, total = layout(128)
        if layout_info(total, self.dtype)["vw"] != 128:
            off, total = layout(512)
        vw = layout_info(total, self.dtype)["vw"]
        ti, tj = [], []
        for o, n in zip(off, sizes):
            b0, b1 = o // vw, (o + n + vw - 1) // vw
            jj, ii = numpy.meshgrid(numpy.arange(b0, b1), numpy.arange(b0, b1))
            sel = ii <= jj
            ti.append(ii[sel])
            tj.append(jj[sel])
        ti, tj = numpy.concatenate(ti), numpy.concatenate(tj)
        order = numpy.lexsort((ti, tj))
        tiles = (ti[order].astype(numpy.int32), tj[order].astype(numpy.int32))
        lrs = [1.0 / (2.0 * n) if self.lr == "auto" else float(self.lr) for n in sizes]
        device = self._pick_device(1)
        x0 = numpy.zeros((total, 3))
        for m, (o, n) in enumerate(zip(off, sizes)):
            init = inits[m]
            if init is None and self.init == "spectral":
                # each map's classical-MDS start from a solver of its own (device form)
                one = StructureSolver(n_iter=0, lr=self.lr, dtype=self.dtype, alpha=self.alpha,
                                      kind=self.kind, seed=self.seed, device=device,
                                      distributed=False, init="spectral",
                                      spectral_iter=self.spectral_iter,
                                      spectral_tol=self.spectral_tol,
                                      engine=self._engine_factory).fit(maps[m])
                init = one.structure_
            elif init is None:
                init = numpy.random.default_rng(self.seed).standard_normal((n, 3))
            x0[o:o + n] = _check_coords(init, n)
        eng = self._engine_factory(total, self.dtype, rank=0, world=1, device=device, tiles=tiles)
        try:
            eng.set_maps(off + [total], lrs)
            for o, n, src in zip(off, sizes, srcs):
                if getattr(src, "is_resident", False) and src._resident().device == eng.device:
                    eng.set_wish_from_cm_block(src._resident(), o, self.kind, self.alpha)
                else:
                    m = src.to_host() if getattr(src, "is_resident", False) else src
                    eng.set_wish_dense_block(m, o, self.kind, self.alpha)
            if self.degree_steps:
                # SPEC 2.4.1 per map: bin i steps by 1 / (2 (deg_i + 1)) (lr = 1 below); a float
                # `lr` stays the step of each map's best-connected bin
                deg = eng.degrees()
                steps = numpy.ones(total)
                for q, (o, n) in enumerate(zip(off, sizes)):
                    d = deg[o:o + n]
                    top = 1.0 / (2.0 * (d.max() + 1.0)) if self.lr == "auto" else float(self.lr)
                    steps[o:o + n] = top * (d.max() + 1.0) / (d + 1.0)
                    lrs[q] = top
                eng.set_bin_steps(steps)
            eng.set_coords(x0)
            if self.momentum:
                eng.set_momentum(self.momentum)
            nm = len(sizes)
            if self.tol is None:
                eng.iterate(self.n_iter, 1.0)
            else:
                done = 0
                while done < self.n_iter:
                    k = min(self.check_every, self.n_iter - done)
                    eng.iterate(k, 1.0)
                    done += k
                    h = eng.stress_history().reshape(-1, nm)
                    if h.shape[0] >= 2 and numpy.all(
                            numpy.abs(h[-2] - h[-1]) <= self.tol * numpy.maximum(h[-2], 1e-300)):
                        break
            X = eng.get_coords()
            hist = eng.stress_history().reshape(-1, nm)
        finally:
            eng.close()
        self.structures_ = [X[o:o + n].copy() for o, n in zip(off, sizes)]
        self.stresses_ = [hist[:, m].copy() for m in range(nm)]
        self.n_bins_many_, self.lrs_, self.n_iter_ = sizes, lrs, int(hist.shape[0])
        return self

    def fit_transform(self, X, init=None):
        """`fit(X)` and return the (n_bins, 3) coordinates."""
        return self.fit(X, init=init).structure_


def spectral_init(eng, n, world, n_iter=40, seed=0, tol=0.0, return_iterations=False):
    """Classical-MDS start: the top three eigenpairs of B = -1/2 J (D o D) J,
    J = I - 11'/n, by block power iteration with a Rayleigh-Ritz step, using the
    device matvec over the resident units (`bb_solver_matvec_sq`); X0 = V sqrt(L).
    This is the host-driven form (several ranks: the per-rank products are summed over
    the ranks between steps; and engines without a device); on one rank
    `HipEngine.spectral_init_device` runs the same iteration without leaving the device.
    Exact (up to a rigid motion) for a complete, noise-free distance matrix; for
    incomplete maps the missing pairs count as zero distance, so it is a start,
    not a solution.  tol > 0: the stopping rule of `bb_solver_spectral_init_tol` (after every
    product but the first, ||Z - V V'Z||_F / ||Z||_F < tol ends the loop).  Plays the part
    SURVEY.md 8(f)-2 assigns to the reference's `ContactMap.eigenvector`
    (`blueberry/datatypes.pyx:216-235`)."""
    def apply_B(V):
        U = V - V.mean(axis=0)
        W = eng.matvec_sq(U)
        if world > 1:
            import torch
            import torch.distributed as dist
            t = torch.from_numpy(W)
            if dist.get_backend() == "nccl":
                t = t.to(torch.device("cuda", getattr(eng, "device", 0)))
                dist.all_reduce(t)
                W = t.cpu().numpy()
            else:
                dist.all_reduce(t)
        return -0.5 * (W - W.mean(axis=0))

    def orth(A):
        # fewer than 3 bins: QR gives fewer than 3 columns; the missing ones are zero directions
        Q = numpy.linalg.qr(A)[0]
        if Q.shape[1] < 3:
            Q = numpy.hstack([Q, numpy.zeros((n, 3 - Q.shape[1]))])
        return Q

    G0 = numpy.random.default_rng(seed).standard_normal((n, 3))
    V = orth(G0)
    done, Z = 0, None
    for it in range(int(n_iter)):
        Z = apply_B(V)
        if tol > 0.0 and it > 0:
            zz, G = float((Z * Z).sum()), V.T @ Z
            if (numpy.sqrt(max(0.0, zz - float((G * G).sum())) / zz) if zz > 0.0 else 0.0) < tol:
                break
        V, Z = orth(Z), None
        done = it + 1
    if Z is None:
        Z = apply_B(V)
    evals, evecs = numpy.linalg.eigh(0.5 * (V.T @ Z + Z.T @ V))       # Rayleigh-Ritz, 3x3
    order = numpy.argsort(evals)[::-1]
    U = V @ evecs[:, order]
    # an eigenvector's sign is arbitrary (numpy's eigh and the device path's Jacobi sweeps
    # need not agree): every Ritz vector is turned to the side of the start's first column,
    # here and in bb_solver_spectral_init, so one seed gives one start on every world size
    U = U * numpy.where(U.T @ G0[:, 0] < 0.0, -1.0, 1.0)
    x0 = U * numpy.sqrt(numpy.maximum(evals[order], 0.0))
    return (x0, done) if return_iterations else x0


def _all_ranks(ok):
    """Collective AND over the ranks of one local outcome: every rank learns whether
    ALL of them succeeded, so that all of them take the same next step."""
    import torch.distributed as dist
    flags = [None] * dist.get_world_size()
    dist.all_gather_object(flags, bool(ok))
    return all(flags)


_TRIAL_PEER_TIMEOUT_MS = 2000
_TRIAL_SYNC_TIMEOUT_MS = int(os.environ.get("BB_TRIAL_SYNC_TIMEOUT_MS", "30000"))


def _trial_leg(eng, name, step, lr, x0, iters):
    """One transport's trial run from the start x0: one step (coordinates kept for the
    agreement check), ten more to warm up, `iters` timed.  After EVERY stage the ranks
    agree on whether all of them got through it; the first stage that failed anywhere
    ends the leg on every rank, so nobody enters a collective that a peer has left.
    Returns (ok, coordinates after one step, seconds per iteration)."""
    import time
    import torch.distributed as dist
    box = {}

    def stage(fn):
        try:
            fn()
            ok = True
        except Exception as exc:                 # this transport is just not used
            eng._comm_trial_error = "%s: %s" % (name, exc)
            ok = False
        return _all_ranks(ok)

    def settle():
        # bounded: a collective that a peer never joined must end the leg, not the job
        bounded = getattr(eng, "sync_timeout", None)
        if bounded:
            bounded(_TRIAL_SYNC_TIMEOUT_MS)
        else:
            eng.sync()
        if name == "peer":
            eng.peer_status()                    # raises when a wait ran into its limit

    def first():
        eng.set_coords(x0)
        step(1, lr)
        settle()
        box["x1"] = eng.get_coords()

    def warm():
        step(10, lr)
        settle()

    def timed():
        t0 = time.perf_counter()
        step(iters, lr)
        settle()
        box["dt"] = (time.perf_counter() - t0) / iters
        box["xk"] = eng.get_coords()             # after 1 + 10 + iters steps (outside the clock)

    for k, fn in enumerate((first, warm, timed)):
        if k == 2:
            dist.barrier()
        if not stage(fn):
            return False, None, float("inf")
    # coordinates after the FIRST step and after the LAST: a transport that delivers a stale
    # or torn partial now and then shows in the second even when the first step went well
    return True, (box["x1"], box["xk"]), box["dt"]


def comm_reuse(eng):
    """Borrow the communicator an earlier solver of this job made: the library keeps one
    per (device, rank, world) for the life of the process, so only the first multi-rank
    fit() pays ncclCommInitRank (0.1-1 s; round 2 made and destroyed one per fit).
    Collective, and the same on every rank by construction: the ranks first agree that
    EVERY one of them holds a free cached communicator, then that every attach worked;
    otherwise nobody uses the cache and `comm_setup` makes a fresh one everywhere."""
    if hasattr(eng, "_comm_generation"):
        # ... and that it is the SAME communicator on all of them: the cache is keyed by
        # (device, rank, world) only, and a rank can hold one of another generation -- made
        # while a peer's earlier solver still held the previous one, or by an earlier process
        # group of the same size.  Attaching those would hang the first all-reduce.
        import torch.distributed as dist
        gens = [None] * dist.get_world_size()
        dist.all_gather_object(gens, eng._comm_generation())
        if gens[0] == 0 or any(g != gens[0] for g in gens):
            return False
    elif not _all_ranks(eng._comm_cached()):
        return False
    ok = eng._comm_attach()
    if not _all_ranks(ok):
        if ok:
            eng._comm_detach()
        return False
    return True


def _comm_get(eng):
    """The library's communicator for this engine: the one an earlier fit() of this job
    left in the library's cache (`comm_reuse`), else a new one (`comm_setup`).  Collective."""
    if hasattr(eng, "_comm_cached") and comm_reuse(eng):
        return True
    return eng.comm_setup()


def select_exchange(eng, lr, trial=False):
    """Decide, once per engine and identically on every rank, how the partial
    gradients are summed over the ranks.  Collective.

    BB_COMM = auto (default) | peer | rccl | torch | host
      peer   one-shot exchange inside the solver's own kernels (IPC-mapped arenas)
      rccl   the library's own RCCL communicator, all-reduce enqueued from C
      torch  torch.distributed all-reduce on a tensor aliasing the exchange buffer
      host   exchange buffer staged through host memory (gloo / CPU rehearsals)
    auto on an RCCL job means rccl (torch if the communicator cannot be made).  With
    `trial` true -- bench.py, or BB_COMM_TRIAL=1 -- and the coordinates just set, auto
    also sets up peer and runs a few iterations through both from the same start:
    peer is taken only if its coordinates agree with RCCL's and it is faster on the
    slowest rank; the start is restored afterwards.  Every stage of the trial ends
    with an agreement between the ranks (`_trial_leg`), and the peer waits are cut
    to 2 s while it runs, so a transport that fails on one rank is dropped by all of
    them instead of leaving the others inside a collective.  The outcome is kept in
    eng._comm_state / eng._comm_trial."""
    if getattr(eng, "_comm_state", None):
        return eng._comm_state
    import torch.distributed as dist
    want = os.environ.get("BB_COMM", "auto")
    if want not in ("auto", "peer", "rccl", "torch", "host"):
        raise ValueError("BB_COMM must be auto, peer, rccl, torch or host, not %r" % want)
    trial = bool(trial) or os.environ.get("BB_COMM_TRIAL") == "1"
    native = hasattr(eng, "peer_setup")
    nccl = dist.get_backend() == "nccl"
    state = None
    if not native or want == "host" or (not nccl and want != "peer"):
        state = "host"
    elif want == "peer":
        if not eng.peer_setup():
            raise RuntimeError("BB_COMM=peer but the peer exchange could not be set up: %s"
                               % eng._peer_error)
        # No trial has compared this exchange with RCCL: keep the two-launch form, which
        # applies a step whole or not at all and whose flags are ordered by release / acquire.
        # The one-launch form (data as its own flag, no fences) is taken where a trial has
        # validated it against RCCL on this very job (below), or on request (BB_PEER_FUSED=1).
        if not trial and os.environ.get("BB_PEER_FUSED") is None and hasattr(eng, "peer_set_form"):
            eng.peer_set_form(False)
        state = "peer"
    elif want == "torch":
        state = "torch"
    elif want == "rccl" or not trial:
        state = "rccl" if _comm_get(eng) else "torch"
    else:
        have_rccl = _comm_get(eng)
        have_peer = eng.peer_setup()
        # what the peer exchange is held against, and what runs if it loses: the library's
        # communicator, or -- when that could not be made -- torch.distributed's all-reduce on
        # the exchange buffer (round 3 gave the peer exchange up with it: no RCCL, no trial)
        ref = "rccl" if have_rccl else "torch"

        def torch_step(k, step_lr):
            t = eng.exchange_tensor()
            for _ in range(k):
                eng.grad()
                allreduce_exchange(t)
                eng.apply(step_lr)

        ref_step = eng.iterate_dist if have_rccl else torch_step
        if have_peer and (have_rccl or hasattr(eng, "exchange_tensor")):
            x0 = eng.get_coords()
            set_limit = getattr(eng, "peer_set_timeout", None)
            if set_limit:
                set_limit(_TRIAL_PEER_TIMEOUT_MS)
            runs = {}

            def abort_if_failed(name):
                if name == "rccl" and not runs[name][0] and hasattr(eng, "comm_abort"):
                    # the library's communicator is suspect (this rank may still sit in a
                    # collective a peer never joined): abort it -- every rank does,
                    # the leg's outcome is shared -- so that the stream drains again
                    try:
                        eng.comm_abort()
                    except Exception as exc:
                        eng._comm_trial_error = "%s; abort: %s" % (
                            getattr(eng, "_comm_trial_error", None), exc)

            for name, step in ((ref, ref_step), ("peer", eng.iterate_peer)):
                runs[name] = _trial_leg(eng, name, step, lr, x0, 30)
                abort_if_failed(name)
            runs["rccl"] = runs[ref]            # (the reference leg, whichever transport it is)
            if runs["rccl"][0] and runs["peer"][0]:
                # the leg that runs first is timed on a chip whose clocks have not settled
                # (a block right after idle runs 4-19 % slow): RCCL gets a
                # second timing behind the peer leg and keeps its better one.  (Leg outcomes
                # are agreed between the ranks, so every rank takes this branch or none.)
                again = _trial_leg(eng, ref, ref_step, lr, x0, 30)
                if again[0]:
                    runs["rccl"] = (True, runs["rccl"][1], min(runs["rccl"][2], again[2]))
                else:
                    runs["rccl"] = runs[ref] = (False, None, float("inf"))
                    abort_if_failed(ref)
            if set_limit and runs["peer"][0]:
                set_limit(int(os.environ.get("BB_PEER_TIMEOUT_MS", "10000")))
            eng.set_coords(x0)
            scale = float(numpy.abs(x0).max() + 1e-30)
            # one step: 1e-4 (the transports add the ranks' partials in different orders); the
            # whole leg, 41 steps: 1e-3 -- far above what the order of a sum does over that
            # many steps in fp32 (1e-5), far below what a lost or torn partial does
            agree = bool(runs["rccl"][0] and runs["peer"][0]
                         and numpy.allclose(runs["rccl"][1][0], runs["peer"][1][0], rtol=1e-4,
                                            atol=1e-6 * scale)
                         and numpy.allclose(runs["rccl"][1][1], runs["peer"][1][1], rtol=1e-3,
                                            atol=1e-5 * scale))
            mine = (agree, runs["rccl"][2], runs["peer"][2])
            every = [None] * eng.world
            dist.all_gather_object(every, mine)
            t_rccl = max(e[1] for e in every)
            t_peer = max(e[2] for e in every)
            # the library communicator is the plain path: the peer exchange has to win by
            # more than the trial's own noise (3 %), not by a coin flip
            use_peer = all(e[0] for e in every) and t_peer < 0.97 * t_rccl
            ms = lambda t: t * 1e3 if numpy.isfinite(t) else None
            eng._comm_trial = {"agree": all(e[0] for e in every), "reference": ref,
                               "rccl_ms": ms(t_rccl),      # (the reference leg's time)
                               "peer_ms": ms(t_peer),
                               "error": getattr(eng, "_comm_trial_error", None)}
            if use_peer:
                state = "peer"
            elif runs["rccl"][0]:
                state = ref
            else:
                state = "torch"
        elif have_rccl:
            state = "rccl"
        else:
            state = "torch"
    eng._comm_state = state
    return state


def run_iterations(eng, n_iter, lr, world):
    """n_iter solver iterations on an engine whose inputs are set.

    world == 1: the whole loop is enqueued by one C call.  world > 1: per
    iteration, local partial gradient -> sum over ranks -> identical update on
    every rank, so the replicas of X stay identical; `select_exchange` picks the
    transport."""
    if world == 1:
        eng.iterate(n_iter, lr)
        return
    state = select_exchange(eng, lr)
    if state == "peer":
        eng.iterate_peer(n_iter, lr)
    elif state == "rccl":
        eng.iterate_dist(n_iter, lr)
    elif state == "torch":
        t = eng.exchange_tensor()
        for _ in range(n_iter):
            eng.grad()
            allreduce_exchange(t)
            eng.apply(lr)
    else:
        for _ in range(n_iter):
            eng.grad()
            allreduce_exchange_host(eng)
            eng.apply(lr)
