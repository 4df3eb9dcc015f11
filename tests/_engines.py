"""Test-only solver engine that plays one rank with the CPU oracle.

It exists so that the multi-rank orchestration in blueberry_amd.solver
(partition of the packed units over ranks, exchange-buffer layout, one
all-reduce per iteration, identical update on every rank) can be rehearsed with
world_size > 1 on a machine without GPUs.  The partition itself comes from the
PRODUCT library's host-only layout functions (bb_layout_*), so what is tested
is the real unit ranges.  Never imported by the package."""
import numpy

from blueberry_amd import _lib
from tests import _oracle


class OracleEngine(object):
    def __init__(self, n_bins, dtype, rank=0, world=1, device=0, tiles=None):
        lib = _lib.load()
        self.n_bins, self.dtype, self.rank, self.world = n_bins, dtype, rank, world
        code = _lib.BB_F32 if dtype == "float32" else _lib.BB_F64
        self.info = _lib.LayoutInfo()
        _lib.check(lib.bb_layout_dense_info(n_bins, code, self.info))
        self.ti = numpy.zeros(self.info.n_tiles, dtype=numpy.int32)
        self.tj = numpy.zeros(self.info.n_tiles, dtype=numpy.int32)
        _lib.check(lib.bb_layout_dense_tiles(n_bins, code, self.ti.ctypes.data_as(_lib.p_i32),
                                             self.tj.ctypes.data_as(_lib.p_i32),
                                             self.info.n_tiles))
        ub, ue = _lib.c_i64(), _lib.c_i64()
        _lib.check(lib.bb_layout_rank_units(self.info.n_units, rank, world, ub, ue))
        self.u_begin, self.u_end = int(ub.value), int(ue.value)
        self.oracle = _oracle.load()
        self.tiles_arg = tiles
        self.device = device
        self.exch = numpy.zeros(3 * self.info.n_pad + 2)
        self.hist = []
        self.mu = 0.0

    def close(self):
        pass

    def set_wish_dense(self, matrix, kind, alpha):
        m = numpy.ascontiguousarray(matrix, dtype=numpy.float64)
        self.w = self.oracle.counts_to_wish(m, alpha) if kind == "counts" else m

    def set_momentum(self, mu):
        self.mu = float(mu)

    def set_block_steps(self, scale):
        self.set_bin_steps(None if scale is None else
                           numpy.repeat(numpy.asarray(scale, dtype=numpy.float64), self.info.vw)[:self.n_bins])

    def set_bin_steps(self, scale):
        self.bin_scale = None if scale is None else numpy.asarray(scale, dtype=numpy.float64)

    def _own_pairs(self):
        """Mask of the pairs i < j this rank's units hold."""
        n, vw, upt = self.n_bins, self.info.vw, self.info.units_per_tile
        rpu = vw // upt
        own = numpy.zeros((n, n), dtype=bool)
        for u in range(self.u_begin, self.u_end):
            t, sub = divmod(u, upt)
            i0, j0 = int(self.ti[t]) * vw + sub * rpu, int(self.tj[t]) * vw
            own[i0:min(i0 + rpu, n), j0:min(j0 + vw, n)] = True
        return numpy.triu(own, 1)

    def degrees(self):
        if getattr(self, "maps", None):
            deg = numpy.zeros(self.n_bins, dtype=numpy.int64)
            for a, w, _ in self.maps:
                if w is not None:
                    deg[a:a + w.shape[0]] = (numpy.triu(w, 1) > 0).sum(axis=0) + (numpy.triu(w, 1) > 0).sum(axis=1)
            return deg
        on = self._own_pairs() & (self.w > 0)
        return (on.sum(axis=0) + on.sum(axis=1)).astype(numpy.int64)

    def set_coords(self, x0):
        self.X = numpy.ascontiguousarray(x0, dtype=numpy.float64).copy()
        self.V = numpy.zeros_like(self.X)
        self.hist = []

    def get_coords(self):
        return self.X.copy()

    def grad(self):
        s, g = self.oracle.stress_grad_units(self.w, self.X, self.ti, self.tj,
                                             self.info.units_per_tile, self.info.vw,
                                             self.u_begin, self.u_end)
        if getattr(self, "bin_scale", None) is not None:     # scaled where it leaves the sum
            g = g * self.bin_scale[:, None]
        self.exch[:] = 0
        self.exch[:3 * self.n_bins] = g.ravel()
        self.exch[-2] = s

    def read_exchange(self):
        return self.exch.copy()

    def write_exchange(self, host):
        self.exch[:] = host

    def apply(self, lr):
        self.V = self.mu * self.V - lr * self.exch[:3 * self.n_bins].reshape(self.n_bins, 3)
        self.X += self.V
        self.hist.append(self.exch[-2] + self.exch[-1])

    def iterate(self, iters, lr):
        if getattr(self, "maps", None):
            return self._iterate_maps(iters, lr)
        for _ in range(iters):
            self.grad()
            self.apply(lr)

    def stress_history(self):
        return numpy.array(self.hist)

    # -- several maps in one solver (the host logic of StructureSolver.fit_many on CPU) --------
    def set_maps(self, bin_begin, lr_scale):
        assert self.world == 1
        self.maps = [(int(a), None, float(s)) for a, s in zip(bin_begin[:-1], lr_scale)]
        self.n_maps = len(self.maps)
        vw = self.info.vw
        assert all(a % vw == 0 for a, _, _ in self.maps) and int(bin_begin[-1]) == self.n_bins
        # what bb_solver_create / bb_solver_set_maps check of the tile list the host wrote:
        # strictly ordered by (J, I), I <= J, and no tile joins two maps
        ti, tj = (numpy.asarray(t, dtype=numpy.int64) for t in self.tiles_arg)
        key = tj * self.info.n_blocks + ti
        assert (ti <= tj).all() and (numpy.diff(key) > 0).all() and tj.max() < self.info.n_blocks
        starts = numpy.asarray([a for a, _, _ in self.maps]) // vw
        assert (numpy.searchsorted(starts, ti, side="right") ==
                numpy.searchsorted(starts, tj, side="right")).all()

    def set_wish_dense_block(self, matrix, bin_offset, kind, alpha):
        m = numpy.ascontiguousarray(matrix, dtype=numpy.float64)
        w = self.oracle.counts_to_wish(m, alpha) if kind == "counts" else m
        for q, (a, _, s) in enumerate(self.maps):
            if a == int(bin_offset):
                self.maps[q] = (a, w, s)
                return
        raise ValueError("no map starts at bin %d" % bin_offset)

    def _iterate_maps(self, iters, lr):
        for _ in range(iters):
            row = []
            for a, w, scale in self.maps:
                n = w.shape[0]
                s, g = self.oracle.stress_grad(w, self.X[a:a + n], f64=self.dtype == "float64")
                if getattr(self, "bin_scale", None) is not None:      # these replace the maps' steps
                    g, scale = g * self.bin_scale[a:a + n, None], 1.0
                self.V[a:a + n] = self.mu * self.V[a:a + n] - lr * scale * g
                self.X[a:a + n] += self.V[a:a + n]
                row.append(s)
            self.hist.extend(row)

    def matvec_sq(self, x):
        """(D o D) @ x restricted to the pairs this rank's units own (numpy)."""
        a = numpy.where(self._own_pairs(), self.w * self.w, 0.0)
        x = numpy.asarray(x, dtype=numpy.float64)
        return a @ x + a.T @ x


class ScriptedRankEngine(object):
    """One rank of a multi-rank job with no compute at all: what select_exchange's
    trial sees of an engine, with the "RCCL" transport played by an ASYNC all-reduce on
    a process group of its own (enqueue returns at once, like ncclAllReduce; the wait
    happens in sync / sync_timeout) and the peer transport by nothing.  `fail_leg` makes
    this rank raise at the start of that transport's first step, before it has joined
    the collective."""

    def __init__(self, rank, world, data_group, fail_leg=None):
        self.rank, self.world, self.group, self.fail_leg = rank, world, data_group, fail_leg
        self.x = numpy.arange(12.0).reshape(4, 3)
        self.pending = []
        self.aborted = False
        self._comm_state = self._comm_trial = None
        self._peer_error = ""
        self.peer_failed = False

    def comm_setup(self):
        return True

    def peer_setup(self):
        return True

    def peer_set_timeout(self, ms):
        pass

    def get_coords(self):
        return self.x.copy()

    def set_coords(self, x):
        self.sync()
        self.x = numpy.array(x, dtype=float)

    def iterate_dist(self, k, lr):
        import torch
        import torch.distributed as dist
        if self.fail_leg == "rccl":
            raise RuntimeError("scripted: rccl leg fails on this rank before the collective")
        for _ in range(k):
            t = torch.ones(4)
            self.pending.append(dist.all_reduce(t, group=self.group, async_op=True))
            self.x = self.x - lr

    def iterate_peer(self, k, lr):
        if self.fail_leg == "peer":
            self.peer_failed = True
            return
        self.x = self.x - lr * k

    def peer_status(self):
        if self.peer_failed:
            raise RuntimeError("scripted: peer leg: time limit")
        return 0

    def sync(self):
        self.sync_timeout(60000)

    def sync_timeout(self, ms):
        import datetime
        work, self.pending = self.pending, []
        for w in work:
            if not w.wait(timeout=datetime.timedelta(milliseconds=ms)):
                raise RuntimeError("scripted: stream did not drain")

    def comm_abort(self):
        self.pending = []
        self.aborted = True
