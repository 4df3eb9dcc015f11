"""Test-side binding of the CPU oracle (oracle/bb_oracle.c) and helpers that
build the synthetic inputs BASELINE.md section 3 describes.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg use this module."""
import ctypes
import os
import subprocess

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libbb_oracle.so")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

c_long, c_int, c_dbl = ctypes.c_long, ctypes.c_int, ctypes.c_double
p_dbl = ctypes.POINTER(c_dbl)
p_flt = ctypes.POINTER(ctypes.c_float)
p_i32 = ctypes.POINTER(ctypes.c_int32)


def _p(a):
    return a.ctypes.data_as(p_dbl)


class Oracle(object):
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.bbo_count_band_regions.restype = c_long
        L.bbo_count_band_regions.argtypes = [p_dbl, c_int, c_int, c_int]
        L.bbo_count_band_regions_rows.restype = c_long
        L.bbo_count_band_regions_rows.argtypes = [p_dbl, c_int, c_int, c_int, c_int, c_int]
        L.bbo_contactmap_scatter.restype = None
        L.bbo_contactmap_scatter.argtypes = [p_dbl, c_long, c_int, p_dbl, c_long]
        L.bbo_contactmap_normalize.restype = None
        L.bbo_contactmap_normalize.argtypes = [p_dbl, c_long, p_dbl, p_dbl]
        L.bbo_benjamini_hochberg.restype = None
        L.bbo_benjamini_hochberg.argtypes = [p_dbl, c_long, c_long, p_dbl]
        L.bbo_downsample.restype = None
        L.bbo_downsample.argtypes = [p_flt, c_long, p_flt, c_long]
        L.bbo_counts_to_wish.restype = None
        L.bbo_counts_to_wish.argtypes = [p_dbl, c_long, c_long, c_dbl, p_dbl, c_long]
        L.bbo_stress_grad.restype = c_dbl
        L.bbo_stress_grad.argtypes = [p_dbl, c_long, c_long, p_dbl, c_int, p_dbl]
        L.bbo_solve.restype = None
        L.bbo_solve.argtypes = [p_dbl, c_long, c_long, p_dbl, c_long, c_dbl, c_int, p_dbl, p_dbl]
        L.bbo_solve_momentum.restype = None
        L.bbo_solve_momentum.argtypes = [p_dbl, c_long, c_long, p_dbl, c_long, c_dbl, c_dbl,
                                         c_int, p_dbl, p_dbl]
        L.bbo_stress_grad_units.restype = c_dbl
        L.bbo_stress_grad_units.argtypes = [p_dbl, c_long, c_long, p_dbl, c_int, p_i32, p_i32,
                                            c_long, c_long, c_long, c_long, p_dbl]

    # K1
    def count_band_regions(self, regions, low=25000, high=10000000):
        r = numpy.ascontiguousarray(regions, dtype=numpy.float64)
        return int(self.lib.bbo_count_band_regions(_p(r), r.shape[0], low, high))

    def count_band_regions_rows(self, regions, i_begin, i_end, low=25000, high=10000000):
        r = numpy.ascontiguousarray(regions, dtype=numpy.float64)
        return int(self.lib.bbo_count_band_regions_rows(_p(r), r.shape[0], low, high,
                                                        int(i_begin), int(i_end)))

    # A2 / A3
    def contactmap_scatter(self, triples, resolution, n_bins):
        t = numpy.nan_to_num(numpy.asarray(triples, dtype=numpy.float64))
        cols = numpy.ascontiguousarray(t.T)
        d = n_bins + 1
        m = numpy.zeros((d, d))
        self.lib.bbo_contactmap_scatter(_p(cols), t.shape[0], int(resolution), _p(m), d)
        return m

    def contactmap_normalize(self, matrix, kr, krexp):
        m = numpy.ascontiguousarray(matrix, dtype=numpy.float64).copy()
        kr = numpy.ascontiguousarray(kr, dtype=numpy.float64)
        ke = numpy.ascontiguousarray(krexp, dtype=numpy.float64)
        self.lib.bbo_contactmap_normalize(_p(m), m.shape[0] - 1, _p(kr), _p(ke))
        return m

    def benjamini_hochberg(self, p, n):
        p = numpy.ascontiguousarray(p, dtype=numpy.float64)
        q = numpy.zeros_like(p)
        self.lib.bbo_benjamini_hochberg(_p(p), p.shape[0], int(n), _p(q))
        return q

    def downsample(self, yp1, yp5i):
        yp1 = numpy.ascontiguousarray(yp1, dtype=numpy.float32)
        out = numpy.ascontiguousarray(yp5i, dtype=numpy.float32).copy()
        self.lib.bbo_downsample(yp1.ctypes.data_as(p_flt), yp1.shape[0],
                                out.ctypes.data_as(p_flt), out.shape[0])
        return out

    # S0
    def counts_to_wish(self, counts, alpha=3.0):
        c = numpy.ascontiguousarray(counts, dtype=numpy.float64)
        w = numpy.zeros_like(c)
        self.lib.bbo_counts_to_wish(_p(c), c.shape[0], c.shape[1], float(alpha), _p(w), c.shape[1])
        return w

    def stress_grad(self, wish, X, f64=True):
        w = numpy.ascontiguousarray(wish, dtype=numpy.float64)
        X = numpy.ascontiguousarray(X, dtype=numpy.float64)
        g = numpy.zeros_like(X)
        s = self.lib.bbo_stress_grad(_p(w), w.shape[0], w.shape[1], _p(X), 1 if f64 else 0, _p(g))
        return float(s), g

    def solve(self, wish, X0, iters, lr, f64=True):
        w = numpy.ascontiguousarray(wish, dtype=numpy.float64)
        X = numpy.ascontiguousarray(X0, dtype=numpy.float64).copy()
        hist = numpy.zeros(iters)
        g = numpy.zeros_like(X)
        self.lib.bbo_solve(_p(w), w.shape[0], w.shape[1], _p(X), int(iters), float(lr),
                           1 if f64 else 0, _p(hist), _p(g))
        return X, hist

    def solve_momentum(self, wish, X0, iters, lr, mu, f64=True):
        w = numpy.ascontiguousarray(wish, dtype=numpy.float64)
        X = numpy.ascontiguousarray(X0, dtype=numpy.float64).copy()
        hist = numpy.zeros(iters)
        scratch = numpy.zeros((2 * X.shape[0], 3))
        self.lib.bbo_solve_momentum(_p(w), w.shape[0], w.shape[1], _p(X), int(iters), float(lr),
                                    float(mu), 1 if f64 else 0, _p(hist), _p(scratch))
        return X, hist

    def stress_grad_units(self, wish, X, tile_I, tile_J, upt, vw, u_begin, u_end, f64=True):
        w = numpy.ascontiguousarray(wish, dtype=numpy.float64)
        X = numpy.ascontiguousarray(X, dtype=numpy.float64)
        ti = numpy.ascontiguousarray(tile_I, dtype=numpy.int32)
        tj = numpy.ascontiguousarray(tile_J, dtype=numpy.int32)
        g = numpy.zeros_like(X)
        s = self.lib.bbo_stress_grad_units(_p(w), w.shape[0], w.shape[1], _p(X),
                                           1 if f64 else 0, ti.ctypes.data_as(p_i32),
                                           tj.ctypes.data_as(p_i32), int(upt), int(vw),
                                           int(u_begin), int(u_end), _p(g))
        return float(s), g


_cached = None


def load():
    global _cached
    if _cached is None:
        src = os.path.join(ORACLE_DIR, "bb_oracle.c")
        if (not os.path.exists(ORACLE_SO)
                or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src)):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
        _cached = Oracle(ctypes.CDLL(ORACLE_SO))
    return _cached


_cached_mt = None


def _load_mt():
    """oracle/libbb_oracle_mt.so (OpenMP); OSError / CalledProcessError without libgomp."""
    global _cached_mt
    if _cached_mt is None:
        so = os.path.join(ORACLE_DIR, "libbb_oracle_mt.so")
        src = os.path.join(ORACLE_DIR, "bb_oracle_mt.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libbb_oracle_mt.so"])
        lib = ctypes.CDLL(so)
        lib.bbo_solve_mt.restype = ctypes.c_int
        lib.bbo_solve_mt.argtypes = [p_dbl, c_long, c_long, p_dbl, c_long, c_dbl, ctypes.c_int,
                                     p_dbl, ctypes.c_int]
        lib.bbo_solve_momentum_mt.restype = ctypes.c_int
        lib.bbo_solve_momentum_mt.argtypes = [p_dbl, c_long, c_long, p_dbl, c_long, c_dbl, c_dbl,
                                              ctypes.c_int, p_dbl, ctypes.c_int]
        lib.bbo_solve_gen_mt.restype = ctypes.c_int
        lib.bbo_solve_gen_mt.argtypes = [p_dbl, c_long, p_i32, p_i32, c_long, c_long, p_dbl, c_long,
                                         c_dbl, c_dbl, ctypes.c_int, ctypes.c_int, c_dbl, p_dbl,
                                         ctypes.c_int]
        lib.bbo_solve_gen_steps_mt.restype = ctypes.c_int
        lib.bbo_solve_gen_steps_mt.argtypes = lib.bbo_solve_gen_mt.argtypes + [p_dbl]
        _cached_mt = lib
    return _cached_mt


def solve_mt(wish, X0, iters, lr, threads, f64=True):
    """bbo_solve on `threads` host cores (oracle/bb_oracle_mt.c, OpenMP): the
    multi-core cpu_baseline of bench.py.  Raises OSError if it cannot be built or
    loaded (no libgomp): callers fall back to the scalar oracle."""
    lib = _load_mt()
    w = numpy.ascontiguousarray(wish, dtype=numpy.float64)
    X = numpy.ascontiguousarray(X0, dtype=numpy.float64).copy()
    hist = numpy.zeros(iters)
    rc = lib.bbo_solve_mt(_p(w), w.shape[0], w.shape[1], _p(X), int(iters), float(lr),
                                 1 if f64 else 0, _p(hist), int(threads))
    if rc != 0:
        raise MemoryError("bbo_solve_mt: out of memory")
    return X, hist


def solve_momentum_mt(wish, X0, iters, lr, mu, threads, f64=True):
    """bbo_solve_momentum on `threads` host cores (mu = 0: plain steps)."""
    lib = _load_mt()
    w = numpy.ascontiguousarray(wish, dtype=numpy.float64)
    X = numpy.ascontiguousarray(X0, dtype=numpy.float64).copy()
    hist = numpy.zeros(iters)
    rc = lib.bbo_solve_momentum_mt(_p(w), w.shape[0], w.shape[1], _p(X), int(iters), float(lr),
                                   float(mu), 1 if f64 else 0, _p(hist), int(threads))
    if rc != 0:
        raise MemoryError("bbo_solve_momentum_mt: out of memory")
    return X, hist


def dense_tiles(n, vw=512):
    """Tile list of the dense upper triangle, device order (J, then I ascending)."""
    nb = -(-int(n) // vw)
    tj, ti = numpy.meshgrid(numpy.arange(nb), numpy.arange(nb))
    sel = ti <= tj
    order = numpy.lexsort((ti[sel], tj[sel]))
    return ti[sel][order].astype(numpy.int32), tj[sel][order].astype(numpy.int32)


def solve_gen_mt(xstar, X0, iters, lr, threads, tiles=None, vw=512, mu=0.0, f64=True,
                 delta_f32=None, blk_scale=None, bin_scale=None):
    """The solver loop with delta_ij = |x*_i - x*_j| formed on the fly over a tile list
    (None: the dense upper triangle) -- no matrix in memory, so N = 50,000 dense and
    BASELINE config 5 (N = 309,568 block-sparse) run on the host.  delta_f32 (default:
    not f64) rounds delta to float exactly as the device's fp32 pack does.  bin_scale: a
    step factor per bin (bb_solver_set_bin_steps); blk_scale: one per block of vw bins
    (bb_solver_set_block_steps)."""
    lib = _load_mt()
    xs = numpy.ascontiguousarray(xstar, dtype=numpy.float64)
    X = numpy.ascontiguousarray(X0, dtype=numpy.float64).copy()
    n = xs.shape[0]
    ti, tj = dense_tiles(n, vw) if tiles is None else tiles
    ti = numpy.ascontiguousarray(ti, dtype=numpy.int32)
    tj = numpy.ascontiguousarray(tj, dtype=numpy.int32)
    if delta_f32 is None:
        delta_f32 = not f64
    hist = numpy.zeros(iters)
    args = (_p(xs), n, ti.ctypes.data_as(p_i32), tj.ctypes.data_as(p_i32),
            ti.shape[0], int(vw), _p(X), int(iters), float(lr), float(mu),
            1 if f64 else 0, 1 if delta_f32 else 0,
            1e-30 if delta_f32 else 1e-290, _p(hist), int(threads))
    if blk_scale is not None:
        assert numpy.shape(blk_scale) == (-(-n // int(vw)),)
        bin_scale = numpy.repeat(numpy.asarray(blk_scale, dtype=numpy.float64), int(vw))[:n]
    if bin_scale is None:
        rc = lib.bbo_solve_gen_mt(*args)
    else:
        sc = numpy.ascontiguousarray(bin_scale, dtype=numpy.float64)
        assert sc.shape == (n,)
        rc = lib.bbo_solve_gen_steps_mt(*(args + (_p(sc),)))
    if rc != 0:
        raise MemoryError("bbo_solve_gen_mt: out of memory")
    return X, hist


def golden(name):
    return numpy.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


# -- synthetic inputs of BASELINE.md section 3 ------------------------------
def random_walk(n, seed=0):
    """Ground truth X*: 3-D Gaussian random walk, centred."""
    x = numpy.cumsum(numpy.random.default_rng(seed).standard_normal((n, 3)), axis=0)
    return x - x.mean(axis=0)


def wish_from_coords(x):
    d = x[:, None, :] - x[None, :, :]
    return numpy.sqrt((d * d).sum(-1))


def noisy_init(xstar, seed=1, scale=0.5):
    return xstar + scale * numpy.random.default_rng(seed).standard_normal(xstar.shape)
