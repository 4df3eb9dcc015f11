"""CPU: host-side logic, the C-ABI surface, and loud failure without a GPU."""
import os
import re

import numpy
import pytest

import blueberry_amd as bb
from blueberry_amd import _lib
from blueberry_amd.band import band_row_share

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def have_gpu():
    n = _lib.ctypes.c_int(0)
    return _lib.load().bb_device_count(n) == _lib.BB_OK and n.value > 0


# ---- the C-ABI ---------------------------------------------------------------
def header_symbols():
    text = open(os.path.join(ROOT, "include", "blueberry_hip.h")).read()
    return sorted(set(re.findall(r"BB_API\s+[\w\s\*]+?\b(bb_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 30
    for name in names:
        assert getattr(lib, name) is not None, name
    # and the ctypes table binds exactly the header's surface
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.bb_version() == 100
    assert lib.bb_layout_dense_info(0, _lib.BB_F32, _lib.LayoutInfo()) == _lib.BB_ERR_INVALID
    assert "n_bins" in _lib.last_error()


def test_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "blueberry_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "libbb_oracle" not in src and "bbo_" not in src, f
                assert not re.search(r"^\s*(from|import)\s+(tests|oracle)\b", src, re.M), f


@pytest.mark.skipif(have_gpu(), reason="checks the no-GPU failure mode")
def test_compute_fails_loudly_without_gpu():
    with pytest.raises(RuntimeError, match="HIP device"):
        bb.count_band_regions(numpy.arange(10.0))
    with pytest.raises(RuntimeError, match="HIP device"):
        bb.StructureSolver(n_iter=1).fit(numpy.ones((4, 4)))
    cm = bb.ContactMap.from_matrix(numpy.ones((3, 3)), KRnorm=numpy.ones(2),
                                   KRexpected=numpy.ones(2))
    with pytest.raises(RuntimeError, match="HIP device"):
        cm.normalize()
    with pytest.raises(RuntimeError, match="HIP device"):
        cm.filter()                      # filter runs on the device too: no numpy stand-in
    with pytest.raises(RuntimeError, match="HIP device"):
        cm.correlation()
    with pytest.raises(RuntimeError, match="HIP device"):
        cm.eigenvector()
    with pytest.raises(RuntimeError, match="HIP device"):
        bb.ContactMap.from_triples(numpy.array([[0.0, 5000.0, 3.0]]), 5000, 4)
    # the round-4 entry points: several maps in one solver, triples resident on the device
    with pytest.raises(RuntimeError, match="HIP device"):
        bb.StructureSolver(n_iter=1).fit_many([numpy.ones((4, 4)), numpy.ones((5, 5))])
    with pytest.raises(RuntimeError, match="HIP device"):
        bb.StructureSolver(n_iter=1).fit_triples(numpy.array([[0.0, 5000.0, 3.0]]), 5000, 4)
    with pytest.raises(ValueError):      # (argument checks come before the device is needed)
        bb.StructureSolver(n_iter=1).fit_triples(numpy.zeros((3, 2)), 5000, 4)
    assert cm.matrix.shape == (3, 3)     # the host copy was never lost
    n = _lib.ctypes.c_int(5)
    assert _lib.load().bb_device_count(n) == _lib.BB_ERR_HIP and n.value == 0


# ---- layout ------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [_lib.BB_F32, _lib.BB_F64])
@pytest.mark.parametrize("n", [1, 2, 127, 128, 129, 511, 512, 513, 963, 4096, 4097, 24926, 50000,
                               309568])
def test_dense_layout(dtype, n):
    lib = _lib.load()
    info = _lib.LayoutInfo()
    _lib.check(lib.bb_layout_dense_info(n, dtype, info))
    # fp32: 4 rows x 512 columns per unit; fp64: 8 x 128 up to 4096 bins, 2 x 512 above
    vw, rpu = (512, 4) if dtype == _lib.BB_F32 else ((128, 8) if n <= 4096 else (512, 2))
    assert info.vw == vw and info.rows_per_unit == rpu and info.units_per_tile == vw // rpu
    assert info.n_pad % vw == 0 and 0 <= info.n_pad - n < vw
    nb = info.n_pad // vw
    assert info.n_blocks == nb and info.n_tiles == nb * (nb + 1) // 2
    assert info.n_units * 8192 == info.n_tiles * vw * vw * (4 if dtype == _lib.BB_F32 else 8)
    if info.n_tiles <= 20000:
        ti = numpy.zeros(info.n_tiles, dtype=numpy.int32)
        tj = numpy.zeros(info.n_tiles, dtype=numpy.int32)
        _lib.check(lib.bb_layout_dense_tiles(n, dtype, ti.ctypes.data_as(_lib.p_i32),
                                             tj.ctypes.data_as(_lib.p_i32), info.n_tiles))
        assert (ti <= tj).all() and tj.max() == nb - 1
        key = tj.astype(numpy.int64) * nb + ti          # strictly increasing: (J, I) order
        assert (numpy.diff(key) > 0).all()
        assert lib.bb_layout_dense_tiles(n, dtype, ti.ctypes.data_as(_lib.p_i32),
                                         tj.ctypes.data_as(_lib.p_i32),
                                         info.n_tiles - 1) == _lib.BB_ERR_INVALID


@pytest.mark.parametrize("n_units,world", [(0, 1), (1, 8), (7, 8), (620928, 8), (10**12, 7)])
def test_rank_units_partition(n_units, world):
    lib = _lib.load()
    prev = 0
    sizes = []
    for r in range(world):
        a, b = _lib.c_i64(), _lib.c_i64()
        _lib.check(lib.bb_layout_rank_units(n_units, r, world, a, b))
        assert a.value == prev and b.value >= a.value
        sizes.append(b.value - a.value)
        prev = b.value
    assert prev == n_units and max(sizes) - min(sizes) <= 1
    assert lib.bb_layout_rank_units(10, 3, 3, a, b) == _lib.BB_ERR_INVALID


@pytest.mark.parametrize("n,world", [(0, 2), (1, 2), (1000, 2), (1000, 8), (50000, 8), (7, 8)])
def test_band_row_share_partition(n, world):
    prev, areas = 0, []
    for r in range(world):
        a, b = band_row_share(n, r, world)
        assert a == prev and b >= a
        areas.append(sum(range(a, b)))
        prev = b
    assert prev == n
    if n >= 1000:       # triangle cut into bands of (nearly) equal pair counts
        assert max(areas) < 1.1 * (sum(areas) / world) + n


# ---- host-side argument handling -----------------------------------------------
def test_band_count_device_default(monkeypatch):
    """distributed=True must not default every rank to cuda:0."""
    from blueberry_amd.band import pick_device
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert pick_device(None, True) == 5 and pick_device(None, False) == 0
    assert pick_device(2, True) == 2 and pick_device(0, True) == 0
    monkeypatch.delenv("LOCAL_RANK")
    assert pick_device(None, True) == 0


def test_solver_argument_validation():
    for kw in ({"dtype": "float16"}, {"kind": "p"}, {"n_iter": -1}, {"lr": 0}, {"lr": -1.0},
               {"alpha": 0}):
        with pytest.raises(ValueError):
            bb.StructureSolver(**kw)
    s = bb.StructureSolver()
    for bad in (numpy.zeros((3, 4)), numpy.zeros(5), numpy.zeros((1, 1))):
        with pytest.raises(ValueError):
            s.fit(bad)


def test_solver_with_injected_engine_single_rank(oracle):
    """The host loop (auto lr, default seeded init, engine protocol) against the
    oracle's own K-step solve."""
    from tests._engines import OracleEngine
    from tests import _oracle
    n = 150
    w = _oracle.wish_from_coords(_oracle.random_walk(n))
    s = bb.StructureSolver(n_iter=7, dtype="float64", kind="wish", seed=3, distributed=False,
                           engine=OracleEngine).fit(w)
    x0 = numpy.random.default_rng(3).standard_normal((n, 3))
    X_ref, h_ref = oracle.solve(w, x0, 7, 1.0 / (2 * n))
    assert s.lr_ == 1.0 / (2 * n) and s.n_bins_ == n
    assert numpy.abs(s.structure_ - X_ref).max() < 1e-12 * numpy.abs(X_ref).max()
    assert numpy.abs(s.stress_ / h_ref - 1).max() < 1e-12
    assert numpy.array_equal(bb.StructureSolver(n_iter=7, dtype="float64", kind="wish", seed=3,
                                                distributed=False, engine=OracleEngine)
                             .fit_transform(w), s.structure_)


def test_spectral_init_host_loop_is_classical_mds():
    from tests._engines import OracleEngine
    from tests import _oracle
    n = 200
    xs = _oracle.random_walk(n)
    w = _oracle.wish_from_coords(xs)
    s = bb.StructureSolver(n_iter=1, dtype="float64", kind="wish", init="spectral",
                           distributed=False, engine=OracleEngine).fit(w)
    assert numpy.abs(_oracle.wish_from_coords(s.structure_) - w).max() < 1e-6 * w.max()
    assert s.stress_[0] < 1e-10 * (w ** 2).sum()
    # the signs of the Ritz vectors are pinned (to the side of the start's first column), so
    # the start is one configuration and not one of its eight mirror images
    from blueberry_amd.solver import spectral_init
    eng = OracleEngine(n, "float64")
    eng.set_wish_dense(w, "wish", 3.0)
    x0 = spectral_init(eng, n, 1, seed=0)
    p = numpy.random.default_rng(0).standard_normal((n, 3))[:, 0]
    assert (x0.T @ p > 0).all()


def test_spectral_start_stopping_rule_host_form():
    """spectral_tol: a complete noise-free map has a rank-3 B, so the second product already
    lies in span(V) and the loop ends after ONE orthonormalised product with the start the
    fixed 40 give; tol = 0 makes exactly spectral_iter; an incomplete map takes more products
    the tighter the tolerance."""
    from tests._engines import OracleEngine
    from tests import _oracle
    from blueberry_amd.solver import spectral_init
    n = 150
    w = _oracle.wish_from_coords(_oracle.random_walk(n))
    eng = OracleEngine(n, "float64")
    eng.set_wish_dense(w, "wish", 3.0)
    x40 = spectral_init(eng, n, 1, seed=0)
    xt, done = spectral_init(eng, n, 1, seed=0, tol=1e-3, return_iterations=True)
    assert done == 1 and numpy.abs(xt - x40).max() < 1e-8 * numpy.abs(x40).max()
    assert spectral_init(eng, n, 1, n_iter=7, seed=0, return_iterations=True)[1] == 7
    hole = numpy.triu(numpy.random.default_rng(5).random((n, n)) < 0.1, 1)
    wm = w.copy()
    wm[hole | hole.T] = 0.0
    eng.set_wish_dense(wm, "wish", 3.0)
    counts = [spectral_init(eng, n, 1, n_iter=80, seed=0, tol=t, return_iterations=True)[1]
              for t in (1e-1, 3e-2, 1e-2, 1e-9)]
    assert 1 < counts[0] < counts[1] < counts[2] < counts[3] == 80
    s = bb.StructureSolver(n_iter=1, dtype="float64", kind="wish", init="spectral",
                           distributed=False, engine=OracleEngine, spectral_tol=3e-2,
                           spectral_iter=80).fit(wm)
    assert s.spectral_iterations_ == counts[1]
    with pytest.raises(ValueError):
        bb.StructureSolver(spectral_tol=1.5)
    with pytest.raises(ValueError):
        bb.StructureSolver(spectral_iter=-1)


@pytest.mark.parametrize("n", [2, 3])
def test_spectral_init_on_two_and_three_bins(n):
    """numpy's QR of a (2, 3) start has 2 columns; the host loop pads the missing
    direction with zeros, so fit(init='spectral') runs for every n >= 2 and still recovers
    the pairwise distances (2 or 3 points always embed exactly)."""
    from tests._engines import OracleEngine
    from tests import _oracle
    xs = numpy.random.default_rng(4).standard_normal((n, 3)) * 3.0
    w = _oracle.wish_from_coords(xs)
    s = bb.StructureSolver(n_iter=2, dtype="float64", kind="wish", init="spectral",
                           distributed=False, engine=OracleEngine).fit(w)
    assert s.structure_.shape == (n, 3) and numpy.isfinite(s.structure_).all()
    assert numpy.abs(_oracle.wish_from_coords(s.structure_) - w).max() < 1e-9 * w.max()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_fit_many_host_logic_with_the_oracle_engine(oracle, dtype):
    """StructureSolver.fit_many's own work -- the joint layout (every map at a multiple of the
    tile edge: 128 for a small fp64 job, else 512), the tile list it hands to the engine
    (strictly ordered, no tile joining two maps: checked by the test engine as
    bb_solver_set_maps checks it), per-map steps, starts and results -- with each map's
    compute played by the oracle: every map must come out as the oracle's solve of that map."""
    from tests._engines import OracleEngine
    from tests import _oracle
    sizes = [130, 700, 513, 2]
    mats = [_oracle.wish_from_coords(_oracle.random_walk(n, seed=q)) + (0 if n > 2 else 1.0 - numpy.eye(2))
            for q, n in enumerate(sizes)]
    x0 = [_oracle.noisy_init(_oracle.random_walk(n, seed=q), seed=9) for q, n in enumerate(sizes)]
    s = bb.StructureSolver(n_iter=4, dtype=dtype, kind="wish", engine=OracleEngine,
                           distributed=False, momentum=0.3).fit_many(mats, inits=x0)
    assert s.n_bins_many_ == sizes and s.n_iter_ == 4
    for q, n in enumerate(sizes):
        X, h = oracle.solve_momentum(mats[q], x0[q], 4, 1.0 / (2 * n), 0.3, f64=dtype == "float64")
        assert numpy.array_equal(s.structures_[q], X) and numpy.array_equal(s.stresses_[q], h)
    # default starts are fit()'s; a ContactMap is taken by its matrix
    a = bb.StructureSolver(n_iter=2, dtype=dtype, kind="wish", seed=5, engine=OracleEngine,
                           distributed=False).fit_many([mats[0], bb.ContactMap.from_matrix(mats[1])])
    for q in (0, 1):
        one = bb.StructureSolver(n_iter=2, dtype=dtype, kind="wish", seed=5, engine=OracleEngine,
                                 distributed=False).fit(mats[q])
        assert numpy.array_equal(a.structures_[q], one.structure_)
    # degree_steps: per-bin factors replace the per-map steps; map by map the single fit()
    holes = []
    for q, m in enumerate(mats[:3]):
        keep = numpy.triu(numpy.random.default_rng(q).random(m.shape) < 0.3, 1)
        keep |= numpy.triu(numpy.ones(m.shape, dtype=bool), 1) & ~numpy.triu(numpy.ones(m.shape, dtype=bool), 2)
        holes.append(numpy.where(keep | keep.T, m, 0.0))
    d = bb.StructureSolver(n_iter=3, dtype=dtype, kind="wish", engine=OracleEngine, distributed=False,
                           degree_steps=True).fit_many(holes, inits=x0[:3])
    for q, m in enumerate(holes):
        one = bb.StructureSolver(n_iter=3, dtype=dtype, kind="wish", engine=OracleEngine,
                                 distributed=False, degree_steps=True).fit(m, init=x0[q])
        assert d.lrs_[q] == one.lr_ == 1.0 / (2 * ((m > 0).sum(axis=0).max() + 1))
        assert numpy.abs(d.stresses_[q] / one.stress_ - 1).max() < 1e-12
        assert numpy.abs(d.structures_[q] - one.structure_).max() < 1e-12 * numpy.abs(one.structure_).max()
    with pytest.raises(ValueError):
        bb.StructureSolver(engine=OracleEngine, distributed=False).fit_many(mats, inits=x0[:2])
    with pytest.raises(ValueError):
        bb.StructureSolver(engine=OracleEngine, distributed=False).fit_many([numpy.zeros((1, 1))])


def test_count_band_regions_input_checks():
    with pytest.raises(ValueError):
        bb.band._as_regions(numpy.zeros((2, 2)))
    r = bb.band._as_regions([1, 2, 3])
    assert r.dtype == numpy.float64 and r.flags.c_contiguous
    assert (bb.LOW_FITHIC_CUTOFF, bb.HIGH_FITHIC_CUTOFF) == (25000, 10000000)
    assert (bb.Q_LOWER_BOUND, bb.Q_UPPER_BOUND) == (0.01, 0.50)


# ---- ContactMap host logic (no GPU needed) ---------------------------------------
def test_contactmap_from_arrays_and_filter():
    res = 5000
    mids = lambda b: b * res + res / 2.0
    contacts = numpy.array([[mids(0), mids(1), 5.0], [mids(1), mids(3), 7.0],
                            [mids(0), mids(1), 6.0]])          # later row wins
    cm = bb.ContactMap.from_arrays("GM12878_combined", 21, res, contacts, n_bins=5)
    assert cm.matrix.shape == (6, 6) and cm.n_bins == 5 and cm.resolution == res
    assert cm.matrix[0, 1] == cm.matrix[1, 0] == 6.0 and cm.matrix[1, 3] == 7.0
    assert cm.celltype == "GM12878_combined" and cm.chromosome == 21
    assert numpy.array_equal(cm.regions, [mids(0), mids(1), mids(3)])
    one_sided = bb.ContactMap.from_arrays("x", 1, res, contacts, n_bins=5, symmetric=False)
    assert one_sided.matrix[1, 0] == 0.0 and one_sided.matrix[0, 1] == 6.0
    with pytest.raises(ValueError):
        bb.ContactMap.from_arrays("x", 1, res, numpy.array([[mids(9), mids(1), 1.0]]), n_bins=5)
    # the matrix attribute: assignment replaces it, to_host() copies it
    cm.matrix = numpy.eye(6)
    assert cm.shape == (6, 6) and not cm.is_resident
    c = cm.to_host()
    c[0, 0] = 9.0
    assert cm.matrix[0, 0] == 1.0


def test_contactmap_normalize_precheck_and_shapes():
    cm = bb.ContactMap.from_matrix(numpy.ones((4, 4)), KRnorm=numpy.array([1.0, 0.0, 1.0]),
                                   KRexpected=numpy.ones(3))
    with pytest.raises(ZeroDivisionError):
        cm.normalize()                      # same exception type as the reference raises
    with pytest.raises(ValueError):
        bb.ContactMap.from_matrix(numpy.ones((4, 4))).normalize()
    with pytest.raises(ValueError):
        bb.ContactMap.from_matrix(numpy.ones((3, 4)))


# ---- FithicContactMap (datatypes.pyx:274-388) against the real reference ----------
def test_fithic_map_matches_reference_golden(tmp_path):
    import gzip
    from tests import _oracle
    z = _oracle.golden("fithic_map")
    res, n_bins = int(z["fh_resolution"]), int(z["fh_n_bins"])
    # through a file, like the reference's constructor
    path = tmp_path / ("x.chr7.spline_pass1.res%d.significances.txt.gz" % res)
    with gzip.open(str(path), "wt") as fh:
        fh.write("chr1\tfragmentMid1\tchr2\tfragmentMid2\tcontactCount\tp-value\tq-value\n")
        for m1, m2, c, p, q in z["fh_map"]:
            fh.write("7\t%d\t7\t%d\t%d\t%.17e\t%.17e\n" % (m1, m2, c, p, q))
    old = bb.datatypes.DATA_DIR
    bb.datatypes.DATA_DIR = str(tmp_path / "{0}.chr{1}.spline_pass1.res{2}.significances.txt.gz")
    try:
        fm = bb.FithicContactMap("x", 7, res)
    finally:
        bb.datatypes.DATA_DIR = old
    # the same pandas C parser reads both; re-printed text may round-trip 1 ulp off
    assert numpy.allclose(fm.map, z["fh_map"], rtol=4e-16, atol=0)
    assert numpy.array_equal(fm.regions, z["fh_regions"])
    fm = bb.FithicContactMap.from_array(z["fh_map"], res, "x", 7)      # exact from here on
    assert numpy.array_equal(fm.regions, z["fh_regions"])
    assert numpy.array_equal(fm.contacts(), z["fh_contacts"])
    for stat in ("count", "p", "q"):
        assert numpy.array_equal(fm.to_matrix(stat, n_bins=n_bins), z["fh_matrix_" + stat])
        sp = fm.to_sparse(stat, n_bins=n_bins)
        assert numpy.array_equal(sp.toarray(), z["fh_matrix_" + stat])
    with pytest.raises(ValueError):
        fm.to_matrix("z", n_bins=n_bins)


def test_fithic_decimate_py2_semantics():
    # two 1 kb contacts that fall in the same 5 kb pair of bins, one that does not
    m = numpy.array([[500.0, 6500.0, 3, 0.1, 0.5],
                     [1500.0, 7500.0, 4, 0.2, 0.3],
                     [500.0, 11500.0, 5, 0.5, 1.0]])
    fm = bb.FithicContactMap.from_array(m, 1000)
    assert fm.decimate(5000) is None and fm.resolution == 5000
    # (int + 5000) // 5000 * 5000 - 2500: 500 -> 2500, 1500 -> 2500, 6500/7500 -> 7500, 11500 -> 12500
    want = numpy.array([[2500.0, 7500.0, 7.0, 0.1 * 0.2, 0.3], [2500.0, 12500.0, 5.0, 0.5, 1.0]])
    assert numpy.allclose(fm.map, want) and numpy.array_equal(fm.regions, [2500.0, 7500.0, 12500.0])


def test_fithic_decimate_equals_the_reference_class():
    """`FithicContactMap.decimate` against the REAL class (`datatypes.pyx:317-339`), captured
    by tests/golden/make_golden.py on a map whose midpoints are exact multiples of the target
    resolution (there the line's Python 2 integer division and Python 3's true division
    agree): 130 rows over 38 bin pairs -- counts summed, p-values multiplied in file order,
    q-values minimised, rows in the order of first occurrence.  Bit for bit."""
    from tests import _oracle
    z = _oracle.golden("fithic_decimate")
    fm = bb.FithicContactMap.from_array(z["dec_in_map"], int(z["dec_in_resolution"]))
    assert fm.decimate(int(z["dec_resolution"])) is None
    assert fm.resolution == int(z["dec_resolution_attr"])
    assert numpy.array_equal(fm.map, z["dec_map"])
    assert numpy.array_equal(fm.regions, z["dec_regions"])


def test_early_stop_with_tolerance():
    from tests._engines import OracleEngine
    from tests import _oracle
    n = 120
    xs = _oracle.random_walk(n)
    w = _oracle.wish_from_coords(xs)
    kw = dict(dtype="float64", kind="wish", distributed=False, engine=OracleEngine)
    full = bb.StructureSolver(n_iter=200, **kw).fit(w, init=_oracle.noisy_init(xs))
    early = bb.StructureSolver(n_iter=200, tol=0.1, check_every=5, **kw).fit(
        w, init=_oracle.noisy_init(xs))
    assert full.n_iter_ == 200 and 5 <= early.n_iter_ < 200 and early.n_iter_ % 5 == 0
    assert numpy.array_equal(early.stress_, full.stress_[:early.n_iter_])
    with pytest.raises(ValueError):
        bb.StructureSolver(tol=0)


def test_tiles_from_entries_and_plot_smoke():
    from blueberry_amd.solver import tiles_from_entries
    ti, tj = tiles_from_entries(1300, [0, 1299, 600, 5], [1299, 0, 700, 5], "float64")
    # fp64, 1300 bins: vw = 128: pairs (0,1299)->tile (0,10) once, (600,700)->(4,5),
    # (5,5)->(0,0); order (J, I)
    assert list(zip(ti.tolist(), tj.tolist())) == [(0, 0), (4, 5), (0, 10)]
    ti, tj = tiles_from_entries(5000, [0, 4999, 600], [4999, 0, 700], "float64")   # wide: 512
    assert list(zip(ti.tolist(), tj.tolist())) == [(1, 1), (0, 9)]
    ti32, tj32 = tiles_from_entries(1300, [0, 1299, 600, 5], [1299, 0, 700, 5], "float32")
    assert list(zip(ti32.tolist(), tj32.tolist())) == [(0, 0), (1, 1), (0, 2)]
    with pytest.raises(ValueError):
        tiles_from_entries(10, [0], [10], "float32")
    import matplotlib
    matplotlib.use("Agg")
    cm = bb.ContactMap.from_matrix(numpy.arange(16.0).reshape(4, 4), resolution=5000,
                                   celltype="K562", chromosome=3)
    cm.plot(arcsinh=True)
    cm.plot(arcsinh=False, cmap="Reds")


def test_bench_cli_contract():
    """bench.py takes the driver's flags; `--gpus N` without a launcher starts its own N
    ranks as child processes (never re-executing a process that holds a GPU) and relays
    their exit code -- here, without a GPU, the ranks fail loudly and so does bench.py."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in r.stdout
    if have_gpu():
        return
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0                     # the children's failure is ours
    assert "local_rank: 1" in r.stderr           # two ranks really were started
    assert "No HIP GPUs" in r.stderr or "no usable HIP device" in r.stderr
    assert '{"metric"' not in r.stdout           # and no result line was invented


def test_bench_self_launch_relays_rank0_line(tmp_path):
    """The launcher half of bench.py on its own: a stand-in script plays the ranks."""
    import subprocess
    import sys
    import textwrap
    fake = tmp_path / "bench.py"
    src = open(os.path.join(ROOT, "bench.py")).read()
    # keep parse() + self_launch(), replace the measurement by a stub
    head = src[:src.index("def random_walk")]
    fake.write_text(head + textwrap.dedent("""
        def main():
            a = parse()
            if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
                self_launch(a)
            print("noise from rank %s" % os.environ["RANK"])
            if os.environ["RANK"] == "0":
                print(json.dumps({"metric": "m", "world": int(os.environ["WORLD_SIZE"])}))
            if a.steps == 13 and os.environ["RANK"] == "1":
                sys.exit(3)
        main()
        """))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(fake), "--gpus", "3"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip() == '{"metric": "m", "world": 3}'     # ONE line, rank 0's
    assert "noise from rank 2" in r.stderr
    r = subprocess.run([sys.executable, str(fake), "--gpus", "2", "--steps", "13"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0                                       # a rank's failure is relayed


def test_n_iter_beyond_the_device_history_is_refused_up_front():
    import blueberry_amd as bb
    with pytest.raises(ValueError, match="2\\*\\*20"):
        bb.StructureSolver(n_iter=(1 << 20) + 1)
    bb.StructureSolver(n_iter=1 << 20)


def test_entry_loops_without_python_loops_keep_last_wins():
    """`ContactMap.from_arrays` and `FithicContactMap.to_matrix` place their entries without a
    Python loop over them; of several entries for one cell the LAST stays,
    as in the reference's loops (`datatypes.pyx:268-271, 376-386`) -- checked against the loop."""
    from blueberry_amd.datatypes import _assign_last_wins
    rng = numpy.random.default_rng(21)
    for _ in range(10):
        d, n = int(rng.integers(1, 40)), int(rng.integers(0, 600))
        r, c, v = rng.integers(0, d, n), rng.integers(0, d, n), rng.random(n)
        want = numpy.zeros((d, d))
        for k in range(n):
            want[r[k], c[k]] = v[k]
        got = numpy.zeros((d, d))
        _assign_last_wins(got, r, c, v)
        assert numpy.array_equal(got, want)
    # through the public entry points, duplicates included
    res = 1000
    mids = (rng.integers(0, 12, (300, 2)) * res + res // 2).astype(float)
    contacts = numpy.column_stack([mids, rng.random(300)])
    cm = bb.ContactMap.from_arrays("c", 1, res, contacts, n_bins=12)
    want = numpy.zeros((13, 13))
    for m1, m2, val in contacts:
        j, k = int((m1 - res / 2) / res), int((m2 - res / 2) / res)
        want[j, k] = val
        want[k, j] = val
    assert numpy.array_equal(cm.matrix, want)
    fm = bb.FithicContactMap.from_array(numpy.column_stack([mids, rng.random((300, 3))]), res)
    got = fm.to_matrix("p", n_bins=12)
    want = numpy.zeros((13, 13))
    for m1, m2, _, p, _q in fm.map:
        want[int((m1 - res / 2) / res), int((m2 - res / 2) / res)] = p
    assert numpy.array_equal(got, want)


def _kernel_metadata(so_path):
    """{kernel name: {vgpr, sgpr, scratch}} from EVERY gfx950 code object embedded in a built
    .so -- one per translation unit (the fat binary holds several offload bundles; the
    bundler tool unbundles only the first): the AMDGPU ELF images are cut out of the file
    by their headers and read with llvm-readelf --notes.  No GPU needed."""
    import struct
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    blob = open(so_path, "rb").read()
    out = {}
    pos, images = 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        while True:
            pos = blob.find(b"\x7fELF", pos)
            if pos < 0:
                break
            hdr = blob[pos:pos + 64]
            pos += 4
            if len(hdr) < 64 or hdr[4] != 2 or struct.unpack_from("<H", hdr, 18)[0] != 224:
                continue                                   # not a 64-bit EM_AMDGPU image
            shoff, = struct.unpack_from("<Q", hdr, 40)
            shentsize, shnum = struct.unpack_from("<HH", hdr, 58)
            size = shoff + shentsize * shnum
            co = os.path.join(tmp, "k%d.co" % images)
            with open(co, "wb") as fh:
                fh.write(blob[pos - 4:pos - 4 + size])
            images += 1
            notes = subprocess.run([llvm + "/llvm-readelf", "--notes", co], text=True,
                                   capture_output=True).stdout
            for m in re.finditer(r"\.name:\s+(\S+).*?(?=\.name:|\Z)", notes, re.S):
                blk = m.group(0)
                if ".vgpr_count" not in blk:
                    continue
                g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
                out[m.group(1)] = {"vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"),
                                   "scratch": g("private_segment_fixed_size")}
    out["__images__"] = images
    return out


def test_no_product_kernel_spills():
    """round 3 shipped a 20-byte scratch spill in the narrow fp64 sweep and
    nothing noticed.  Every kernel of the product library must compile without scratch
    (private_segment_fixed_size = 0 in the code object's metadata), and the dominant
    kernels within the register budget their launch bounds assume."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("no LLVM binutils here")
    meta = _kernel_metadata(_lib.LIB_PATH)
    assert meta.pop("__images__") >= 4, "one code object per .hip translation unit"
    assert len(meta) > 120, "kernel metadata not found in %s" % _lib.LIB_PATH
    for name in ("gram_kernel", "normalize128_kernel", "band_count", "downsample_kernel"):
        assert any(name in k for k in meta), name          # every translation unit was read
    spills = {k: v["scratch"] for k, v in meta.items() if v["scratch"] > 0}
    assert not spills, "kernels with scratch: %r" % spills
    sweeps = {k: v for k, v in meta.items() if "stress_grad_kernel" in k}
    assert len(sweeps) >= 12
    for k, v in sweeps.items():
        if "IfLb" in k:                       # fp32: two workgroups of 8 waves per CU
            assert v["vgpr"] <= 128, (k, v)
        else:                                 # fp64: two waves per SIMD
            assert v["vgpr"] <= 256, (k, v)


def test_genome_boundaries_and_block_tile_list():
    """BASELINE config 5's tile list (bench.py --workload genome10kb): hg19 at 10 kb is
    309,568 bins, at 50 kb 61,914; tiles_from_blocks keeps exactly the tiles
    that hold a same-chromosome pair or a pair within the band, in device order, and
    counts the stored pairs i < j < n -- checked against a brute-force mask."""
    from blueberry_amd.solver import max_degree, tiles_from_blocks
    from blueberry_amd.utils import genome_boundaries
    b = genome_boundaries()
    assert b[0] == 0 and b[-1] == 309568 and len(b) == 25 and (numpy.diff(b) > 0).all()
    assert b[1] == 24926                                  # chr1 at 10 kb (config 3's size)
    assert genome_boundaries(resolution=50000)[-1] == 61914
    n, band = 5000, 300
    bs = genome_boundaries(n)
    assert bs[-1] == n and len(bs) == 25
    (ti, tj), pairs = tiles_from_blocks(n, bs, band, "float32")
    vw = 512
    chrom = numpy.searchsorted(bs, numpy.arange(n), side="right") - 1
    i, j = numpy.triu_indices(n, 1)
    wanted = (chrom[i] == chrom[j]) | (j - i <= band)
    nb = -(-n // vw)
    want_tiles = numpy.unique((j[wanted] // vw) * nb + i[wanted] // vw)
    got = tj.astype(numpy.int64) * nb + ti
    assert numpy.array_equal(got, want_tiles)             # same set, and (J, I) ascending
    have = numpy.zeros((nb, nb), dtype=bool)
    have[ti, tj] = True
    assert pairs == int(have[i // vw, j // vw].sum())
    deg = max_degree(n, (ti, tj), "float32")
    cnt = numpy.bincount(i[have[i // vw, j // vw]], minlength=n) + \
        numpy.bincount(j[have[i // vw, j // vw]], minlength=n)
    assert cnt.max() <= deg <= n
    # SPEC 2.4.1: a step per block -- every block's bound holds for each of its bins, the
    # factors are max / own (>= 1, exactly 1 where the bound is largest)
    from blueberry_amd.solver import block_degrees, block_step_factors
    degs = block_degrees(n, (ti, tj), "float32")
    assert degs.shape == (nb,) and degs.max() == deg
    assert (cnt <= numpy.repeat(degs, vw)[:n]).all()
    lr, scale = block_step_factors(n, (ti, tj), "float32")
    assert lr == 1.0 / (2 * deg) and scale.min() == 1.0 and numpy.allclose(scale * degs, deg)
    from blueberry_amd.solver import degree_step_factors
    lr_d, sc_d = degree_step_factors(cnt)
    assert lr_d == 1.0 / (2 * (cnt.max() + 1)) and sc_d.min() == 1.0
    assert numpy.allclose(lr_d * sc_d, 1.0 / (2 * (cnt + 1.0)))
    assert degree_step_factors(numpy.full(7, 6)) == (1.0 / 14, None)      # a complete map of 7 bins
    with pytest.raises(ValueError):
        tiles_from_blocks(n, [0, 10, 5, n], band, "float32")
