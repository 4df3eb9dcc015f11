"""CPU: the oracle against the reference's golden vectors (captured from the
real Cython functions by tests/golden/make_golden.py) and, for the solver --
which the reference does not contain -- against closed forms."""
import numpy
import pytest

from tests import _oracle


# ---- K1 count_band_regions (blueberry.pyx:77-91): exact -------------------
def band_cases():
    z = _oracle.golden("band_count")
    return sorted(k[3:] for k in z.files if k.startswith("in_"))


@pytest.mark.parametrize("name", band_cases())
def test_band_count_matches_reference(oracle, name):
    z = _oracle.golden("band_count")
    assert oracle.count_band_regions(z["in_" + name]) == int(z["out_" + name])


def test_band_count_closed_form(oracle):
    # uniform 50 kb grid: pairs at distance d*50kb, d = 1..200
    r = numpy.arange(1000) * 50000.0 + 25000
    assert oracle.count_band_regions(r) == sum(1000 - d for d in range(1, 201)) == 179900


def test_band_count_rows_partition(oracle):
    r = _oracle.golden("band_count")["in_gappy_n1000"]
    full = oracle.count_band_regions(r)
    cuts = [0, 1, 333, 334, 900, 1000]
    assert sum(oracle.count_band_regions_rows(r, a, b) for a, b in zip(cuts, cuts[1:])) == full


# ---- A2/A3 ContactMap scatter + normalize (datatypes.pyx:97-171): bit-exact --
@pytest.mark.parametrize("k", [0, 1, 2])
def test_contactmap_matches_reference(oracle, k):
    z = _oracle.golden("contactmap")
    n_bins = z["cm%d_krnorm" % k].shape[0]
    raw = oracle.contactmap_scatter(z["cm%d_triples" % k], int(z["cm%d_resolution" % k]), n_bins)
    assert raw.shape == z["cm%d_matrix_raw" % k].shape
    assert numpy.array_equal(raw, z["cm%d_matrix_raw" % k])
    norm = oracle.contactmap_normalize(raw, z["cm%d_krnorm" % k], z["cm%d_krexp" % k])
    assert numpy.array_equal(norm, z["cm%d_matrix_norm" % k])       # bit for bit
    assert numpy.array_equal(norm, norm.T)


def test_reference_raises_on_zero_kr():
    assert bool(_oracle.golden("contactmap")["cm_zero_kr_raises_zerodivision"])


# ---- f4: benjamini_hochberg / downsample (blueberry.pyx:40-75, 93-104) -----
@pytest.mark.parametrize("k", range(8))         # 4..7: NaN p-values (NaN out, maximum restarts)
def test_bh_matches_reference(oracle, k):
    z = _oracle.golden("bh_downsample")
    q = oracle.benjamini_hochberg(z["bh_p_%d" % k], int(z["bh_n_%d" % k]))
    assert numpy.array_equal(q, z["bh_q_%d" % k], equal_nan=True)
    if k >= 4:
        assert numpy.isnan(z["bh_q_%d" % k]).sum() == numpy.isnan(z["bh_p_%d" % k]).sum() > 0


@pytest.mark.parametrize("k", [0, 1, 2])
def test_downsample_matches_reference(oracle, k):
    z = _oracle.golden("bh_downsample")
    out = oracle.downsample(z["ds_yp1_%d" % k], z["ds_yp5i_%d" % k])
    assert numpy.array_equal(out, z["ds_out_%d" % k])


# ---- S0 solver: PARITY UNPINNED vs the reference (it has no solver); --------
# ---- pinned by closed forms instead (docs/SPEC.md) --------------------------
def test_stress_zero_at_generating_coordinates(oracle):
    x = _oracle.random_walk(60)
    s, g = oracle.stress_grad(_oracle.wish_from_coords(x), x)
    assert s < 1e-24 and numpy.abs(g).max() < 1e-11


def test_tetrahedron_closed_form(oracle):
    # regular tetrahedron of edge a, wish distance b on every edge:
    # S = 6 (a-b)^2 ; g_i = 2 (a-b)/a * sum_j (x_i - x_j) = 8 (a-b)/a * (x_i - centroid)
    x = numpy.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], dtype=float)
    a, b = numpy.sqrt(8.0), 2.0
    w = numpy.full((4, 4), b)
    numpy.fill_diagonal(w, 0)
    s, g = oracle.stress_grad(w, x)
    assert abs(s - 6 * (a - b) ** 2) < 1e-13
    assert numpy.allclose(g, 8 * (a - b) / a * (x - x.mean(0)), atol=1e-13)


def test_collinear_pair(oracle):
    x = numpy.array([[0, 0, 0], [3, 0, 0]], dtype=float)
    w = numpy.array([[0, 1.0], [1.0, 0]])
    s, g = oracle.stress_grad(w, x)
    assert s == 4.0 and numpy.array_equal(g, [[-4, 0, 0], [4, 0, 0]])


def test_zero_wish_means_no_constraint(oracle):
    x = _oracle.random_walk(10)
    s, g = oracle.stress_grad(numpy.zeros((10, 10)), x)
    assert s == 0 and not g.any()


def test_coincident_points_are_finite(oracle):
    x = numpy.zeros((3, 3))
    w = numpy.ones((3, 3)) - numpy.eye(3)
    for f64 in (True, False):
        s, g = oracle.stress_grad(w, x, f64=f64)
        assert numpy.isfinite(s) and numpy.isfinite(g).all() and abs(s - 3.0) < 1e-12


def test_gradient_matches_finite_differences(oracle):
    rng = numpy.random.default_rng(5)
    n = 12
    x = rng.standard_normal((n, 3))
    w = _oracle.wish_from_coords(_oracle.random_walk(n, seed=7))
    w[rng.random((n, n)) < 0.2] = 0
    w = numpy.triu(w, 1) + numpy.triu(w, 1).T
    _, g = oracle.stress_grad(w, x)
    h = 1e-6
    for i, c in ((0, 0), (5, 1), (11, 2)):
        xp, xm = x.copy(), x.copy()
        xp[i, c] += h
        xm[i, c] -= h
        fd = (oracle.stress_grad(w, xp)[0] - oracle.stress_grad(w, xm)[0]) / (2 * h)
        assert abs(fd - g[i, c]) < 1e-6 * max(1.0, abs(g[i, c]))


def test_counts_to_wish(oracle):
    c = numpy.array([[5.0, 8.0, 0.0], [8.0, 1.0, numpy.inf], [0.0, numpy.inf, 2.0]])
    w = oracle.counts_to_wish(c, alpha=3.0)
    assert w[0, 1] == w[1, 0] == 8.0 ** (-1.0 / 3.0)
    assert w[0, 2] == 0 and w[1, 2] == 0 and not numpy.diag(w).any()


def test_smacof_step_decreases_stress(oracle):
    n = 80
    xs = _oracle.random_walk(n)
    _, hist = oracle.solve(_oracle.wish_from_coords(xs), _oracle.noisy_init(xs), 15, 1.0 / (2 * n))
    assert (numpy.diff(hist) < 0).all() and hist[-1] < 0.05 * hist[0]


def test_units_partition_sums_to_full(oracle):
    """Summing the per-unit-range shares over any partition of the device
    layout's units gives the full lower-triangle result: every pair i<j is
    owned by exactly one unit (the property the N>1 path relies on)."""
    from blueberry_amd import _lib
    lib = _lib.load()
    n = 300
    rng = numpy.random.default_rng(3)
    xs = _oracle.random_walk(n)
    w = _oracle.wish_from_coords(xs)
    x = _oracle.noisy_init(xs)
    s_full, g_full = oracle.stress_grad(w, x)
    for dtype in (_lib.BB_F32, _lib.BB_F64):
        info = _lib.LayoutInfo()
        _lib.check(lib.bb_layout_dense_info(n, dtype, info))
        ti = numpy.zeros(info.n_tiles, dtype=numpy.int32)
        tj = numpy.zeros(info.n_tiles, dtype=numpy.int32)
        _lib.check(lib.bb_layout_dense_tiles(n, dtype, ti.ctypes.data_as(_lib.p_i32),
                                             tj.ctypes.data_as(_lib.p_i32), info.n_tiles))
        cuts = sorted(set([0, info.n_units] + list(rng.integers(0, info.n_units, 4))))
        s_sum, g_sum = 0.0, numpy.zeros_like(x)
        for a, b in zip(cuts, cuts[1:]):
            s, g = oracle.stress_grad_units(w, x, ti, tj, info.units_per_tile, info.vw, a, b)
            s_sum += s
            g_sum += g
        assert abs(s_sum - s_full) <= 1e-12 * s_full
        assert numpy.abs(g_sum - g_full).max() <= 1e-12 * numpy.abs(g_full).max()


def test_momentum_zero_is_plain_solve(oracle):
    n = 60
    xs = _oracle.random_walk(n)
    w, x0 = _oracle.wish_from_coords(xs), _oracle.noisy_init(xs)
    a = oracle.solve(w, x0, 9, 1.0 / (2 * n))
    b = oracle.solve_momentum(w, x0, 9, 1.0 / (2 * n), 0.0)
    assert numpy.array_equal(a[0], b[0]) and numpy.array_equal(a[1], b[1])


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_multicore_baseline_equals_scalar_oracle(oracle, threads):
    """oracle/bb_oracle_mt.c (bench.py's multi-core cpu_baseline) runs the same
    iteration as bbo_solve; only the order of the sums differs."""
    n = 257
    xs = _oracle.random_walk(n)
    w = _oracle.wish_from_coords(xs)
    w[5, 9] = w[9, 5] = 0.0                       # a missing pair
    x0 = _oracle.noisy_init(xs)
    X1, h1 = oracle.solve(w, x0, 6, 1.0 / (2 * n))
    Xt, ht = _oracle.solve_mt(w, x0, 6, 1.0 / (2 * n), threads)
    assert numpy.abs(Xt - X1).max() < 1e-12 * numpy.abs(X1).max()
    assert numpy.abs(ht / h1 - 1).max() < 1e-12
    Xt2, ht2 = _oracle.solve_mt(w, x0, 6, 1.0 / (2 * n), threads)
    assert numpy.array_equal(Xt, Xt2) and numpy.array_equal(ht, ht2)   # reproducible


@pytest.mark.parametrize("threads", [1, 4])
def test_multicore_momentum_twin_equals_scalar_oracle(oracle, threads):
    """bbo_solve_momentum_mt (the K = 20 twin of the depth-parity GPU tests) runs
    bbo_solve_momentum's iteration; mu = 0 is bbo_solve_mt bit for bit."""
    n = 193
    xs = _oracle.random_walk(n)
    w = _oracle.wish_from_coords(xs)
    w[7, 90] = w[90, 7] = 0.0
    x0 = _oracle.noisy_init(xs)
    lr = 1.0 / (2 * n)
    X1, h1 = oracle.solve_momentum(w, x0, 20, lr, 0.5)
    Xt, ht = _oracle.solve_momentum_mt(w, x0, 20, lr, 0.5, threads)
    assert numpy.abs(Xt - X1).max() < 1e-11 * numpy.abs(X1).max()
    assert numpy.abs(ht / h1 - 1).max() < 1e-11
    a = _oracle.solve_mt(w, x0, 7, lr, threads)
    b = _oracle.solve_momentum_mt(w, x0, 7, lr, 0.0, threads)
    assert numpy.array_equal(a[0], b[0]) and numpy.array_equal(a[1], b[1])


@pytest.mark.parametrize("f64", [True, False])
def test_generated_delta_oracle_equals_scalar_oracle_on_the_explicit_matrix(oracle, f64):
    """bbo_solve_gen_mt forms delta_ij = |x*_i - x*_j| on the fly (rounded to float when
    the device would store it as float) over a tile list: on a small map it must equal
    the scalar bbo_solve run on the explicit matrix with the same rounding -- dense, and
    block-sparse (absent tiles = zeros in the matrix)."""
    n, vw, k = 300, 64, 8
    xs = _oracle.random_walk(n)
    xs[11] = xs[10]                               # a zero wish distance: no constraint
    x0 = _oracle.noisy_init(xs)
    w = _oracle.wish_from_coords(xs)
    if not f64:
        w = w.astype(numpy.float32).astype(numpy.float64)
    lr = 1.0 / (2 * n)
    X1, h1 = oracle.solve(w, x0, k, lr, f64=f64)
    Xg, hg = _oracle.solve_gen_mt(xs, x0, k, lr, 3, vw=vw, f64=f64)
    assert numpy.abs(Xg - X1).max() < 1e-12 * numpy.abs(X1).max()
    assert numpy.abs(hg / h1 - 1).max() < 1e-12
    # momentum twin
    X2, h2 = oracle.solve_momentum(w, x0, k, lr, 0.4, f64=f64)
    Xm, hm = _oracle.solve_gen_mt(xs, x0, k, lr, 2, vw=vw, mu=0.4, f64=f64)
    assert numpy.abs(Xm - X2).max() < 1e-12 * numpy.abs(X2).max()
    assert numpy.abs(hm / h2 - 1).max() < 1e-12
    # block-sparse: the diagonal tiles and one band
    ti, tj = _oracle.dense_tiles(n, vw)
    keep = (tj - ti) <= 1
    tiles = (ti[keep], tj[keep])
    mask = numpy.zeros((n, n), dtype=bool)
    for a, b in zip(*tiles):
        mask[a * vw:(a + 1) * vw, b * vw:(b + 1) * vw] = True
    mask |= mask.T
    ws = numpy.where(mask, w, 0.0)
    X3, h3 = oracle.solve(ws, x0, k, lr, f64=f64)
    Xs, hs = _oracle.solve_gen_mt(xs, x0, k, lr, 4, tiles=tiles, vw=vw, f64=f64)
    assert numpy.abs(Xs - X3).max() < 1e-12 * numpy.abs(X3).max()
    assert numpy.abs(hs / h3 - 1).max() < 1e-12


def test_solver_oracle_equals_sklearn_smacof_on_a_complete_map(oracle):
    """S0 has no reference to pin to, so the oracle's solver is
    additionally pinned to a NAMED third-party algorithm: for a complete wish matrix, one
    step of `bbo_solve` with lr = 1/(2N) is the Guttman transform, i.e. one iteration
    of scikit-learn's metric SMACOF (`sklearn.manifold.smacof`, `_smacof_single`:
    X <- B(X) X / n, then the raw stress of the NEW X).  K chained single iterations
    of sklearn from the oracle's own iterates must give the oracle's stress history
    (shifted by one: sklearn reports S(X_{k+1})) and its coordinates up to the centroid
    -- the Guttman transform centres X, the gradient step keeps X's centroid.
    Tolerances: sklearn's euclidean_distances uses the |x|^2 + |y|^2 - 2xy expansion
    (about 1e-12 relative on these distances), so 1e-9, not 1e-12.  Incomplete maps
    (weights) have no sklearn counterpart and stay pinned by closed forms only."""
    smacof = pytest.importorskip("sklearn.manifold").smacof
    n, k = 60, 6
    xs = _oracle.random_walk(n)
    w = _oracle.wish_from_coords(xs)
    x0 = _oracle.noisy_init(xs)
    lr = 1.0 / (2 * n)
    X, hist, iterates = x0.copy(), [], [x0.copy()]
    for _ in range(k + 1):
        X, h = oracle.solve(w, X, 1, lr)          # one step at a time: keep every iterate
        hist.append(h[0])
        iterates.append(X.copy())
    X_all, h_all = oracle.solve(w, x0, k + 1, lr)
    assert numpy.array_equal(h_all, numpy.array(hist))          # chaining changes nothing
    for it in range(k):
        Xs, stress, _ = smacof(w, metric=True, n_components=3, init=iterates[it], n_init=1,
                               max_iter=1, eps=0.0, normalized_stress=False, return_n_iter=True)
        assert abs(stress / hist[it + 1] - 1) < 1e-9, (it, stress, hist[it + 1])
        mine = iterates[it + 1] - iterates[it + 1].mean(axis=0)
        assert numpy.abs(Xs - mine).max() < 1e-9 * numpy.abs(mine).max(), it


# ---- an independent restatement of SPEC 2.1-2.4 ----------
# The C oracle's solver has no reference to be pinned to, and its weighted branch
# (incomplete maps, counts -> delta, the eps^2 floor, momentum) was until round 3 checked
# against the oracle itself, finite differences and -- complete maps only -- scikit-learn.
# This is SPEC 2.1-2.4 written a second time, from the formulas and not from the C: dense
# numpy over all ordered pairs, no loops over pairs, a different summation order.
def _spec_wish(counts, alpha=3.0):                       # SPEC 2.1
    c = numpy.asarray(counts, dtype=numpy.float64)
    ok = numpy.isfinite(c) & (c > 0)
    w = numpy.where(ok, numpy.power(numpy.where(ok, c, 1.0), -1.0 / alpha), 0.0)
    numpy.fill_diagonal(w, 0.0)
    return w


def _spec_stress_grad(wish, X, eps2):                    # SPEC 2.2, 2.3
    sym = numpy.triu(wish, 1)
    sym = sym + sym.T                                    # only [i][j], i < j, counts
    diff = X[:, None, :] - X[None, :, :]
    d = numpy.sqrt((diff ** 2).sum(axis=2) + eps2)
    on = sym > 0
    res = numpy.where(on, d - sym, 0.0)
    stress = 0.5 * (res ** 2).sum()                      # every pair appears twice
    grad = (2.0 * (res / d)[:, :, None] * diff).sum(axis=1)
    return stress, grad


def _spec_solve(wish, X0, iters, lr, mu, eps2):          # SPEC 2.4
    X, V, hist = X0.copy(), numpy.zeros_like(X0), []
    for _ in range(iters):
        s, g = _spec_stress_grad(wish, X, eps2)
        hist.append(s)
        V = mu * V - lr * g
        X = X + V
    return X, numpy.array(hist)


def _awkward_counts(n, seed):
    """A count matrix with everything SPEC 2.1 has a clause for: zeros (about half the
    pairs), whole rows without any contact, inf, NaN and negative entries.  Symmetric, as
    every ContactMap matrix is by construction (datatypes.pyx:115-116): the device packs
    [i][j], i < j, the C oracle walks `for i: for j < i` like blueberry.pyx:86-87."""
    rng = numpy.random.default_rng(seed)
    c = rng.gamma(0.7, 40.0, (n, n))
    c[rng.random((n, n)) < 0.5] = 0.0
    c = numpy.triu(c, 1)
    c = c + c.T
    c[[3, n // 2]] = 0.0
    c[:, [3, n // 2]] = 0.0                              # two bins nobody touches
    c[1, 7] = c[7, 1] = numpy.inf
    c[2, 9] = c[9, 2] = numpy.nan
    c[4, 11] = c[11, 4] = -5.0
    return c


@pytest.mark.parametrize("f64", [True, False])
def test_solver_oracle_equals_the_numpy_restatement_on_incomplete_maps(oracle, f64):
    n, k = 40, 12
    eps2 = 1e-300 if f64 else 1e-30
    counts = _awkward_counts(n, 5)
    w_c = oracle.counts_to_wish(counts)
    w_np = _spec_wish(counts)
    # libm's pow and numpy's may differ in the last bit; "no constraint" must agree exactly
    assert numpy.array_equal(w_c > 0, w_np > 0) and numpy.array_equal(w_c, w_c.T)
    assert numpy.abs(w_c - w_np).max() <= 4e-16 * w_np.max()
    w_np = w_c.copy()                                    # same bits into both solvers
    assert (numpy.triu(w_np, 1) > 0).sum() < 0.6 * n * (n - 1) / 2      # really incomplete
    X = numpy.random.default_rng(6).standard_normal((n, 3)) * 3.0
    X[20] = X[21]                                        # coincident points: d = eps, force 0
    s_c, g_c = oracle.stress_grad(w_c, X, f64=f64)
    s_np, g_np = _spec_stress_grad(w_np, X, eps2)
    assert abs(s_c / s_np - 1) < 1e-13
    assert numpy.abs(g_c - g_np).max() < 1e-13 * numpy.abs(g_np).max()
    assert numpy.all(g_c[[3, n // 2]] == 0.0) and numpy.all(numpy.isfinite(g_c))
    lr = 1.0 / (2 * n)
    for mu in (0.0, 0.5):
        if mu == 0.0:
            X_c, h_c = oracle.solve(w_c, X, k, lr, f64=f64)
        else:
            X_c, h_c = oracle.solve_momentum(w_c, X, k, lr, mu, f64=f64)
        X_np, h_np = _spec_solve(w_np, X, k, lr, mu, eps2)
        assert numpy.abs(h_c / h_np - 1).max() < 1e-12, mu
        assert numpy.abs(X_c - X_np).max() < 1e-12 * numpy.abs(X_np).max(), mu
        assert numpy.array_equal(X_c[[3, n // 2]], X[[3, n // 2]])      # unconstrained bins stay


def test_numpy_restatement_on_a_wish_matrix_with_bad_entries(oracle):
    """kind='wish': anything that is not a finite positive number is "no constraint"."""
    n = 30
    xs = _oracle.random_walk(n, 2)
    w = _oracle.wish_from_coords(xs)
    w[0, 5] = w[5, 0] = 0.0
    w[1, 6] = w[6, 1] = -1.0
    w[2, 8] = w[8, 2] = numpy.nan
    X = _oracle.noisy_init(xs, 3)
    s_c, g_c = oracle.stress_grad(w, X)
    w_np = numpy.where(numpy.isfinite(w) & (w > 0), w, 0.0)
    s_np, g_np = _spec_stress_grad(w_np, X, 1e-300)
    assert abs(s_c / s_np - 1) < 1e-13
    assert numpy.abs(g_c - g_np).max() < 1e-13 * numpy.abs(g_np).max()
