"""fit_triples (triples resident on the device: cleaned, binned, mapped to tiles and scattered
there) over ragged inputs -- sizes from 3 bins to 3,000, repeated pairs in either orientation,
NaN / inf counts, positions inside their bins, C-ordered and column-major arrays, with and
without KR vectors -- against the oracle's restatement of the reference chain
(datatypes.pyx:100-116, :161-171, then SPEC 2) on the dense matrix (functional evidence)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import blueberry_amd as bb
from tests import _oracle

oracle = _oracle.load()
rng = numpy.random.default_rng(2024)
bad = 0
for n_bins in (3, 17, 127, 128, 129, 511, 513, 1000, 2048, 3000):
    for variant in range(3):
        res = int(rng.choice([1000, 5000, 10000]))
        m = int(max(4, n_bins * rng.integers(2, 30)))
        bi = rng.integers(0, n_bins, m)
        bj = numpy.minimum(n_bins - 1, bi + rng.geometric(min(0.9, 8.0 / n_bins + 0.01), m))
        flip = rng.random(m) < 0.5
        bi, bj = numpy.where(flip, bj, bi), numpy.where(flip, bi, bj)
        pos_i = bi * float(res) + rng.integers(0, res, m)
        pos_j = bj * float(res) + rng.integers(0, res, m)
        c = rng.integers(1, 500, m).astype(float)
        c[rng.random(m) < 0.02] = numpy.nan
        c[rng.random(m) < 0.02] = numpy.inf
        tr = numpy.stack([pos_i, pos_j, c], 1)
        if variant == 1:
            tr = numpy.asfortranarray(tr)
        kr = ke = None
        if variant == 2:
            kr = 0.5 + rng.random(n_bins)
            kr[rng.random(n_bins) < 0.05] = numpy.nan
            ke = 30.0 / (1.0 + numpy.arange(n_bins)) + 0.2
        n = n_bins + 1
        x0 = numpy.random.default_rng(5).standard_normal((n, 3))
        lr = 1.0 / (2 * n)
        raw = oracle.contactmap_scatter(numpy.ascontiguousarray(tr), res, n_bins)
        mat = oracle.contactmap_normalize(raw, kr, ke) if kr is not None else raw
        wish = oracle.counts_to_wish(mat, 3.0)
        for dtype, tol in (("float64", 1e-12), ("float32", 1e-5)):
            # SPEC 2.1: a wish distance below the dtype's floor is "no constraint" (an infinite
            # count becomes the largest double, delta = 1.8e-103: a constraint in fp64 only)
            w = numpy.where(wish < (1e-290 if dtype == "float64" else 1e-30), 0.0, wish)
            X_ref, h_ref = oracle.solve(w, x0, 3, lr, f64=dtype == "float64")
            s = bb.StructureSolver(n_iter=3, lr=lr, dtype=dtype).fit_triples(tr, res, n_bins, KRnorm=kr,
                                                                             KRexpected=ke, init=x0)
            es = numpy.abs(s.stress_ - h_ref).max() / max(h_ref[0], 1e-300)
            ex = numpy.abs(s.structure_ - X_ref).max() / numpy.abs(X_ref).max()
            ok = es < tol and ex < tol
            bad += not ok
            print("n_bins %5d variant %d (%s%s) %s: stress %.1e coords %.1e %s"
                  % (n_bins, variant, "F-order " if variant == 1 else "", "KR" if variant == 2 else "",
                     dtype, es, ex, "ok" if ok else "FAIL"), flush=True)
print("FAILURES:", bad)
