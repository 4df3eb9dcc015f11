"""fit_many over ragged batches -- tiny maps, maps that end on a tile edge, many maps, fp64
batches on both sides of the narrow / wide layout switch -- every map against the oracle's
solve of that map alone (functional evidence)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import blueberry_amd as bb
from tests import _oracle

oracle = _oracle.load()
rng = numpy.random.default_rng(123)
batches = [[2, 3, 5], [512, 512, 512], [513, 1, 2][:1] + [2, 640], [128, 256, 384, 127, 129],
           [1024, 1023, 1025], [3000, 2], [2, 3000], [700] * 12, [4096, 100], [100, 3900],
           list(rng.integers(2, 900, 40)), [2049, 2048, 2047]]
bad = 0
for sizes in batches:
    sizes = [int(v) for v in sizes]
    mats, x0s = [], []
    for q, n in enumerate(sizes):
        xs = _oracle.random_walk(n, seed=q)
        w = _oracle.wish_from_coords(xs)
        if n > 50:
            w[3, 40] = w[40, 3] = 0.0
        mats.append(w)
        x0s.append(_oracle.noisy_init(xs, seed=77 + q))
    for dtype, tol in (("float64", 1e-12), ("float32", 1e-5)):
        for mu in (0.0, 0.3):
            s = bb.StructureSolver(n_iter=4, dtype=dtype, kind="wish", momentum=mu).fit_many(mats, inits=x0s)
            worst = 0.0
            for q, n in enumerate(sizes):
                f64 = dtype == "float64"
                X, h = (oracle.solve_momentum(mats[q], x0s[q], 4, 1.0 / (2 * n), mu, f64=f64) if mu
                        else oracle.solve(mats[q], x0s[q], 4, 1.0 / (2 * n), f64=f64))
                ex = numpy.abs(s.structures_[q] - X).max() / max(numpy.abs(X).max(), 1e-300)
                # relative to the start's stress: a 2- or 3-point map reaches its wish
                # distances exactly and its later stress values are rounding noise
                es = numpy.abs(s.stresses_[q] - h).max() / h[0]
                worst = max(worst, ex, es)
            ok = worst < tol
            bad += not ok
            print("%-8s mu=%.1f %3d maps %6d bins (min %d max %d): worst rel err %.2e %s"
                  % (dtype, mu, len(sizes), sum(sizes), min(sizes), max(sizes), worst, "ok" if ok else "FAIL"),
                  flush=True)
print("FAILURES:", bad)
