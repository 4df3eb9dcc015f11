"""Iterations (and device time) to 1e-3 of the start's stress with one step for all -- SPEC
2.4's 1 / (2 N), and the step of the best-connected bin, 1 / (2 (D + 1)) -- against a step per
bin from the map's own degrees (SPEC 2.4.1, StructureSolver(degree_steps=True)), on four kinds
of map of one size: complete; two dense blocks of unequal size plus a band (a genome in small);
8 % of the pairs at random plus the near diagonal; pairs kept with probability ~ 1 / |i - j|
(what a Hi-C map's zeros look like).  Plain steps and heavy-ball 0.5; the plain runs must be
descents (checked down to 1e-6 of the start's stress: below that fp32 rounding moves it).

    python tools/degree_steps_table.py [n_bins=6000]"""
import os
import sys
import time

import numpy

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from blueberry_amd.solver import HipEngine, degree_step_factors      # noqa: E402
from tests import _oracle                                            # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
K = 400
xs = _oracle.random_walk(n)
w = _oracle.wish_from_coords(xs)
x0 = _oracle.noisy_init(xs)
rng = numpy.random.default_rng(3)
i, j = numpy.indices((n, n))
sep = numpy.abs(i - j)
near = sep <= 3
maps = {"complete": numpy.ones((n, n), dtype=bool)}
cut = (5 * n) // 6
maps["two blocks 5:1 + band"] = ((i < cut) & (j < cut)) | ((i >= cut) & (j >= cut)) | (sep <= 40)
u = numpy.triu(rng.random((n, n)), 1)
u = u + u.T
maps["8 % at random + near diagonal"] = (u < 0.08) | near
maps["kept with p = min(1, 30 / |i-j|)"] = (u < numpy.minimum(1.0, 30.0 / numpy.maximum(sep, 1))) | near


def run(e, lr, scale, mu):
    e.set_bin_steps(scale)
    e.set_momentum(mu)
    e.set_coords(x0)
    e.sync()
    t0 = time.perf_counter()
    e.iterate(K, lr)
    e.sync()
    dt = time.perf_counter() - t0
    h = e.stress_history()
    below = numpy.nonzero(h <= 1e-3 * h[0])[0]
    k = int(below[0]) if below.size else None
    live = h[:-1] > 1e-6 * h[0]
    descent = mu > 0 or bool((numpy.diff(h)[live] <= 0).all())
    return k, (None if k is None else k * dt / K * 1e3), h[-1] / h[0], descent


print("n = %d bins, fp32; iterations to 1e-3 of the start's stress (device ms), within %d; plain steps / heavy-ball 0.5" % (n, K))
for name, keep in maps.items():
    wm = numpy.where(keep, w, 0.0)
    numpy.fill_diagonal(wm, 0.0)
    e = HipEngine(n, "float32")
    e.set_wish_dense(wm, "wish", 3.0)
    deg = e.degrees()
    lr_d, scale = degree_step_factors(deg)
    rows = [("1/(2N)", 1.0 / (2 * n), None), ("1/(2(D+1))", lr_d, None), ("per bin", lr_d, scale)]
    out = []
    for label, lr, sc in rows:
        cell = []
        for mu in (0.0, 0.5):
            k, ms, ratio, descent = run(e, lr, sc, mu)
            cell.append((("%d its %.2f ms" % (k, ms)) if k is not None else ("not in %d (%.0e)" % (K, ratio)))
                        + ("" if descent else " NOT A DESCENT"))
        out.append("%s: %s" % (label, " / ".join(cell)))
    print("%-34s degrees %5d..%-5d  %s" % (name, deg.min(), deg.max(), " | ".join(out)), flush=True)
    e.close()
