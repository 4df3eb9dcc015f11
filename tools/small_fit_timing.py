"""Wall clock of a whole StructureSolver.fit() on small maps (chr21@50kb size and around):
how much of it is the solver's set-up (handle, streams, index tables, uploads) rather than
the iterations."""
import os, sys, time
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blueberry_amd as bb
from blueberry_amd.solver import HipEngine

for n in (963, 2500, 5000):
    rng = numpy.random.default_rng(0)
    xs = numpy.cumsum(rng.standard_normal((n, 3)), axis=0)
    d = numpy.sqrt(((xs[:, None, :] - xs[None, :, :]) ** 2).sum(-1))
    x0 = xs + 0.5 * rng.standard_normal(xs.shape)
    for dtype in ("float64", "float32"):
        best = {}
        for rep in range(4):
            t = [time.perf_counter()]
            e = HipEngine(n, dtype); t.append(time.perf_counter())
            e.set_wish_dense(d, "wish", 3.0); t.append(time.perf_counter())
            e.set_coords(x0); t.append(time.perf_counter())
            e.iterate(100, 1.0 / (2 * n)); e.sync(); t.append(time.perf_counter())
            X = e.get_coords(); h = e.stress_history(); t.append(time.perf_counter())
            e.close(); t.append(time.perf_counter())
            dt = numpy.diff(t) * 1e3
            if rep == 0 or dt.sum() < best["sum"]:
                best = {"sum": dt.sum(), "dt": dt}
        dt = best["dt"]
        t0 = time.perf_counter(); s = bb.StructureSolver(n_iter=100, dtype=dtype, kind="wish").fit(d, init=x0); tf = (time.perf_counter() - t0) * 1e3
        print("N=%5d %s: create %.2f ms, set_wish_dense %.2f, set_coords %.2f, 100 iterations %.2f, fetch %.2f, close %.2f  = %.2f ms; StructureSolver.fit %.2f ms"
              % (n, dtype, dt[0], dt[1], dt[2], dt[3], dt[4], dt[5], dt.sum(), tf))
