"""Per-iteration time in the launch-bound regime (small chromosomes)."""
import os, sys, time
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blueberry_amd.solver import HipEngine
for n, dtype in ((963, "float64"), (963, "float32"), (5000, "float32"), (12000, "float32"),
                 (17700, "float32")):
    xs = numpy.cumsum(numpy.random.default_rng(0).standard_normal((n, 3)), axis=0)
    e = HipEngine(n, dtype)
    e.set_wish_from_coords(xs)
    e.set_coords(xs + 0.5)
    e.iterate(50, 1 / (2 * n)); e.sync()
    k = 2000
    t0 = time.perf_counter(); e.iterate(k, 1 / (2 * n)); e.sync(); dt = time.perf_counter() - t0
    e.set_timing(True)
    t0 = time.perf_counter(); e.iterate(k, 1 / (2 * n)); e.sync(); dtt = time.perf_counter() - t0
    tm = e.timing()
    print("N=%d %s: %.2f us per iteration (%.2f us with the timing events on; kernel %.2f us, "
          "reduce+update %.2f us)"
          % (n, dtype, dt / k * 1e6, dtt / k * 1e6, tm["grad_ms"] * 1e3, tm["reduce_ms"] * 1e3))
    e.close()
