"""Per-iteration time in the launch-bound regime (small chromosomes): the row-owner path
(one launch per iteration) against the unit sweep + two reduce launches,
at the same sizes, in one process.  BB_ROW_OWNER_MAX picks the path per engine."""
import os, sys, time
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blueberry_amd.solver import HipEngine
cases = [(963, "float64"), (963, "float32"), (2500, "float64"), (2500, "float32"),
         (4096, "float64"), (4096, "float32"), (6000, "float64"), (6000, "float32"),
         (8192, "float32"), (12000, "float32"), (17700, "float32")]
for n, dtype in cases:
    xs = numpy.cumsum(numpy.random.default_rng(0).standard_normal((n, 3)), axis=0)
    line = "N=%d %s:" % (n, dtype)
    for path, limit in (("row-owner", "1000000"), ("units", "0")):
        if path == "row-owner" and n > 8192:
            continue
        os.environ["BB_ROW_OWNER_MAX"] = limit
        e = HipEngine(n, dtype)
        e.set_wish_from_coords(xs)
        e.set_coords(xs + 0.5)
        e.iterate(50, 1 / (2 * n)); e.sync()
        k = 2000 if n <= 6000 else 500
        t0 = time.perf_counter(); e.iterate(k, 1 / (2 * n)); e.sync(); dt = time.perf_counter() - t0
        e.set_timing(8)
        e.iterate(400, 1 / (2 * n)); e.sync()
        tm = e.timing()
        line += "  %s %.2f us/iter (kernel %.2f, reduce+update %.2f)" % (
            path, dt / k * 1e6, tm["grad_ms"] * 1e3, tm["reduce_ms"] * 1e3)
        e.close()
    print(line, flush=True)
