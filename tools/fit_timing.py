"""Wall-clock of a whole fit() from a host matrix (PCIe-inclusive), for DESIGN.md 5.1."""
import sys, time, os
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blueberry_amd.solver import HipEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = numpy.random.default_rng(0)
xs = numpy.cumsum(rng.standard_normal((n, 3)), axis=0)
w = numpy.empty((n, n))
for a in range(0, n, 1000):
    d = xs[a:a + 1000, None, :] - xs[None, :, :]
    w[a:a + 1000] = numpy.sqrt((d * d).sum(-1))
x0 = xs + 0.5 * rng.standard_normal(xs.shape)
for dtype in ("float32", "float64"):
    t0 = time.perf_counter()
    e = HipEngine(n, dtype)
    t1 = time.perf_counter()
    e.set_wish_dense(w, "wish", 3.0)
    t2 = time.perf_counter()
    e.set_coords(x0)
    e.iterate(k, 1.0 / (2 * n))
    e.sync()
    t3 = time.perf_counter()
    X = e.get_coords()
    t4 = time.perf_counter()
    e.close()
    pairs = n * (n - 1) // 2
    print("%s N=%d K=%d: create %.3f s, upload+pack %.3f s (%.2f GB/s of the %.1f GB upper triangle), "
          "iterate %.3f s (%.1f Gpair/s resident), fetch %.3f s; whole fit %.3f s = %.1f Gpair/s "
          "PCIe-inclusive" % (dtype, n, k, t1 - t0, t2 - t1, pairs * 8 / (t2 - t1) / 1e9, pairs * 8 / 1e9,
                              t3 - t2, pairs * k / (t3 - t2) / 1e9, t4 - t3, t4 - t0,
                              pairs * k / (t4 - t0) / 1e9))
