"""count_band_regions: GPU (end to end, incl. H2D of the regions and allocation)
vs the CPU oracle's restatement of the reference loop (one core)."""
import os, sys, time
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blueberry_amd as bb
from tests import _oracle

o = _oracle.load()
bb.count_band_regions(numpy.arange(10.0))          # HIP init
for n, res in ((20000, 5000), (50000, 5000), (309568, 10000)):
    r = numpy.arange(n) * float(res)
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps):
        g = bb.count_band_regions(r)
    tg = (time.perf_counter() - t0) / reps
    pairs = n * (n - 1) // 2
    line = "N=%d: GPU %.3f ms end-to-end = %.0f Gpair/s, count %d" % (n, tg * 1e3, pairs / tg / 1e9, g)
    if n <= 50000:
        t0 = time.perf_counter(); c = o.count_band_regions(r); tc = time.perf_counter() - t0
        assert c == g
        line += "; CPU oracle 1 core %.2f s = %.2f Gpair/s; x%.0f" % (tc, pairs / tc / 1e9, tc / tg)
    print(line)
r = numpy.arange(1000) * 50000.0
t0 = time.perf_counter()
for _ in range(200):
    bb.count_band_regions(r)
print("N=1000 (chr21 @ 50 kb): %.1f us per call on the GPU" % ((time.perf_counter() - t0) / 200 * 1e6))
t0 = time.perf_counter(); o.count_band_regions(r); print("          CPU oracle: %.1f us" % ((time.perf_counter() - t0) * 1e6))
