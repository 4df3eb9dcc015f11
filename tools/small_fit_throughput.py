"""Whole chr21@50kb-sized fits per second on ONE GPU with 1, 2, 4, 8 host threads (one
handle per thread, streams from the library's pool): what serving many small maps gets."""
import os, sys, time
import numpy
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blueberry_amd as bb

n, K = 963, 100
rng = numpy.random.default_rng(0)
xs = numpy.cumsum(rng.standard_normal((n, 3)), axis=0)
d = numpy.sqrt(((xs[:, None, :] - xs[None, :, :]) ** 2).sum(-1))
x0 = xs + 0.5 * rng.standard_normal(xs.shape)
def fits(count, dtype):
    for _ in range(count):
        bb.StructureSolver(n_iter=K, dtype=dtype, kind="wish", distributed=False).fit(d, init=x0)
for dtype in ("float64", "float32"):
    fits(5, dtype)
    for threads in (1, 2, 4, 8):
        per = 200
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as pool:
            list(pool.map(lambda _: fits(per, dtype), range(threads)))
        dt = time.perf_counter() - t0
        print("N=%d %s K=%d: %d threads  %.0f fits/s  (%.2f ms per fit and thread)" % (n, dtype, K, threads, threads * per / dt, dt / per * 1e3))
