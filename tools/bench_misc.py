"""K1 and the two small helpers end to end (host vector in, result out), each checked and
timed against a plain numpy statement of the same function on this host (vectorised numpy,
one core; the oracle is not used outside tests/ and bench.py's cpu_baseline)."""
import os, sys, time
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blueberry_amd as bb
def best(fn, reps=5):
    t = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); t = min(t, time.perf_counter() - t0)
    return t, r
def band_numpy(r, lo=25000, hi=10000000):
    # sorted input: for every i, the j < i with lo <= r[i] - r[j] <= hi form one interval
    a = numpy.searchsorted(r, r - hi, side="left")
    b = numpy.searchsorted(r, r - lo, side="right")
    return int(numpy.maximum(numpy.minimum(b, numpy.arange(r.size)) - a, 0).sum())
print("count_band_regions (K1): pairs = N(N-1)/2, input 8 B per bin")
for n in (1000, 24926, 50000, 309568):
    r = numpy.arange(n) * 10000.0 + 5000.0
    tg, got = best(lambda: bb.count_band_regions(r))
    os.environ["BB_BAND_SORTED"] = "0"          # the double loop, which unsorted input takes
    tb, got_b = best(lambda: bb.count_band_regions(r))
    del os.environ["BB_BAND_SORTED"]
    tc, want = best(lambda: band_numpy(r), 1)
    assert got == got_b == want, (got, got_b, want)
    print("  N=%-7d GPU %.3f ms end to end on the sorted path (two binary searches per row), %.3f ms "
          "on the double loop (%.2f Tpair/s over all N(N-1)/2 pairs)   numpy searchsorted (sorted "
          "input only) %.2f ms" % (n, tg * 1e3, tb * 1e3, n * (n - 1) / 2 / tb / 1e12, tc * 1e3))
print("benjamini_hochberg: 16 B per element (8 read + 8 written)")
for d in (10**5, 10**7, 5 * 10**7):
    p = numpy.sort(numpy.random.default_rng(0).random(d) ** 3)
    tg, q = best(lambda: bb.benjamini_hochberg(p, 3 * d), 3)
    tc, qc = best(lambda: numpy.maximum.accumulate(numpy.minimum(p * (3 * d) / numpy.arange(1, d + 1), 1.0)), 1)
    assert numpy.array_equal(q, qc)
    print("  d=%-9d GPU %.2f ms end to end (%.1f GB/s incl. PCIe both ways)   numpy %.1f ms"
          % (d, tg * 1e3, d * 16 / tg / 1e9, tc * 1e3))
print("downsample (5x5 max-pool): 4 B read per fine cell")
for n5 in (400, 2000, 5000):
    rng = numpy.random.default_rng(1)
    a = rng.random((5 * n5, 5 * n5), dtype=numpy.float32); b = numpy.zeros((n5, n5), numpy.float32)
    tg, g = best(lambda: bb.downsample(a, b, b.copy()), 3)
    def ref():
        m = a[:5 * (n5 - 1), :5 * (n5 - 1)].reshape(n5 - 1, 5, n5 - 1, 5).max(axis=(1, 3))
        out = b.copy(); out[:n5 - 1, :n5 - 1] = numpy.maximum(out[:n5 - 1, :n5 - 1], m); return out
    tc, c = best(ref, 1)
    assert numpy.array_equal(g, c)
    print("  n5=%-5d GPU %.2f ms end to end (%.1f GB/s incl. PCIe)   numpy %.1f ms"
          % (n5, tg * 1e3, a.nbytes / tg / 1e9, tc * 1e3))
