"""Throughput of the ContactMap stage on the resident matrix (A2/A3/A4 + the hand-over to
the solver), d = n_bins + 1 = 24,927 (chr1 at 10 kb) unless sizes are given.  Wall clock
around each C-ABI call (each call synchronises), inputs already in HBM except the triples
and the KR vectors, which are what the call takes.  Algorithmic bytes:
  scatter    24 B per triple read + 16 B written (two cells), + d^2 * 8 B zero fill
  normalize  8 B read + 16 B written per upper pair  (d^2/2 pairs)
  marginals  8 B per element (one pass over the matrix)
  filter     marginals + 8 B read + 8 B written per kept element
  pack       8 B read per upper pair + 4 B written (fp32 units)"""
import os, sys, time
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blueberry_amd as bb
from blueberry_amd.solver import HipEngine

def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
        best = min(best, dt)
    return best, r

def warm():
    """every kernel of the stage once on a small map: the first launch of a kernel loads
    its code object (milliseconds), which is not what the lines below are about"""
    n = 300
    rng = numpy.random.default_rng(1)
    tr = numpy.stack([rng.integers(0, n, 4000) * 10.0, rng.integers(0, n, 4000) * 10.0, 1.0 + rng.random(4000)], 1)
    c = bb.ContactMap.from_triples(tr, 10, n, KRnorm=1.0 + rng.random(n), KRexpected=1.0 + rng.random(n))
    c.normalize(); c.marginals(); c.eigenvector(); c.filter(0.0)
    c2 = bb.ContactMap.from_triples(tr, 10, n); c2.correlation()
warm()

for n_bins in [int(a) for a in sys.argv[1:]] or [24926]:
    d = n_bins + 1
    rng = numpy.random.default_rng(0)
    nnz = min(40_000_000, d * 400)
    bi = rng.integers(0, n_bins, nnz)
    bj = numpy.minimum(n_bins - 1, bi + rng.geometric(0.002, nnz))
    res = 10000
    triples = numpy.stack([bi * float(res), bj * float(res), rng.integers(1, 500, nnz).astype(float)], 1)
    kr = 0.5 + rng.random(n_bins); ke = 50.0 / (1.0 + numpy.arange(n_bins)) + 0.1
    print("== n_bins %d (matrix %.2f GB fp64), %d triples" % (n_bins, d * d * 8 / 1e9, nnz))
    t, cm = timed(lambda: bb.ContactMap.from_triples(triples, res, n_bins, KRnorm=kr, KRexpected=ke), 2)
    print("ContactMap.from_triples, whole call (host prep + create + scatter + regions)  %.1f ms" % (t * 1e3))
    from blueberry_amd import _lib
    from blueberry_amd.datatypes import _DeviceMatrix
    cols = numpy.ascontiguousarray(triples.T)
    dm = _DeviceMatrix(d, 0)
    t, _ = timed(lambda: _lib.check(dm._lib.bb_cm_scatter(dm._h, _lib.as_f64_ptr(cols), nnz, res), "bb_cm_scatter"))
    dm.close()
    print("scatter: the bb_cm_scatter call (H2D of the triples + zero fill + 2 passes)  %.1f ms  -> %.1f GB/s of %.2f GB"
          % (t * 1e3, (nnz * 40 + d * d * 8) / t / 1e9, (nnz * 40 + d * d * 8) / 1e9))
    def norm():
        cm._KRnorm, cm._KRexpected = kr, ke
        cm.normalize()
    t, _ = timed(norm)
    pairs = d * (d + 1) // 2
    print("normalize (in place)                                 %.2f ms  -> %.0f GB/s algorithmic (24 B per upper pair), %.1f %% of 8 TB/s"
          % (t * 1e3, pairs * 24 / t / 1e9, pairs * 24 / t / 8e12 * 100))
    t, _ = timed(cm.marginals)
    print("marginals (column sums in numpy's order, D2H of d)   %.2f ms  -> %.0f GB/s, %.1f %% of 8 TB/s"
          % (t * 1e3, d * d * 8 / t / 1e9, d * d * 8 / t / 8e12 * 100))
    lr = 1.0 / (2 * d)
    eng = HipEngine(d, "float32")
    t, _ = timed(lambda: eng.set_wish_from_cm(cm._resident(), "counts", 3.0))
    print("pack into the solver's units, device to device       %.2f ms  -> %.0f GB/s algorithmic (12 B per upper pair)"
          % (t * 1e3, pairs * 12 / t / 1e9))
    eng.close()
    # eigenvector (Lanczos over the resident matrix) and correlation (fp64 MFMA Gram kernel)
    try:
        cm.eigenvector(max_matvecs=2)
    except bb.EigenNoConvergence:
        pass
    #                             # the product's scratch is made once per handle
    for label, env in (("upper triangle, one pass serves both ends", None), ("both triangles (round 2)", "1")):
        if env:
            os.environ["BB_CM_SYMV_FULL"] = env
        t0 = time.perf_counter(); v = cm.eigenvector(); t = time.perf_counter() - t0
        os.environ.pop("BB_CM_SYMV_FULL", None)
        print("eigenvector [%s]: %d matrix-vector products, residual %.1e    %.1f ms  -> %.0f GB/s over the products (8 B per PAIR each: %.1f %% of 8 TB/s)"
              % (label, cm.eigen_matvecs_, cm.eigen_residual_, t * 1e3, cm.eigen_matvecs_ * pairs * 8 / t / 1e9,
                 cm.eigen_matvecs_ * pairs * 8 / t / 8e12 * 100))
    x = numpy.random.default_rng(2).standard_normal(d); y = numpy.empty(d)
    for label, env in (("upper triangle", None), ("both triangles", "1")):
        if env:
            os.environ["BB_CM_SYMV_FULL"] = env
        t, _ = timed(lambda: _lib.check(cm._resident()._lib.bb_cm_symv(cm._resident()._h, _lib.as_f64_ptr(x), _lib.as_f64_ptr(y)), "symv"), 5)
        os.environ.pop("BB_CM_SYMV_FULL", None)
        print("  one product through bb_cm_symv [%s] (H2D + product + D2H of d doubles)  %.3f ms" % (label, t * 1e3))
    cc = bb.ContactMap.from_matrix(numpy.zeros((1, 1)))      # a copy to turn into its correlation
    cc._host, cc._dev, cc.n_bins = None, type(cm._resident()).from_host(cm.to_host(), 0), n_bins
    nt = (d + 127) // 128; pairs_t = nt * (nt + 1) // 2; ldx = (d + 15) // 16 * 16
    for tag in ("first call of the handle (allocates its scratch)", "second call"):
        if tag.startswith("second"):
            _lib.check(cc._dev._lib.bb_cm_upload(cc._dev._h, _lib.as_f64_ptr(cm.to_host()), d), "upload")
        t0 = time.perf_counter(); cc.correlation(); t = time.perf_counter() - t0
        gram_ms = pairs_t * 2.0 * 128 * 128 * ldx / (cc.correlation_tflops_ * 1e12) * 1e3
        print("correlation, %s: Gram kernel %.1f TFLOP/s fp64 (%.0f %% of the 78.6 TFLOP/s matrix peak, %.1f ms), whole call %.1f ms = %.2f x the kernel"
              % (tag, cc.correlation_tflops_, cc.correlation_tflops_ / 78.6 * 100, gram_ms, t * 1e3, t * 1e3 / gram_ms))
    del cc
    thr = float(numpy.median(cm.marginals()))
    t0 = time.perf_counter(); cm.filter(thr); t = time.perf_counter() - t0
    dn = cm.shape[0]
    print("filter at the median marginal (-> %d bins, in place)   %.2f ms  -> %.0f GB/s algorithmic (d^2 read + 2 x (read + write) of the kept)"
          % (dn, t * 1e3, (d * d * 8 + dn * dn * 32) / t / 1e9))
    t, m = timed(cm.to_host, 1)
    print("fetch of the filtered matrix (D2H, %.2f GB)           %.1f ms  -> %.1f GB/s (PCIe)"
          % (dn * dn * 8 / 1e9, t * 1e3, dn * dn * 8 / t / 1e9))
