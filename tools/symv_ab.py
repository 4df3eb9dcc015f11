"""A/B of the symmetric matrix-vector product (bb_cm_symv kernels) on the resident chr1@10kb-sized
matrix: device time per product by events inside bb_cm_eigenvector is not exposed, so time
N products enqueued back to back through bb_cm_eigenvector's own loop -- the Lanczos cycle --
and one bb_cm_symv call.  Run under BB_LIB=... for the other build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import blueberry_amd as bb
from blueberry_amd import _lib

d = int(sys.argv[1]) if len(sys.argv) > 1 else 24927
rng = numpy.random.default_rng(0)
a = rng.random((d, d))
m = numpy.triu(a, 1); m = m + m.T + numpy.diag(rng.random(d)); del a
cm = bb.ContactMap.from_matrix(m)
ref = None
x = rng.standard_normal(d); y = numpy.empty(d)
h = cm._resident()
for rep in range(3):
    _lib.check(h._lib.bb_cm_symv(h._h, _lib.as_f64_ptr(x), _lib.as_f64_ptr(y)), "symv")
err = numpy.abs(y - m @ x).max() / numpy.abs(m @ x).max()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter(); v = cm.eigenvector(); t = time.perf_counter() - t0
    best = min(best, t)
pairs = d * (d + 1) // 2
print("%s d=%d: symv max rel err %.1e; eigenvector %d products %.2f ms = %.3f ms per product all in = %.2f TB/s on 8 B per pair"
      % (os.environ.get("BB_LIB", "product"), d, err, cm.eigen_matvecs_, best * 1e3, best * 1e3 / cm.eigen_matvecs_,
         cm.eigen_matvecs_ * pairs * 8 / best / 1e12))
