"""Wall clock of the whole resident pipeline at chr1@10kb size: triples -> ContactMap
(scatter on the device) -> normalize -> filter -> StructureSolver.fit (device-to-device
pack, spectral start on the device, K iterations) -> coordinates on the host.  Second run
of the same process as well: the first pays the first touch of every fresh allocation."""
import os, sys, time
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blueberry_amd as bb

n_bins = int(sys.argv[1]) if len(sys.argv) > 1 else 24926
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = numpy.random.default_rng(0)
nnz = n_bins * 400
bi = rng.integers(0, n_bins, nnz)
bj = numpy.minimum(n_bins - 1, bi + rng.geometric(0.002, nnz))
res = 10000
triples = numpy.stack([bi * float(res), bj * float(res), rng.integers(1, 500, nnz).astype(float)], 1)
kr = 0.5 + rng.random(n_bins); ke = 50.0 / (1.0 + numpy.arange(n_bins)) + 0.1
for run in (1, 2):
    t = [time.perf_counter()]
    cm = bb.ContactMap.from_triples(triples, res, n_bins, KRnorm=kr, KRexpected=ke); t.append(time.perf_counter())
    cm.normalize(); t.append(time.perf_counter())
    cm.filter(0.0); t.append(time.perf_counter())
    s = bb.StructureSolver(n_iter=K, dtype="float32").fit(cm); t.append(time.perf_counter())
    d = numpy.diff(t) * 1e3
    print("run %d: from_triples %.0f ms, normalize %.1f ms, filter %.1f ms (-> %d bins), fit(K=%d, seeded random start) %.0f ms; total %.0f ms; "
          "stress %.3e -> %.3e" % (run, d[0], d[1], d[2], cm.shape[0], K, d[3], sum(d), s.stress_[0], s.stress_[-1]))
    del cm, s
# (after both runs of the pipeline: an allocation in between would change which blocks the
# runtime hands back to the second run, and a FRESH 5-GB block costs 0.2-0.35 s to first touch)
for run in (1, 2):
    # the same input without the dense matrix: fit_triples (bins, cleans and scatters the
    # triples on the device into the occupied tiles only)
    t0 = time.perf_counter()
    s2 = bb.StructureSolver(n_iter=K, dtype="float32").fit_triples(triples, res, n_bins, KRnorm=kr, KRexpected=ke)
    print("run %d: fit_triples(K=%d), whole call %.0f ms; stress %.3e -> %.3e"
          % (run, K, (time.perf_counter() - t0) * 1e3, s2.stress_[0], s2.stress_[-1]))
    del s2
