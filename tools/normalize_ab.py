import os, sys, time, numpy
sys.path.insert(0, os.getcwd())
import blueberry_amd as bb
n_bins = 24926; d = n_bins + 1
rng = numpy.random.default_rng(0)
nnz = d * 400
bi = rng.integers(0, n_bins, nnz); bj = numpy.minimum(n_bins - 1, bi + rng.geometric(0.002, nnz))
tr = numpy.stack([bi * 1e4, bj * 1e4, rng.integers(1, 500, nnz).astype(float)], 1)
kr = 0.5 + rng.random(n_bins); ke = 50.0 / (1.0 + numpy.arange(n_bins)) + 0.1
cm = bb.ContactMap.from_triples(tr, 10000, n_bins, KRnorm=kr, KRexpected=ke)
pairs = d * (d + 1) // 2
ref = None
for tile, wgs in ((128, 1), (64, 4), (64, 3), (64, 2), (128, 1), (64, 4), (32, 0)):
    os.environ["BB_CM_NORMALIZE_TILE"] = str(tile)
    if wgs: os.environ["BB_CM_NORMALIZE_WGS"] = str(wgs)
    best = 1e9
    for _ in range(4):
        cm._KRnorm, cm._KRexpected = kr, ke
        t0 = time.perf_counter(); cm.normalize(); best = min(best, time.perf_counter() - t0)
    print("tile %3d wgs/cu %d: %.3f ms  %.0f GB/s (%.1f %% of 8 TB/s)" % (tile, wgs, best * 1e3, pairs * 24 / best / 1e9, pairs * 24 / best / 8e10))
