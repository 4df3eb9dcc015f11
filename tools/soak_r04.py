"""Leak check of the round-4 entry points: free device memory across 200 life cycles each of
fit_many (bb_solver_set_maps), fit_triples (bb_triples_*), the spectral start (with its stopping
rule), the degree pass and per-bin steps (bb_solver_degrees / _set_bin_steps) and
ContactMap.correlation (per-device scratch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
import torch
import blueberry_amd as bb
from tests import _oracle


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0] / 2**20


torch.zeros(1, device="cuda")
rng = numpy.random.default_rng(0)
sizes = [300, 700, 1100]
mats = [_oracle.wish_from_coords(_oracle.random_walk(n, seed=q)) for q, n in enumerate(sizes)]
n_bins, res = 900, 5000
bi = rng.integers(0, n_bins, 20000); bj = numpy.minimum(n_bins - 1, bi + rng.geometric(0.02, 20000))
tr = numpy.stack([bi * float(res), bj * float(res), rng.integers(1, 99, 20000).astype(float)], 1)


hole = numpy.triu(rng.random(mats[2].shape) < 0.6, 1)
holey = numpy.where(hole | hole.T, 0.0, mats[2])


def cycle():
    bb.StructureSolver(n_iter=3, dtype="float32", kind="wish").fit_many(mats)
    bb.StructureSolver(n_iter=3, dtype="float64").fit_triples(tr, res, n_bins)
    bb.StructureSolver(n_iter=2, dtype="float32", kind="wish", init="spectral").fit(mats[1])
    # (added late in the round) the degree pass + per-bin factors on every input kind, and
    # the spectral start run to its cap on a map with holes
    bb.StructureSolver(n_iter=3, dtype="float32", degree_steps=True).fit_triples(tr, res, n_bins)
    bb.StructureSolver(n_iter=3, dtype="float64", kind="wish", degree_steps=True,
                       init="spectral", spectral_iter=6).fit(holey)
    bb.StructureSolver(n_iter=3, dtype="float32", kind="wish", degree_steps=True).fit_many([holey, mats[0]])
    cm = bb.ContactMap.from_triples(tr, res, n_bins)
    cm.correlation()
    del cm


for _ in range(3):
    cycle()
bb._lib.load().bb_cm_release_scratch(0)
f0 = free_mb()
t0 = time.perf_counter()
for it in range(200):
    cycle()
bb._lib.load().bb_cm_release_scratch(0)
f1 = free_mb()
print("200 cycles of fit_many + fit_triples + spectral fit + degree steps + correlation in %.1f s: free device "
      "memory %.0f -> %.0f MiB (%+.1f)" % (time.perf_counter() - t0, f0, f1, f1 - f0))
assert f0 - f1 < 64, "device memory leaks"
print("soak r04 ok")
