"""Per-chromosome maps batched into one solver (StructureSolver.fit_many / bb_solver_set_maps)
against the same maps solved one after the other, device time per iteration by the host clock
around K enqueued iterations (inputs resident, synthetic wish distances generated on the
device).  Sizes: the 23 chromosomes chr1..chr22, chrX of hg19 at the given bin sizes.
    python tools/batch_timing.py [resolution_bp ...]      default 50000 100000"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy

from blueberry_amd.solver import HipEngine
from blueberry_amd.utils import genome_boundaries

K = 100
for res in [int(a) for a in sys.argv[1:]] or [50000, 100000]:
    sizes = [int(v) for v in numpy.diff(genome_boundaries(resolution=res))[:23]]
    walks = [numpy.cumsum(numpy.random.default_rng(q).standard_normal((n, 3)), axis=0)
             for q, n in enumerate(sizes)]
    # one after the other: every map in a solver of its own (what 23 fit() calls run)
    singles = []
    for n, xs in zip(sizes, walks):
        e = HipEngine(n, "float32")
        e.set_wish_from_coords(xs)
        e.set_coords(xs + 0.5)
        e.iterate(10, 1.0 / (2 * n))
        e.sync()
        singles.append(e)
    best_serial, per_map = 1e9, None
    for _ in range(3):
        t0 = time.perf_counter()
        ts = []
        for e, n in zip(singles, sizes):
            t1 = time.perf_counter()
            e.iterate(K, 1.0 / (2 * n))
            e.sync()
            ts.append((time.perf_counter() - t1) / K)
        dt = (time.perf_counter() - t0) / K
        if dt < best_serial:
            best_serial, per_map = dt, ts
    for e in singles:
        e.close()
    # all of them in one solver
    vw = 512
    off = [0]
    for n in sizes[:-1]:
        off.append(off[-1] + -(-n // vw) * vw)
    total = off[-1] + sizes[-1]
    ti, tj = [], []
    for o, n in zip(off, sizes):
        b0, b1 = o // vw, (o + n + vw - 1) // vw
        jj, ii = numpy.meshgrid(numpy.arange(b0, b1), numpy.arange(b0, b1))
        sel = ii <= jj
        ti.append(ii[sel]); tj.append(jj[sel])
    ti, tj = numpy.concatenate(ti), numpy.concatenate(tj)
    order = numpy.lexsort((ti, tj))
    tiles = (ti[order].astype(numpy.int32), tj[order].astype(numpy.int32))
    e = HipEngine(total, "float32", tiles=tiles)
    e.set_maps(off + [total], [1.0 / (2 * n) for n in sizes])
    xs = numpy.zeros((total, 3))
    for o, n, w in zip(off, sizes, walks):
        xs[o:o + n] = w
    e.set_wish_from_coords(xs)       # (pairs of different maps are not in the tile list)
    e.set_coords(xs + 0.5)
    e.iterate(10, 1.0)
    e.sync()
    best_batch = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        e.iterate(K, 1.0)
        e.sync()
        best_batch = min(best_batch, (time.perf_counter() - t0) / K)
    lay = e.layout()
    unit_mb = lay["n_units"] * 8192 / 1e6
    pair_mb = sum(n * (n - 1) // 2 for n in sizes) * 4 / 1e6
    e.close()
    big = max(per_map)
    print("hg19 at %d bp: 23 maps of %d..%d bins (%d bins in all); stored pairs %.0f MB, "
          "resident units %.0f MB (%d tiles of 512)" % (res, min(sizes), max(sizes), sum(sizes),
                                                       pair_mb, unit_mb, len(tiles[0])))
    print("  one after the other: %.1f us per iteration of all 23 (largest map alone %.1f us, "
          "smallest %.1f us)" % (best_serial * 1e6, big * 1e6, min(per_map) * 1e6))
    print("  one solver of 23 maps: %.1f us per iteration = %.2f x the largest map alone, "
          "%.2f x faster than one after the other; %.2f TB/s on the resident units, floor of "
          "the stored pairs at 8 TB/s %.1f us" % (best_batch * 1e6, best_batch / big,
                                                 best_serial / best_batch,
                                                 unit_mb / 1e6 / best_batch, pair_mb / 8.0))
