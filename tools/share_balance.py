"""Device time of EVERY rank's share at world size R (default 8), one share at a time on one
GPU: the sweep over the share's units plus the reduce that leaves the partial gradient in the
exchange buffer (bb_solver_grad: everything a rank does before a byte crosses xGMI), against
the whole map on one rank.  The slowest share bounds an R-GPU step from below; R x slowest
against the one-rank step is the scaling the partition allows before any exchange cost.

    python tools/share_balance.py [dense|genome10kb ...] [--world R]"""
import os
import sys
import time

import numpy

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from blueberry_amd.solver import HipEngine, max_degree, tiles_from_blocks  # noqa: E402
from blueberry_amd.utils import genome_boundaries                          # noqa: E402
from tests import _oracle                                                  # noqa: E402


def step_us(e, reps=60):
    for _ in range(10):
        e.grad()
    e.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            e.grad()
        e.sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e6


def split_us(e, reps=40):
    """(sweep, reduce) by HIP events around the two launches of bb_solver_grad."""
    e.set_timing(True)
    for _ in range(reps):
        e.grad()
    e.sync()
    t = e.timing()
    e.set_timing(False)
    return t["grad_ms"] * 1e3, t["reduce_ms"] * 1e3


def run(workload, world):
    if workload == "dense":
        n, tiles = 50000, None
    else:
        n = 309568
        tiles, _ = tiles_from_blocks(n, genome_boundaries(n), 1000, "float32")
    xs = _oracle.random_walk(n)
    x0 = _oracle.noisy_init(xs)
    one = HipEngine(n, "float32", tiles=tiles)
    one.set_wish_from_coords(xs)
    one.set_coords(x0)
    t1 = step_us(one, 20)
    one.close()
    ts, parts = [], []
    for r in range(world):
        e = HipEngine(n, "float32", rank=r, world=world, tiles=tiles)
        e.set_wish_from_coords(xs)
        e.set_coords(x0)
        ts.append(step_us(e))
        parts.append(split_us(e))
        e.close()
    ts = numpy.array(ts)
    print("%s N=%d: one rank %.1f us per step; the %d shares alone: %s us (min %.1f, mean %.1f, max %.1f); "
          "one rank / (%d x slowest share) = %.2f of linear before any exchange"
          % (workload, n, t1, world, " ".join("%.1f" % t for t in ts), ts.min(), ts.mean(), ts.max(),
             world, t1 / (world * ts.max())), flush=True)
    print("   by events, sweep + reduce per share: %s"
          % "  ".join("%.1f+%.1f" % p for p in parts), flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    world = int(sys.argv[sys.argv.index("--world") + 1]) if "--world" in sys.argv else 8
    args = [a for a in args if not a.isdigit()]
    for w in (args or ["dense", "genome10kb"]):
        run(w, world)
