"""World size 8 at BASELINE's full sizes on ONE GPU: eight ranks of this process (one solver,
one stream each; arenas connected directly), each holding its 1/8 share of the units, iterate
over the peer exchange (two-launch form: the form ranks that share a GPU use) -- the headline
map (N = 50,000 dense fp32) and config 5 (N = 309,568 block-sparse) -- and must end where ONE
rank ends (1e-5), bit-identical among themselves.  What this cannot show is time: eight
ranks on one chip take turns.  Start with GPU_MAX_HW_QUEUES >= 16 (tools/peer_sequence_fuzz.py
says why).

    GPU_MAX_HW_QUEUES=32 python tools/world8_rehearsal.py [dense|genome10kb ...]"""
import ctypes
import os
import sys
import time

import numpy

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("BB_PEER_FUSED", "0")
os.environ.setdefault("BB_PEER_TIMEOUT_MS", "60000")
from blueberry_amd import _lib                                            # noqa: E402
from blueberry_amd.solver import HipEngine, max_degree, tiles_from_blocks  # noqa: E402
from blueberry_amd.utils import genome_boundaries                          # noqa: E402
from tests import _oracle                                                  # noqa: E402

WORLD, K = 8, 4


def run(workload):
    if workload == "dense":
        n, tiles = 50000, None
        lr = 1.0 / (2 * n)
    else:
        n = 309568
        tiles, _ = tiles_from_blocks(n, genome_boundaries(n), 1000, "float32")
        lr = 1.0 / (2 * max_degree(n, tiles, "float32"))
    xs = _oracle.random_walk(n)
    x0 = _oracle.noisy_init(xs)
    one = HipEngine(n, "float32", tiles=tiles)
    one.set_wish_from_coords(xs)
    one.set_coords(x0)
    one.set_momentum(0.3)
    one.iterate(K, lr)
    X1, h1 = one.get_coords(), one.stress_history()
    n_units = one.layout()["n_units"]
    one.close()
    lib = _lib.load()
    engs = [HipEngine(n, "float32", rank=r, world=WORLD, tiles=tiles) for r in range(WORLD)]
    shares = [e.layout()["u_end"] - e.layout()["u_begin"] for e in engs]
    assert sum(shares) == n_units and max(shares) - min(shares) <= 1, shares
    blobs = []
    for e in engs:
        buf = ctypes.create_string_buffer(_lib.BB_PEER_HANDLE_BYTES)
        _lib.check(lib.bb_solver_peer_export(e._h, buf), "export")
        blobs.append(buf.raw)
    for e in engs:
        _lib.check(lib.bb_solver_peer_connect(e._h, b"".join(blobs)), "connect")
        e.set_wish_from_coords(xs)
        e.set_coords(x0)
        e.set_momentum(0.3)
    t0 = time.perf_counter()
    for _ in range(K):                                  # step by step: every rank's launch of
        for e in engs:                                  # an iteration is enqueued before anybody's next
            e.iterate_peer(1, lr)
    res = []
    for e in engs:
        assert e.peer_status() == 0
        res.append((e.get_coords(), e.stress_history()))
    dt = time.perf_counter() - t0
    for e in engs:
        e.close()
    ex = max(float(numpy.abs(X - X1).max() / numpy.abs(X1).max()) for X, _ in res)
    es = max(float(numpy.abs(h / h1 - 1).max()) for _, h in res)
    same = all(numpy.array_equal(X, res[0][0]) and numpy.array_equal(h, res[0][1]) for X, h in res[1:])
    print("%s: N=%d, %d units in 8 shares of %d-%d; %d iterations (momentum 0.3) on 8 ranks vs 1: "
          "coordinates %.1e, stress history %.1e, ranks bit-identical: %s (%.2f s for the 8 ranks on one chip)"
          % (workload, n, n_units, min(shares), max(shares), K, ex, es, same, dt), flush=True)
    return ex < 1e-5 and es < 1e-5 and same


if __name__ == "__main__":
    ok = all([run(w) for w in (sys.argv[1:] or ["dense", "genome10kb"])])
    print("world-8 rehearsal", "ok" if ok else "FAILED")
    sys.exit(0 if ok else 1)
