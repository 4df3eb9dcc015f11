"""Ranks on separate streams of one process through both forms of the peer exchange, over
ragged sizes down to 3 bins, against the one-rank run (tolerance) and against each other
(bit for bit).  python tools/peer_fuzz.py"""
import os, sys
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BB_PEER_TIMEOUT_MS", "5000")
from blueberry_amd.solver import HipEngine
from tests.test_gpu_distributed import _peer_engines
from tests import _oracle

bad = 0
for dtype, tol in (("float32", 1e-5), ("float64", 1e-12)):
    for n in (3, 4, 17, 129, 512, 513, 1025, 2049, 4096, 4097, 5003):
        xs = _oracle.random_walk(n)
        w = _oracle.wish_from_coords(xs)
        x0 = _oracle.noisy_init(xs)
        lr, k = 1.0 / (2 * n), 6
        one = HipEngine(n, dtype); one.set_wish_dense(w, "wish", 3.0); one.set_coords(x0); one.set_momentum(0.25)
        one.iterate(k, lr); X1, h1 = one.get_coords(), one.stress_history(); one.close()
        ref = None
        for world in (2, 3):
            for fused in ("1", "0"):
                os.environ["BB_PEER_FUSED"] = fused
                engs = _peer_engines(world, n, dtype, w, x0, mu=0.25)
                for e in engs:
                    e.iterate_peer(k, lr)
                got = [(e.get_coords(), e.stress_history(), e.peer_status()) for e in engs]
                for e in engs:
                    e.close()
                ok = all(numpy.array_equal(g[0], got[0][0]) and numpy.array_equal(g[1], got[0][1]) for g in got)
                err = numpy.abs(got[0][0] - X1).max() / max(numpy.abs(X1).max(), 1e-300)
                herr = numpy.abs(got[0][1] / h1 - 1).max()
                same = True
                if fused == "0":
                    same = numpy.array_equal(got[0][0], prev[0]) and numpy.array_equal(got[0][1], prev[1])
                prev = got[0]
                flag = "" if (ok and err < tol and herr < tol and same) else "   <-- FAIL"
                bad += bool(flag)
                print("%s n=%5d world %d %s: ranks identical %s, forms identical %s, vs one rank %.1e / %.1e%s" % (
                    dtype, n, world, "one launch " if fused == "1" else "two launches", ok, same, err, herr, flag), flush=True)
print("FAILURES: %d" % bad)
sys.exit(1 if bad else 0)
