"""Leaks and long runs: (1) free device memory before and after hundreds of solver /
ContactMap / peer-arena life cycles, (2) one long run per iteration path (row-owner, unit
sweep, peer exchange in both forms) with the stress checked to keep falling.
python tools/soak.py"""
import os, sys, time
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BB_PEER_TIMEOUT_MS", "20000")
import torch
import blueberry_amd as bb
from blueberry_amd.solver import HipEngine
from tests.test_gpu_distributed import _peer_engines
from tests import _oracle


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0] / 2**20


def problem(n):
    xs = _oracle.random_walk(n)
    return xs, _oracle.wish_from_coords(xs), _oracle.noisy_init(xs)


torch.zeros(1, device="cuda")
xs, w, x0 = problem(2000)
for warm in range(3):
    s = bb.StructureSolver(n_iter=5, dtype="float32", kind="wish").fit(w, init=x0)
f0 = free_mb()
t0 = time.perf_counter()
for it in range(300):
    dt = "float32" if it % 2 else "float64"
    s = bb.StructureSolver(n_iter=10, dtype=dt, kind="wish").fit(w, init=x0)
    if it % 10 == 0:
        cm = bb.ContactMap.from_arrays("x", 1, 5000, numpy.array([[2500.0, 7500.0, 3.0]]), n_bins=400)
        cm.normalize() if False else None
        del cm
    if it % 25 == 0:
        engs = _peer_engines(2, 700, dt, w[:700, :700], x0[:700])
        for e in engs:
            e.iterate_peer(3, 1.0 / 1400)
        for e in engs:
            assert e.peer_status() == 0
            e.close()
f1 = free_mb()
for it in range(300):
    s = bb.StructureSolver(n_iter=10, dtype="float32" if it % 2 else "float64", kind="wish").fit(w, init=x0)
f2 = free_mb()
print("300 more plain fits: free device memory %.0f -> %.0f MiB (%+.1f)" % (f1, f2, f2 - f1), flush=True)
print("300 solver life cycles (+30 ContactMaps, +12 pairs of peer arenas) in %.1f s: free device memory %.0f -> %.0f MiB (%+.1f)" % (
    time.perf_counter() - t0, f0, f1, f1 - f0), flush=True)
assert f0 - f1 < 64, "device memory leaks"

for n, dtype, iters in ((963, "float64", 200000), (3000, "float32", 60000), (9000, "float32", 20000)):
    xs, w, x0 = problem(n)
    e = HipEngine(n, dtype); e.set_wish_dense(w, "wish", 3.0); e.set_coords(x0); e.set_momentum(0.3)
    t0 = time.perf_counter()
    done = 0
    while done < iters:
        k = min(50000, iters - done); e.iterate(k, 1.0 / (2 * n)); done += k
    h = e.stress_history(); dt_ = time.perf_counter() - t0
    ok = numpy.isfinite(h).all() and h[-1] <= h[0] and h.shape == (iters,)
    print("n=%5d %s: %d iterations in %.2f s (%.2f us each), stress %.3e -> %.3e  %s" % (
        n, dtype, iters, dt_, dt_ / iters * 1e6, h[0], h[-1], "ok" if ok else "FAIL"), flush=True)
    assert ok
    e.close()
for form in ("1", "0"):
    os.environ["BB_PEER_FUSED"] = form
    n, iters = 3000, 30000
    xs, w, x0 = problem(n)
    engs = _peer_engines(3, n, "float32", w, x0, mu=0.3)
    t0 = time.perf_counter()
    # (ranks of ONE process share its host thread: enqueue in turns and in small pieces --
    # a rank whose queue is full blocks the thread that would feed the ranks it waits for)
    for chunk in range(iters // 100):
        for e in engs:
            e.iterate_peer(100, 1.0 / (2 * n))
    hs = []
    for e in engs:
        assert e.peer_status() == 0
        hs.append((e.get_coords(), e.stress_history()))
    dt_ = time.perf_counter() - t0
    same = all(numpy.array_equal(hs[0][0], x) and numpy.array_equal(hs[0][1], h) for x, h in hs)
    print("peer exchange (%s), 3 ranks in one process, n=%d: %d iterations in %.2f s, ranks identical %s, stress %.3e -> %.3e" % (
        engs[0].peer_form(), n, iters, dt_, same, hs[0][1][0], hs[0][1][-1]), flush=True)
    assert same and hs[0][1][-1] <= hs[0][1][0]
    for e in engs:
        e.close()
print("soak ok")
