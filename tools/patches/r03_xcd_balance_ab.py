"""Shares of the sweep by XCD (bb_solver_xcd_calibrate / _split): step time with equal
shares and with shares in proportion to the measured speeds, interleaved, one process.
    python tools/xcd_balance_ab.py [N ...]        BB_AB_DTYPE=float64 for the fp64 sweep
"""
import os, sys, time
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blueberry_amd.solver import HipEngine

dtype = os.environ.get("BB_AB_DTYPE", "float32")
sizes = [int(a) for a in sys.argv[1:]] or [17700, 24926, 50000]


def step_us(e, iters, lr):
    e.iterate(max(20, iters // 4), lr); e.sync()
    t0 = time.perf_counter()
    e.iterate(iters, lr); e.sync()
    return (time.perf_counter() - t0) / iters * 1e6


for n in sizes:
    rng = numpy.random.default_rng(0)
    xs = numpy.cumsum(rng.standard_normal((n, 3)), axis=0)
    e = HipEngine(n, dtype)
    e.set_wish_from_coords(xs)
    e.set_coords(xs + 0.5 * rng.standard_normal(xs.shape))
    lr = 1 / (2 * n)
    iters = max(200, int(3e5 / (n / 1000) ** 2))
    step_us(e, iters, lr)                                  # clocks up
    sp = e.xcd_calibrate(8)
    print("== N=%d %s: speeds with equal shares  %s  (spread %.1f %%)" % (
        n, dtype, " ".join("%.3f" % v for v in sp), 100 * (max(sp) - min(sp))), flush=True)
    print("   workgroup time by XCD, equal shares: %s us" % " ".join("%.1f" % v for v in e.xcd_wg_us))
    w = list(sp)
    for rnd in range(3):
        e.xcd_split(None); a = step_us(e, iters, lr)
        e.xcd_split(w); b = step_us(e, iters, lr)
        e.xcd_split(None); a2 = step_us(e, iters, lr)
        e.xcd_split(w); b2 = step_us(e, iters, lr)
        sp2 = e.xcd_calibrate(8)
        print("  round %d: equal %.2f %.2f us   by speed %.2f %.2f us   (%.1f %%)   residual speeds %s" % (
            rnd, a, a2, b, b2, 100 * ((b + b2) / (a + a2) - 1), " ".join("%.3f" % v for v in sp2)), flush=True)
        print("     workgroup time by XCD under the plan: %s us" % " ".join("%.1f" % v for v in e.xcd_wg_us))
        w = sp2      # speeds measured under the plan ARE the next weights (units / time)
    print("  plan: %s" % e.xcd_info(), flush=True)
    st = e.stress()
    e.xcd_split(None)
    print("  stress under the plan %.9e, with equal shares %.9e" % (st, e.stress()), flush=True)
    e.close()
