"""Does a block of iterations run at a different speed (a) from the noisy start than
from nearly converged coordinates, (b) with the timing events on?  (DVFS: the clock the
chip holds depends on what the data make the VALU do.)"""
import os, sys, time
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blueberry_amd.solver import HipEngine
for n in [int(a) for a in sys.argv[1:]] or [24926, 50000]:
    xs = numpy.cumsum(numpy.random.default_rng(0).standard_normal((n, 3)), axis=0)
    xs -= xs.mean(0)
    x0 = xs + 0.5 * numpy.random.default_rng(1).standard_normal(xs.shape)
    e = HipEngine(n, "float32")
    e.set_wish_from_coords(xs)
    lr = 1.0 / (2 * n)
    def block(k=50):
        e.sync(); t0 = time.perf_counter(); e.iterate(k, lr); e.sync()
        return (time.perf_counter() - t0) / k * 1e3
    e.set_coords(x0); e.iterate(200, lr); e.sync()           # settle the clocks
    out = []
    for rep in range(3):
        e.set_coords(x0); e.iterate(5, lr)
        a = block()                                           # noisy start, no events
        b = block(); c = block()                              # later: converging
        e.set_coords(x0); e.iterate(5, lr); e.set_timing(8)
        d = block(); tm = e.timing(); e.set_timing(0)         # noisy start, events on
        e.set_coords(xs); e.iterate(5, lr)
        f = block()                                           # at the solution: residuals ~ 0
        out.append("start %.4f  then %.4f %.4f | start+events %.4f (kernel by events %.4f) | at solution %.4f"
                   % (a, b, c, d, tm["grad_ms"], f))
    print("N=%d ms/step:" % n)
    for o in out:
        print("   " + o)
    e.close()
