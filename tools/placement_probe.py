"""Does the kernel time depend on where hipMalloc puts the 5 GB buffer?"""
import os, sys, time
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blueberry_amd.solver import HipEngine
n = 50000
xs = numpy.cumsum(numpy.random.default_rng(0).standard_normal((n, 3)), axis=0)
x0 = xs + 0.5 * numpy.random.default_rng(1).standard_normal(xs.shape)
keep = []
for trial in range(8):
    e = HipEngine(n, "float32")
    e.set_wish_from_coords(xs)
    e.set_coords(x0)
    e.iterate(5, 1 / (2 * n)); e.sync()
    e.set_timing(True)
    e.iterate(30, 1 / (2 * n))
    t = e.timing()
    r = e.stream_read_ms(10)
    print("trial %d: kernel %.4f ms  read %.4f ms" % (trial, t["grad_ms"], r), flush=True)
    if trial % 2 == 0:
        keep.append(e)          # hold some allocations so later ones land elsewhere
    else:
        e.close()
