"""A genome-wide loop holds one ContactMap at a time: what does creating the next one cost when
the previous one (of another size) has just been freed?  bb_cm_create = hipMalloc + zero fill
of d*d doubles; sizes = the 23 hg19 chromosomes at 10 kb, in genome order, each destroyed
before the next is made.  (Round 4: a per-device cache of freed blocks was built on top of
this and measured with it -- docs/EXPERIMENTS.md round 4 item 5 -- and not kept.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from blueberry_amd import _lib
from blueberry_amd.utils import genome_boundaries
L = _lib.load()
sizes = [int(v) + 1 for v in numpy.diff(genome_boundaries())[:23]]
for rep in range(2):
    tot = 0.0
    line = []
    for d in sizes:
        h = _lib.c_void_p()
        t = time.perf_counter()
        _lib.check(L.bb_cm_create(h, d, 0), "create")
        dt = (time.perf_counter() - t) * 1e3
        L.bb_cm_destroy(h)
        tot += dt
        line.append("%.0f" % dt)
    print("pass %d: create ms per chromosome: %s; total %.0f ms" % (rep + 1, " ".join(line), tot))
