"""Workload for a rocprofv3 --kernel-trace timeline (tools/timeline.sh): for each size, a
settling phase and then K plain iterations enqueued by one C call, with a small D2H copy
between sizes so that tools/timeline_parse.py can cut the trace into segments.
Usage: python3 tools/timeline.py [dtype] n_bins [n_bins ...]   (BB_* knobs apply)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from blueberry_amd.solver import HipEngine

args = sys.argv[1:]
dtype = "float32"
if args and args[0] in ("float32", "float64"):
    dtype = args.pop(0)
sizes = [int(v) for v in args] or [17700]
for n in sizes:
    xs = numpy.cumsum(numpy.random.default_rng(0).standard_normal((n, 3)), axis=0)
    e = HipEngine(n, dtype)
    e.set_wish_from_coords(xs)
    e.set_coords(xs + 0.5)
    k = max(200, min(4000, int(0.25 / (1e-6 * (10 + n * n / 2 * 4 / 6.0e6)))))
    e.iterate(k, 1.0 / (2 * n))      # ~0.25 s: clocks settle
    e.sync()
    e.get_coords()                   # a copy kernel / memcpy: segment marker
    e.iterate(400, 1.0 / (2 * n))
    e.sync()
    e.get_coords()
    e.close()
print("timeline done", sizes)
