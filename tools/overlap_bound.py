"""VERDICT r2 #1 lever (a), bounded from above: the reduce of iteration k and sweep k + 1 as a
software pipeline on two streams (one iteration stale, timing only) against the plain loop.
Needs a trace build (tools/build_variant.sh UTRACE -DBB_UNIT_TRACE); run with
BB_LIB=$PWD/tools/variants/libabl_UTRACE.so python tools/overlap_bound.py [N ...]"""
import ctypes, os, sys
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blueberry_amd import _lib
from blueberry_amd.solver import HipEngine

lib = _lib.load()
fn = lib.bb_solver_debug_overlap_bound
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
               ctypes.POINTER(ctypes.c_double)]
for n in [int(a) for a in sys.argv[1:]] or [5000, 8000, 12000, 17700]:
    rng = numpy.random.default_rng(0)
    xs = numpy.cumsum(rng.standard_normal((n, 3)), axis=0)
    e = HipEngine(n, "float32")
    e.set_wish_from_coords(xs)
    e.set_coords(xs + 0.5 * rng.standard_normal(xs.shape))
    e.iterate(300, 1 / (2 * n)); e.sync()
    for rep in range(3):
        a, b = ctypes.c_double(), ctypes.c_double()
        _lib.check(fn(e._h, max(300, int(2e5 / (n / 1000) ** 2)), 1 / (2 * n), a, b), "overlap_bound")
        print("N=%6d: plain loop %.2f us per iteration, reduce k pipelined under sweep k+1 on a second stream %.2f us (%+.1f %%)" % (
            n, a.value, b.value, 100 * (b.value / a.value - 1)), flush=True)
    e.close()
