"""Unit-by-unit time line of one stress_grad_kernel launch (diagnostic build).

    tools/build_variant.sh UTRACE -DBB_UNIT_TRACE
    BB_LIB=$PWD/tools/variants/libabl_UTRACE.so python tools/unit_trace.py [bins ...]

Every wave keeps the time at the top of each of its units in LDS (one s_memrealtime and one
ds_write per unit; the unit loop is the product's: same wait counts, nothing peeled) and
writes them out when it has finished.  Prints, per size: when the waves start, how long the
prologue takes, the duration of unit 0, 1, 2, ... (median over the waves), the steady
per-unit time, and how far apart the waves end.  BB_WAVES_PER_CU etc. apply."""
import ctypes
import os
import sys

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blueberry_amd import _lib
from blueberry_amd.solver import HipEngine

sizes = [int(a) for a in sys.argv[1:]] or [17700]
lib = _lib.load()
wt = lib.bb_solver_debug_wave_trace
wt.restype = ctypes.c_int
wt.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int64,
               ctypes.POINTER(ctypes.c_int64)]
ut = lib.bb_solver_debug_unit_trace
ut.restype = ctypes.c_int
ut.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int64,
               ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
P = ctypes.POINTER(ctypes.c_uint64)


def pct(a, scale=0.01):
    return "min %6.2f  p10 %6.2f  med %6.2f  p90 %6.2f  max %6.2f" % tuple(
        scale * numpy.percentile(a, p) for p in (0, 10, 50, 90, 100))


for n in sizes:
    xs = numpy.cumsum(numpy.random.default_rng(0).standard_normal((n, 3)), axis=0)
    e = HipEngine(n, os.environ.get("BB_TRACE_DTYPE", "float32"))
    e.set_wish_from_coords(xs)
    e.set_coords(xs + 0.5)
    e.iterate(300, 1.0 / (2 * n))
    e.sync()
    for rep in range(int(os.environ.get("BB_TRACE_REPS", "1"))):
        brief = rep > 0
        if brief:
            e.iterate(7 + rep, 1.0 / (2 * n))
            e.sync()
        nw, slots = ctypes.c_int64(), ctypes.c_int64()
        ut(e._h, None, 0, nw, slots)
        if slots.value == 0:
            raise SystemExit("not a -DBB_UNIT_TRACE build (set BB_LIB)")
        w8 = numpy.zeros(8 * nw.value, dtype=numpy.uint64)
        _lib.check(wt(e._h, w8.ctypes.data_as(P), w8.size, nw))
        u = numpy.zeros(slots.value * nw.value, dtype=numpy.uint64)
        _lib.check(ut(e._h, u.ctypes.data_as(P), u.size, nw, slots))
        w8 = w8.reshape(-1, 8).astype(numpy.int64)
        u = u.reshape(-1, slots.value).astype(numpy.int64)
        lay = e.layout()
        t_start, t_loop_end, t_end = w8[:, 0], w8[:, 2], w8[:, 3]
        xcc = (w8[:, 4] >> 32) & 0xF
        nu = (u > 0).sum(axis=1) - 1            # units stamped per wave (slot n = end of the last)
        have = nu >= 2
        base = t_start[have].min()
        print("== N=%d %s: %d waves (%d with units), %.1f units per wave, waves per CU %s"
              % (n, e.dtype, nw.value, have.sum(), (lay["u_end"] - lay["u_begin"]) / float(nw.value),
                 os.environ.get("BB_WAVES_PER_CU", "default")))
        print("  launch span (first start -> last end)     %.2f us" % (0.01 * (t_end[have].max() - base)))
        print("  wave start after the first                " + pct(t_start[have] - base))
        print("  prologue: start -> top of unit 0          " + pct(u[have, 0] - t_start[have]))
        maxu = int(nu[have].min())
        dur = numpy.diff(u[have, :maxu + 1], axis=1)      # (waves, maxu)
        for k in ([] if brief else list(range(min(12, maxu)))):
            print("  unit %2d                                   %s" % (k, pct(dur[:, k])))
        if maxu > 16:
            steady = dur[:, 12:maxu]
            print("  units 12..%d (steady), per unit            %s" % (maxu - 1, pct(steady.mean(axis=1))))
            b = numpy.median(steady.mean(axis=1))
            first12 = dur[:, :12].sum(axis=1)
            print("  excess of the first 12 units over steady  " + pct(first12 - 12 * b))
        print("  epilogue: loop end -> wave end            " + pct(t_end[have] - t_loop_end[have]))
        print("  wave end after the first start            " + pct(t_end[have] - base))
        print("  idle tail: last end - own end             " + pct(t_end[have].max() - t_end[have]))
        # the same by workgroup class b % 8 (blocks of one class share an XCD under round-robin
        # placement): is the class -> XCD map the same from launch to launch?
        wpb = int(os.environ.get("BB_TRACE_WPB", "8" if nw.value >= 2048 else "4"))
        cls = (numpy.arange(nw.value) // wpb) % 8
        for c in range(8):
            m = have & (cls == c)
            if m.any():
                ids = numpy.unique(xcc[m])
                print("  class %d: xcc ids %s, ends med %.2f max %.2f us, steady per unit med %.3f" % (
                    c, list(ids), 0.01 * numpy.median(t_end[m] - base), 0.01 * (t_end[m].max() - base),
                    0.01 * numpy.median(numpy.diff(u[m, :maxu + 1], axis=1)[:, min(12, maxu - 1):].mean(axis=1))))
        # the clock each XCD ran at: shader-clock ticks (s_memtime, slots 5 / 6) over the
        # 100-MHz ticks (slot 7 / 2) of the same stretch
        dclk = (w8[:, 6] - w8[:, 5]).astype(numpy.float64)
        drt = (t_loop_end - w8[:, 7]).astype(numpy.float64)
        mhz = numpy.where(drt > 0, dclk / numpy.maximum(drt, 1) * 100.0, 0.0)
        for x in ([] if brief else range(8)):
            m = have & (xcc == x)
            if m.any():
                print("  xcd %d: shader clock over the loop med %.0f MHz (p10 %.0f, p90 %.0f)" % (
                    x, numpy.median(mhz[m]), numpy.percentile(mhz[m], 10), numpy.percentile(mhz[m], 90)))
        for x in ([] if brief else range(8)):
            m = have & (xcc == x)
            if m.any():
                print("  xcd %d: %4d waves, loop start med %.2f, ends med %.2f max %.2f us" % (
                    x, m.sum(), 0.01 * numpy.median(u[m, 0] - base),
                    0.01 * numpy.median(t_end[m] - base), 0.01 * (t_end[m].max() - base)))
        sys.stdout.flush()
    e.close()
