"""Where a stress_grad_kernel launch spends its time, wave by wave (diagnostic build).

    BB_LIB=$PWD/tools/variants/libabl_TRACE.so python tools/wave_trace.py [bins ...]

libabl_TRACE.so = the product sources + -DBB_WAVE_TRACE (bb_ablate.h): every wave leaves
eight stamps (10-ns ticks): start, first unit done, last unit consumed, end, where it ran,
first load / whole window / coordinates landed.
Prints, per problem size, for the LAST of a few warm launches: the launch's span, how far
apart the waves start and finish, the prologue (start -> first unit done), the steady
per-unit time, the epilogue, and the same per XCD."""
import ctypes, os, sys
import numpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blueberry_amd import _lib
from blueberry_amd.solver import HipEngine

sizes = [int(a) for a in sys.argv[1:]] or [17700, 24926, 50000]
lib = _lib.load()
fn = lib.bb_solver_debug_wave_trace
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int64,
               ctypes.POINTER(ctypes.c_int64)]
for n in sizes:
    xs = numpy.cumsum(numpy.random.default_rng(0).standard_normal((n, 3)), axis=0)
    e = HipEngine(n, "float32")
    e.set_wish_from_coords(xs)
    e.set_coords(xs + 0.5)
    e.iterate(30, 1.0 / (2 * n))
    e.sync()
    e.set_timing(1)
    e.iterate(5, 1.0 / (2 * n))
    e.sync()
    tm = e.timing()
    if os.environ.get("BB_TRACE_BACK_TO_BACK"):
        # the stamps then come from the LAST of several sweeps launched with no other
        # kernel in between (same code still in the instruction cache?)
        rep = lib.bb_solver_debug_grad_repeat
        rep.restype = ctypes.c_int
        rep.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _lib.check(rep(e._h, int(os.environ["BB_TRACE_BACK_TO_BACK"])))
        print("(stamps: last of %s back-to-back sweep launches)" % os.environ["BB_TRACE_BACK_TO_BACK"])
    nw = ctypes.c_int64()
    fn(e._h, None, 0, nw)
    buf = numpy.zeros(8 * nw.value, dtype=numpy.uint64)
    _lib.check(fn(e._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), buf.size, nw))
    t = buf.reshape(-1, 8)
    lay = e.layout()
    units = (lay["u_end"] - lay["u_begin"]) / float(nw.value)
    t0 = t[:, 0].astype(numpy.int64); t1 = t[:, 1].astype(numpy.int64)
    t2 = t[:, 2].astype(numpy.int64); t3 = t[:, 3].astype(numpy.int64)
    xcc = (t[:, 4] >> numpy.uint64(32)).astype(numpy.int64) & 0xF
    base = t0.min()
    us = lambda a: a * 0.01
    q = lambda a: "min %.2f  p10 %.2f  med %.2f  p90 %.2f  max %.2f" % tuple(
        us(numpy.percentile(a, p)) for p in (0, 10, 50, 90, 100))
    print("== N=%d: %d waves, %.1f units per wave; kernel (HIP events) %.1f us"
          % (n, nw.value, units, tm["grad_ms"] * 1e3))
    print("  launch span (first start -> last end)   %.2f us" % us(t3.max() - base))
    print("  wave start after the first start        " + q(t0 - base))
    t5 = t[:, 5].astype(numpy.int64); t6 = t[:, 6].astype(numpy.int64); t7 = t[:, 7].astype(numpy.int64)
    print("  prologue: start -> first unit done      " + q(t1 - t0))
    print("     start -> first 1 KiB of matrix here  " + q(t5 - t0))
    print("     start -> whole 8-KiB window here     " + q(t6 - t0))
    print("     start -> coordinates here            " + q(t7 - t0))
    print("     coordinates here -> first unit done  " + q(t1 - t7))
    print("  steady: per unit after the first        " + q((t2 - t1) / max(units - 1, 1)))
    print("  epilogue: last unit consumed -> end     " + q(t3 - t2))
    print("  wave end after the first start          " + q(t3 - base))
    print("  idle tail: last end - own end           " + q(t3.max() - t3))
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print("  xcd %d: %4d waves, ends med %.2f max %.2f us, per-unit med %.3f us"
              % (x, m.sum(), us(numpy.median(t3[m] - base)), us((t3[m] - base).max()),
                 us(numpy.median((t2[m] - t1[m]) / max(units - 1, 1)))))
    # strip crossings per wave (a crossing = write the column partials, load the next
    # strip's coordinates, drain the window): does a wave's end time follow them?
    vw, upt = lay["vw"], lay["units_per_tile"]
    ntile_b = lay["n_blocks"]
    tile_J = numpy.concatenate([numpy.full(J + 1, J) for J in range(ntile_b)])
    nloc = lay["u_end"] - lay["u_begin"]
    q_, r_ = divmod(nloc, nw.value)
    wv_ = numpy.arange(nw.value)
    ua_ = wv_ * q_ + numpy.minimum(wv_, r_)
    ub_ = ua_ + q_ + (wv_ < r_)
    Ja = tile_J[(lay["u_begin"] + ua_) // upt]
    Jb = tile_J[(lay["u_begin"] + ub_ - 1) // upt]
    cross = Jb - Ja
    dur = t3 - t0
    for c in sorted(set(cross.tolist())):
        m = cross == c
        print("  waves crossing %d strip boundaries: %4d, duration med %.2f us (p10 %.2f, p90 %.2f), units %.1f"
              % (c, m.sum(), us(numpy.median(dur[m])), us(numpy.percentile(dur[m], 10)),
                 us(numpy.percentile(dur[m], 90)), float(numpy.mean((ub_ - ua_)[m]))))
    # who shares a SIMD with whom: HW_ID = wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
    hw = (t[:, 4] & numpy.uint64(0xFFFFFFFF)).astype(numpy.int64)
    simd_key = (xcc << 16) | (hw & 0xFFF0)
    per_unit = (t2 - t1) / max(units - 1, 1)
    groups = {}
    for w in range(t.shape[0]):
        groups.setdefault(int(simd_key[w]), []).append(w)
    sizes_ = sorted(set(len(g) for g in groups.values()))
    print("  waves per (xcd, se, sh, cu, simd):", {k: sum(1 for g in groups.values() if len(g) == k) for k in sizes_})
    pairs = [g for g in groups.values() if len(g) == 2]
    if pairs:
        fast = numpy.array([min(per_unit[g[0]], per_unit[g[1]]) for g in pairs])
        slow = numpy.array([max(per_unit[g[0]], per_unit[g[1]]) for g in pairs])
        older_fast = numpy.mean([per_unit[g[0] if t0[g[0]] <= t0[g[1]] else g[1]] <=
                                 per_unit[g[1] if t0[g[0]] <= t0[g[1]] else g[0]] for g in pairs])
        lowslot_fast = numpy.mean([per_unit[g[0] if (hw[g[0]] & 15) <= (hw[g[1]] & 15) else g[1]] <=
                                   per_unit[g[1] if (hw[g[0]] & 15) <= (hw[g[1]] & 15) else g[0]] for g in pairs])
        print("  SIMD pairs: %d; per-unit us of the faster partner %s" % (len(pairs), q(fast * 100)))
        print("              per-unit us of the slower partner %s" % q(slow * 100))
        print("              slower / faster %s   (x100)" % q(slow / fast * 100 * 100))
        print("              the partner that started first is the faster one in %.0f %% of the pairs; "
              "the lower wave slot in %.0f %%" % (100 * older_fast, 100 * lowslot_fast))
    e.close()
