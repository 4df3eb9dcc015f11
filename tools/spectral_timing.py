"""Cost of the device-resident spectral start (bb_solver_spectral_init: 41 products of the
block power iteration + the N x 3 algebra between them) against 41 solver iterations of the
same map, on one rank: plain, and with the per-product sum going through the peer exchange
(a one-rank solver pushing to itself, both forms) as it does on several ranks.  Sizes default
to the pairs of a 1/8 share of N = 50,000 and of N = 61,914, and the whole N = 50,000.
    python tools/spectral_timing.py [n_bins ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy

from blueberry_amd.solver import HipEngine

sizes = [int(v) for v in sys.argv[1:]] or [17700, 21890, 50000]
for n in sizes:
    rng = numpy.random.default_rng(0)
    xs = numpy.cumsum(rng.standard_normal((n, 3)), axis=0)
    v0 = rng.standard_normal((n, 3))
    for name, fused in (("plain", None), ("peer, one launch", "1"), ("peer, two launches", "0")):
        if fused is not None:
            os.environ["BB_PEER_FUSED"] = fused
        e = HipEngine(n, "float32")
        e.set_wish_from_coords(xs)
        if fused is not None:
            import ctypes
            from blueberry_amd import _lib
            buf = ctypes.create_string_buffer(_lib.BB_PEER_HANDLE_BYTES)
            _lib.check(e._lib.bb_solver_peer_export(e._h, buf), "export")
            _lib.check(e._lib.bb_solver_peer_connect(e._h, buf.raw), "connect")
        e.spectral_init_device(40, v0)             # warm: allocations, first launches
        e.iterate(41, 1.0 / (2 * n))
        e.sync()
        best_sp = best_it = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            e.spectral_init_device(40, v0)
            best_sp = min(best_sp, time.perf_counter() - t0)
            t0 = time.perf_counter()
            e.iterate(41, 1.0 / (2 * n))
            e.sync()
            best_it = min(best_it, time.perf_counter() - t0)
        print("n=%6d %-18s spectral start (41 products) %.3f ms = %.1f us per product; 41 "
              "iterations %.3f ms = %.1f us each; ratio %.2f"
              % (n, name, best_sp * 1e3, best_sp / 41 * 1e6, best_it * 1e3, best_it / 41 * 1e6,
                 best_sp / best_it))
        if fused is None:
            # the stopping rule (spectral_tol = 1e-3, the product's default): the map is
            # complete and noise-free, so the second product already lies in span(V)
            best_tol, made = 1e9, None
            for _ in range(5):
                t0 = time.perf_counter()
                made = e.spectral_init_device(40, v0, tol=1e-3)[0] + 1
                best_tol = min(best_tol, time.perf_counter() - t0)
            print("n=%6d %-18s spectral start ended by spectral_tol=1e-3 after %d products: %.3f ms "
                  "(%.1fx less than all 41)" % (n, name, made, best_tol * 1e3, best_sp / best_tol))
        e.close()
