"""What a matrix-sized hipMalloc / hipFree costs on this box (bb_cm_create = stream +
hipMalloc + zero fill of d*d doubles; bb_cm_destroy = hipFree + stream destroy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blueberry_amd import _lib
L = _lib.load()
def create(d):
    h = _lib.c_void_p(); t = time.perf_counter()
    _lib.check(L.bb_cm_create(h, d, 0), "create"); return h, (time.perf_counter() - t) * 1e3
def destroy(h):
    t = time.perf_counter(); L.bb_cm_destroy(h); return (time.perf_counter() - t) * 1e3
for rep in range(4):
    for d in (24927, 12464):
        h, tc = create(d); td = destroy(h)
        print("d=%5d (%.2f GB): create %.2f ms, destroy %.2f ms" % (d, d * d * 8 / 1e9, tc, td))
# the filter's pattern: big one alive, allocate the small one, free the big one
hb, _ = create(24927)
hs, tc = create(12464); td = destroy(hb)
print("with the 4.97 GB matrix alive: create 1.24 GB %.2f ms; then destroy 4.97 GB %.2f ms" % (tc, td))
destroy(hs)
# plain hipMalloc / hipFree of bounce-buffer sizes, through the HIP runtime the library uses
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
for rep in range(2):
    for mib in (1, 8, 64, 256, 1024):
        p = ctypes.c_void_p(); t = time.perf_counter()
        rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(mib << 20)); tm = (time.perf_counter() - t) * 1e3
        t = time.perf_counter(); hip.hipFree(p); tf = (time.perf_counter() - t) * 1e3
        print("hipMalloc %4d MiB: %.3f ms (rc %d), hipFree %.3f ms" % (mib, tm, rc, tf))
# bb_cm_filter by itself on a random symmetric matrix
import numpy
d = 24927
h, _ = create(d)
rng = numpy.random.default_rng(0)
row = rng.random(d)
for i in range(0, d, 1024):     # upload in slabs: rows i..i+1024 of an outer-product matrix
    pass
m = numpy.outer(row[:4096], row[:4096])
hs, _ = create(4096)
_lib.check(L.bb_cm_upload(hs, _lib.as_f64_ptr(m), 4096), "upload")
keep = numpy.zeros(4096, dtype=numpy.uint8); dn = _lib.c_i64()
t = time.perf_counter()
_lib.check(L.bb_cm_filter(hs, float(numpy.median(m.sum(0))), dn, keep.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))), "filter")
print("bb_cm_filter d=4096 -> %d: %.3f ms" % (dn.value, (time.perf_counter() - t) * 1e3))
destroy(hs)
# d = 24,927: zero matrix except a diagonal band uploaded by scatter
n = 2_000_000
bi = rng.integers(0, d - 1, n); bj = numpy.minimum(d - 2, bi + rng.geometric(0.01, n))
cols = numpy.ascontiguousarray(numpy.stack([bi * 10.0, bj * 10.0, 1.0 + rng.random(n)], 1).T)
_lib.check(L.bb_cm_scatter(h, _lib.as_f64_ptr(cols), n, 10), "scatter")
sums = numpy.empty(d); _lib.check(L.bb_cm_marginals(h, _lib.as_f64_ptr(sums)), "marginals")
keep = numpy.zeros(d, dtype=numpy.uint8)
for rep in range(1):
    t = time.perf_counter()
    _lib.check(L.bb_cm_filter(h, float(numpy.median(sums)), dn, keep.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))), "filter")
    print("bb_cm_filter d=%d -> %d: %.3f ms" % (d, dn.value, (time.perf_counter() - t) * 1e3))
t = time.perf_counter()
_lib.check(L.bb_cm_filter(h, -1.0, dn, keep.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))), "filter")
print("bb_cm_filter d=%d keeping everything (sums + scan only): %.3f ms" % (dn.value, (time.perf_counter() - t) * 1e3))
destroy(h)
