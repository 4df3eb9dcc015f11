"""What the HIP calls around a small solver cost on this box: stream create / destroy,
hipMalloc / hipFree of arena sizes, small pageable H2D copies, memsets."""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
def t(f, reps=20):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best * 1e6
hip.hipSetDevice(0)
p0 = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(p0), ctypes.c_size_t(1 << 20))
st = ctypes.c_void_p()
def sc():
    hip.hipStreamCreateWithFlags(ctypes.byref(st), 1)
print("hipStreamCreateWithFlags %.0f us" % t(sc, 5))
streams = []
def sc2():
    s = ctypes.c_void_p(); hip.hipStreamCreateWithFlags(ctypes.byref(s), 1); streams.append(s)
print("hipStreamCreateWithFlags (more) %.0f us" % t(sc2, 8))
def sd():
    hip.hipStreamDestroy(streams.pop())
print("hipStreamDestroy %.0f us" % t(sd, 8))
for mib in (1, 16, 64, 256):
    ptrs = []
    def ma():
        p = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(mib << 20)); ptrs.append(p)
    tm = t(ma, 6)
    def fr():
        hip.hipFree(ptrs.pop())
    tf = t(fr, 6)
    print("hipMalloc %3d MiB %.0f us, hipFree %.0f us" % (mib, tm, tf))
import numpy
h = numpy.zeros(4096, dtype=numpy.uint8)
d = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(1 << 24))
def cp():
    hip.hipMemcpyAsync(d, h.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(4096), 1, st)
print("hipMemcpyAsync 4 KiB pageable H2D %.0f us" % t(cp))
def ms():
    hip.hipMemsetAsync(d, 0, ctypes.c_size_t(1 << 24), st)
print("hipMemsetAsync 16 MiB (enqueue) %.0f us" % t(ms))
def sy():
    hip.hipStreamSynchronize(st)
hip.hipMemsetAsync(d, 0, ctypes.c_size_t(1 << 24), st)
print("hipStreamSynchronize after a 16 MiB memset %.0f us" % t(sy, 1))
print("hipStreamSynchronize idle %.0f us" % t(sy))
