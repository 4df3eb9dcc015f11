import os, sys, time, subprocess
code = r'''
import os, sys, time, numpy
sys.path.insert(0, os.getcwd())
from blueberry_amd.solver import HipEngine
for n, dtype in ((963, "float64"), (2500, "float64"), (6000, "float64"), (963, "float32")):
    xs = numpy.cumsum(numpy.random.default_rng(0).standard_normal((n, 3)), axis=0)
    e = HipEngine(n, dtype); e.set_wish_from_coords(xs); e.set_coords(xs + 0.5)
    e.iterate(50, 1 / (2 * n)); e.sync()
    k = 2000
    t0 = time.perf_counter(); e.iterate(k, 1 / (2 * n)); e.sync(); dt = time.perf_counter() - t0
    print("wpc=%s N=%d %s: %.2f us/iter" % (os.environ.get("BB_WAVES_PER_CU", "default"), n, dtype, dt / k * 1e6), flush=True)
    e.close()
'''
for wpc in (None, "1", "2"):
    env = dict(os.environ)
    if wpc: env["BB_WAVES_PER_CU"] = wpc
    subprocess.run([sys.executable, "-c", code], env=env)
env = dict(os.environ, BB_LIB=os.getcwd() + "/tools/variants/libabl_OLD.so")
print("OLD:"); subprocess.run([sys.executable, "-c", code], env=env)
