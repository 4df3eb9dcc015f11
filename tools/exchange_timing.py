"""Device-side step time of the four ways an iteration can be closed, on ONE rank
(RCCL world size 1): fused reduce+update, library RCCL all-reduce, peer exchange,
torch.distributed all-reduce.  What differs between them is fixed cost only
(launches, fences, the collective's own latency), which is what limits strong
scaling.  Usage: python tools/exchange_timing.py [n_bins ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29655")
import numpy
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
fd = os.dup(1)
os.dup2(2, 1)                      # RCCL prints a banner on fd 1
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.all_reduce(torch.zeros(1, device="cuda"))
torch.cuda.synchronize()
from blueberry_amd.solver import HipEngine, run_iterations

sizes = [int(v) for v in sys.argv[1:]] or [50000, 17700]
rows = []
for n in sizes:
    rng = numpy.random.default_rng(0)
    xs = numpy.cumsum(rng.standard_normal((n, 3)), axis=0)
    x0 = xs + 0.5 * rng.standard_normal(xs.shape)
    for name, world, comm in (("fused", 1, "auto"), ("rccl", 2, "rccl"), ("peer", 2, "peer"),
                              ("peer2", 2, "peer"), ("torch", 2, "torch")):
        os.environ["BB_COMM"] = comm
        # peer: reduce + push + wait + sum + update in one launch; peer2: the two-launch form
        os.environ["BB_PEER_FUSED"] = "0" if name == "peer2" else "1"
        e = HipEngine(n, "float32")
        e.set_wish_from_coords(xs)
        e.set_coords(x0)
        run_iterations(e, 10, 1.0 / (2 * n), world)
        e.sync()
        e.set_timing(True)
        run_iterations(e, 50, 1.0 / (2 * n), world)
        e.sync()
        t = e.timing()
        rows.append("n=%6d %-6s step %.4f ms = kernel %.4f + reduce %.4f + rest %.4f" % (
            n, name, t["step_ms"], t["grad_ms"], t["reduce_ms"],
            t["step_ms"] - t["grad_ms"] - t["reduce_ms"]))
        # the same without any timing events: host clock around 1000 enqueued iterations
        e.set_timing(False)
        import time
        run_iterations(e, 50, 1.0 / (2 * n), world)
        e.sync()
        t0 = time.perf_counter()
        run_iterations(e, 1000, 1.0 / (2 * n), world)
        e.sync()
        rows[-1] += "   | no events: %.4f ms per step" % ((time.perf_counter() - t0))
        e.close()
os.dup2(fd, 1)
print("\n".join(rows))
dist.destroy_process_group()
