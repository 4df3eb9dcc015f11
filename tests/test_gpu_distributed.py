"""GPU: the multi-rank path with the real HIP kernels.

* 2 processes share the single GPU of the test box, each playing one rank with
  its own HipEngine; the all-reduce runs over gloo on host memory
  (solver.allreduce_exchange_host) -- transport only.  The sharded result must
  equal the single-rank GPU result within the config's fp tolerance (summation
  order differs) and the replicas must be bit-identical.
* 1 process, backend nccl (= RCCL), world_size 1: the device-resident path
  (torch-allocated exchange tensor, solver on torch's stream) that the 8-GPU
  run uses, rehearsed as far as one GPU allows.
"""
import os
import socket
import subprocess
import sys
import traceback

import numpy
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, k, dtype, q, comm="host"):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["BB_COMM"] = comm.split("-")[0]
        os.environ["BB_PEER_TIMEOUT_MS"] = "5000"
        if comm == "peer-one-launch":
            # ranks that share a GPU get the two-launch exchange by default (waiting workgroups
            # of one rank can keep another's sweep off the device); a small problem is safe
            os.environ["BB_PEER_FUSED"] = "1"
        comm = comm.split("-")[0]
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import blueberry_amd as bb
        from tests import _oracle
        xs = _oracle.random_walk(n)
        w = _oracle.wish_from_coords(xs)
        s = bb.StructureSolver(n_iter=k, dtype=dtype, kind="wish", device=0,
                               momentum=0.3 if comm == "peer" else 0.0)
        s.fit(w, init=_oracle.noisy_init(xs))
        band = bb.count_band_regions(_oracle.golden("band_count")["in_gappy_n5000"],
                                     distributed=True)
        q.put((rank, s.structure_, s.stress_, band))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, traceback.format_exc(), None, None))


@pytest.mark.parametrize("dtype,tol,world,comm,n", [
    ("float64", 1e-12, 2, "host", 1300), ("float32", 1e-5, 2, "host", 1300),
    # the peer exchange between real processes: arenas mapped through HIP IPC,
    # flags and partials written by the other process's kernels
    ("float64", 1e-12, 2, "peer", 1300), ("float32", 1e-5, 2, "peer", 1300),
    ("float32", 1e-5, 4, "peer", 1300),
    # the same with the whole exchange in one launch (what ranks with a GPU each get)
    ("float64", 1e-12, 2, "peer-one-launch", 1300), ("float32", 1e-5, 4, "peer-one-launch", 1300),
    # fp64 above 4096 bins: the 2 x 512 units (MFMA row reduction, parked row sums), an odd
    # number of ranks, heavy-ball momentum
    ("float64", 1e-12, 3, "peer", 7000)])
def test_ranks_share_one_gpu(dtype, tol, world, comm, n):
    import torch.multiprocessing as mp
    import blueberry_amd as bb
    from tests import _oracle
    k = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, k, dtype, q, comm))
             for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert not isinstance(r[1], str), r[1]
    xs = _oracle.random_walk(n)
    w = _oracle.wish_from_coords(xs)
    one = bb.StructureSolver(n_iter=k, dtype=dtype, kind="wish", distributed=False,
                             momentum=0.3 if comm.startswith("peer") else 0.0)
    one.fit(w, init=_oracle.noisy_init(xs))
    for rank, X, hist, band in results:
        assert numpy.abs(X - one.structure_).max() < tol * numpy.abs(one.structure_).max()
        assert numpy.abs(hist / one.stress_ - 1).max() < tol
        assert band == int(_oracle.golden("band_count")["out_gappy_n5000"])
    for r in results[1:]:
        assert numpy.array_equal(results[0][1], r[1])
        assert numpy.array_equal(results[0][2], r[2])


def _worker_genome(rank, world, port, n, k, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["BB_COMM"] = "peer"
        os.environ["BB_PEER_TIMEOUT_MS"] = "20000"
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from blueberry_amd.solver import HipEngine, run_iterations
        from tests import _oracle
        xs = _oracle.random_walk(n)
        eng = HipEngine(n, "float32", rank=rank, world=world, device=0)
        eng.set_wish_from_coords(xs)           # this rank's share, generated on the device
        eng.set_coords(_oracle.noisy_init(xs))
        run_iterations(eng, k, 1.0 / (2 * n), world)
        eng.sync()
        eng.peer_status()
        q.put((rank, eng.get_coords(), eng.stress_history(), eng._comm_state))
        dist.barrier()
        eng.close()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, traceback.format_exc(), None, None))


def _band_tiles(n, vw=512, width=2):
    """BASELINE config 5's own tile list -- what `bench.py --workload genome10kb` runs: one
    block per hg19 chromosome + a 1000-bin band (blueberry_amd.solver.tiles_from_blocks)."""
    from blueberry_amd.solver import tiles_from_blocks
    from blueberry_amd.utils import genome_boundaries
    return tiles_from_blocks(n, genome_boundaries(n), 1000, "float32")[0]


def _worker_band(rank, world, port, n, k, lr, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["BB_COMM"] = "peer"
        os.environ["BB_PEER_TIMEOUT_MS"] = "20000"
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from blueberry_amd.solver import HipEngine, run_iterations
        from tests import _oracle
        xs = _oracle.random_walk(n)
        eng = HipEngine(n, "float32", rank=rank, world=world, device=0, tiles=_band_tiles(n))
        eng.set_wish_from_coords(xs)
        eng.set_coords(_oracle.noisy_init(xs))
        run_iterations(eng, k, lr, world)
        eng.sync()
        eng.peer_status()
        q.put((rank, eng.get_coords(), eng.stress_history(), eng._comm_state))
        dist.barrier()
        eng.close()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, traceback.format_exc(), None, None))


def test_config5_shape_four_ranks_over_the_peer_exchange():
    """BASELINE config 5 at its real size and on the tile list bench.py times (N = 309,568
    bins, fp32, blocked-sparse: one block per chromosome + a band, 10.5 GB of units split
    four ways) on FOUR ranks -- four processes sharing the test box's GPU, partial gradients
    of 929 k coordinates summed through the peer exchange every iteration -- against the
    same iterations on one rank (1e-5), the ranks bit-identical among themselves."""
    import torch.multiprocessing as mp
    from blueberry_amd.solver import HipEngine
    from tests import _oracle
    from blueberry_amd.solver import max_degree
    n, k, world = 309568, 4, 4
    lr = 1.0 / (2 * max_degree(n, _band_tiles(n), "float32"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_band, args=(r, world, port, n, k, lr, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=400) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert not isinstance(r[1], str), r[1]
        assert r[3] == "peer"
    xs = _oracle.random_walk(n)
    one = HipEngine(n, "float32", tiles=_band_tiles(n))
    one.set_wish_from_coords(xs)
    one.set_coords(_oracle.noisy_init(xs))
    one.iterate(k, lr)
    X1, h1 = one.get_coords(), one.stress_history()
    one.close()
    for rank, X, hist, _ in results:
        assert numpy.abs(X - X1).max() < 1e-5 * numpy.abs(X1).max()
        assert numpy.abs(hist / h1 - 1).max() < 1e-5
    for r in results[1:]:
        assert numpy.array_equal(results[0][1], r[1]) and numpy.array_equal(results[0][2], r[2])
    assert (numpy.diff(h1) < 0).all()


def test_config4_size_two_ranks_over_the_peer_exchange():
    """BASELINE config 4 at its real size (N = 61,914 bins fp32, 7.67 GB of units) with the
    multi-rank path proper: two processes, each holding half of the units, summing their
    partial gradients through the in-kernel peer exchange (IPC arenas; the two share the one
    GPU of the test box) -- against the same iterations on one rank, within the config's
    1e-5, and bit-identical between the ranks."""
    import torch.multiprocessing as mp
    from blueberry_amd.solver import HipEngine
    from tests import _oracle
    n, k, world = 61914, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_genome, args=(r, world, port, n, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=400) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert not isinstance(r[1], str), r[1]
        assert r[3] == "peer"
    xs = _oracle.random_walk(n)
    one = HipEngine(n, "float32")
    one.set_wish_from_coords(xs)
    one.set_coords(_oracle.noisy_init(xs))
    one.iterate(k, 1.0 / (2 * n))
    X1, h1 = one.get_coords(), one.stress_history()
    one.close()
    for rank, X, hist, _ in results:
        assert numpy.abs(X - X1).max() < 1e-5 * numpy.abs(X1).max()
        assert numpy.abs(hist / h1 - 1).max() < 1e-5
    assert numpy.array_equal(results[0][1], results[1][1])
    assert numpy.array_equal(results[0][2], results[1][2])
    assert h1[-1] < h1[0]


def _peer_engines(world, n, dtype, wish, x0, mu=0.0):
    """`world` ranks inside this process, one HipEngine (own stream) each, their
    receive arenas connected directly (same-process shortcut of peer_connect)."""
    import ctypes
    from blueberry_amd import _lib
    from blueberry_amd.solver import HipEngine
    lib = _lib.load()
    engs = [HipEngine(n, dtype, rank=r, world=world) for r in range(world)]
    blobs = []
    for e in engs:
        buf = ctypes.create_string_buffer(_lib.BB_PEER_HANDLE_BYTES)
        _lib.check(lib.bb_solver_peer_export(e._h, buf), "export")
        blobs.append(buf.raw)
    for e in engs:
        _lib.check(lib.bb_solver_peer_connect(e._h, b"".join(blobs)), "connect")
        e.set_wish_dense(wish, "wish", 3.0)
        e.set_coords(x0)
        e.set_momentum(mu)
    return engs


@pytest.mark.parametrize("form", ["one launch", "two launches", "one launch, 8 slices"])
@pytest.mark.parametrize("dtype,tol,world", [("float32", 1e-5, 2), ("float64", 1e-12, 2),
                                             ("float32", 1e-5, 3)])
def test_peer_exchange_in_one_process(dtype, tol, world, form, monkeypatch):
    """Ranks on separate streams of one process: every iteration of every rank is
    enqueued up front, the kernels meet through the arenas' flags.  Both forms of the
    exchange (include/blueberry_hip.h): reduce + push + wait + sum + update in one launch,
    workgroup by workgroup, and the two launches with one flag per rank."""
    monkeypatch.setenv("BB_PEER_TIMEOUT_MS", "5000")
    if form.endswith("8 slices"):
        # the kernel's other instantiation (1024 threads, lists in 8 slices), which problems
        # of this size do not reach by themselves
        monkeypatch.setenv("BB_REDUCE_SLICES", "8")
        form = "one launch"
    monkeypatch.setenv("BB_PEER_FUSED", "1" if form == "one launch" else "0")
    from blueberry_amd.solver import HipEngine
    from tests import _oracle
    n, k = 1100, 7
    xs = _oracle.random_walk(n)
    w = _oracle.wish_from_coords(xs)
    x0 = _oracle.noisy_init(xs)
    lr = 1.0 / (2 * n)
    engs = _peer_engines(world, n, dtype, w, x0, mu=0.2)
    assert all(e.peer_form() == form for e in engs)
    for it in (3, k - 3):                       # two calls: the sequence carries over
        for e in engs:
            e.iterate_peer(it, lr)
    got = []
    for e in engs:
        assert e.peer_status() == 0
        got.append((e.get_coords(), e.stress_history()))
        e.close()
    one = HipEngine(n, dtype)
    one.set_wish_dense(w, "wish", 3.0)
    one.set_coords(x0)
    one.set_momentum(0.2)
    one.iterate(k, lr)
    X1, h1 = one.get_coords(), one.stress_history()
    one.close()
    for X, h in got:
        assert numpy.array_equal(X, got[0][0]) and numpy.array_equal(h, got[0][1])
        assert numpy.abs(X - X1).max() < tol * numpy.abs(X1).max()
        assert h.shape == h1.shape and numpy.abs(h / h1 - 1).max() < tol


def test_peer_exchange_forms_agree_bit_for_bit_over_a_long_run(monkeypatch):
    """600 iterations of three ranks through the one-launch exchange and through the two
    launches: both add the ranks' partials in rank order, so every rank of either run must
    hold the same bits -- a workgroup that ever read a slot before its data had landed (or a
    parity that was overwritten too early) would show here."""
    monkeypatch.setenv("BB_PEER_TIMEOUT_MS", "20000")
    from tests import _oracle
    n, k, world = 2600, 600, 3
    xs = _oracle.random_walk(n)
    w = _oracle.wish_from_coords(xs)
    x0 = _oracle.noisy_init(xs)
    runs = {}
    for form in ("one launch", "two launches"):
        monkeypatch.setenv("BB_PEER_FUSED", "1" if form == "one launch" else "0")
        engs = _peer_engines(world, n, "float32", w, x0, mu=0.3)
        assert engs[0].peer_form() == form
        # in turns and in small pieces: the ranks of ONE process share its host thread, and a
        # rank whose launch queue is full would block the thread that feeds the ranks it waits for
        for chunk in (1, 99) + (100,) * 5:
            for e in engs:
                e.iterate_peer(chunk, 1.0 / (2 * n))
        got = []
        for e in engs:
            assert e.peer_status() == 0
            got.append((e.get_coords(), e.stress_history()))
            e.close()
        for X, h in got[1:]:
            assert numpy.array_equal(X, got[0][0]) and numpy.array_equal(h, got[0][1])
        runs[form] = got[0]
    assert numpy.array_equal(runs["one launch"][0], runs["two launches"][0])
    assert numpy.array_equal(runs["one launch"][1], runs["two launches"][1])
    assert runs["one launch"][1].shape == (k,) and runs["one launch"][1][-1] < runs["one launch"][1][0]


@pytest.mark.parametrize("form", ["one launch", "two launches"])
def test_peer_exchange_times_out_cleanly(form, monkeypatch):
    """A rank whose peer never delivers must not hang: the wait is bounded, the
    update is skipped as a whole, the failure is sticky and reported."""
    monkeypatch.setenv("BB_PEER_TIMEOUT_MS", "200")
    monkeypatch.setenv("BB_PEER_FUSED", "1" if form == "one launch" else "0")
    from tests import _oracle
    n = 600
    xs = _oracle.random_walk(n)
    x0 = _oracle.noisy_init(xs)
    engs = _peer_engines(2, n, "float32", _oracle.wish_from_coords(xs), x0)
    engs[0].iterate_peer(3, 1.0 / (2 * n))      # rank 1 never runs
    with pytest.raises(RuntimeError, match="time limit"):
        engs[0].peer_status()
    assert numpy.array_equal(engs[0].get_coords(), x0.astype(numpy.float32).astype(numpy.float64))
    for e in engs:
        e.close()


@pytest.mark.parametrize("form", ["one launch", "two launches"])
def test_stalled_rank_fails_every_rank_and_leaves_x_whole(form, monkeypatch):
    """with every workgroup of the update polling for itself, a
    time-out could apply a step to part of X, and the timed-out rank kept feeding its
    peers partials of coordinates that no longer moved.  Two launches: one wave decides
    per launch; one launch: a workgroup applies its 128 elements only when all ranks'
    copies of them have arrived, and of a rank that never runs nothing arrives.  Either
    way a failed rank poisons its word on every peer.  Three ranks, rank 2
    stalled for good: rank 0 (200 ms limit) times out; rank 1 (60 s limit) must fail
    right behind it -- through the poison, not through its own clock -- and on both X
    is exactly the start, on every element."""
    import time
    monkeypatch.setenv("BB_PEER_TIMEOUT_MS", "60000")
    monkeypatch.setenv("BB_PEER_FUSED", "1" if form == "one launch" else "0")
    from tests import _oracle
    n = 5000                                       # 30 workgroups in the update launch
    xs = _oracle.random_walk(n)
    x0 = _oracle.noisy_init(xs)
    engs = _peer_engines(3, n, "float32", _oracle.wish_from_coords(xs), x0)
    engs[0].peer_set_timeout(200)
    t0 = time.perf_counter()
    for e in engs[:2]:
        e.iterate_peer(4, 1.0 / (2 * n))           # rank 2 never runs
    for e in engs[:2]:
        with pytest.raises(RuntimeError, match="time limit"):
            e.peer_status()
    assert time.perf_counter() - t0 < 20.0         # nobody sat out a 60 s limit
    want = x0.astype(numpy.float32).astype(numpy.float64)
    for e in engs[:2]:
        assert numpy.array_equal(e.get_coords(), want)
        assert e.stress_history().size == 4        # the slots exist; none was applied
    # the stalled rank, once it does run, finds the poison and stops as well
    engs[2].iterate_peer(1, 1.0 / (2 * n))
    with pytest.raises(RuntimeError, match="time limit"):
        engs[2].peer_status()
    assert numpy.array_equal(engs[2].get_coords(), want)
    for e in engs:
        e.close()


_NCCL_SCRIPT = r"""
import os, sys, numpy
sys.path.insert(0, %(root)r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="%(port)d")
import torch                       # BEFORE the first blueberry_amd compute call
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import blueberry_amd as bb
from blueberry_amd import _lib
from blueberry_amd.solver import HipEngine, run_iterations, select_exchange
from tests import _oracle
n, k = 900, 6
xs = _oracle.random_walk(n); w = _oracle.wish_from_coords(xs); x0 = _oracle.noisy_init(xs)
lr = 1.0 / (2 * n)
# the second pass: the same with a step per bin (bb_solver_set_bin_steps: the gradient is
# scaled where it leaves the reduce, so every transport carries it)
for factors in (None, numpy.random.default_rng(3).uniform(0.5, 1.5, n)):
  out = {}
  # world=2 forces grad/all-reduce/apply; BB_COMM picks the collective's owner
  for name, world, comm in (("fused", 1, None), ("exchange", 2, "rccl"), ("torch", 2, "torch"),
                            ("peer", 2, "peer"), ("auto", 2, "auto")):
    if comm: os.environ["BB_COMM"] = comm
    e = HipEngine(n, "float32")
    e.set_wish_dense(w, "wish", 3.0); e.set_coords(x0)
    if factors is not None: e.set_bin_steps(factors)
    if comm == "auto":
        # the trial: both transports from the same start, coordinates compared, the
        # faster one kept, the start restored
        state = select_exchange(e, lr, trial=True)
        assert state in ("peer", "rccl") and e._comm_trial["agree"], (state, e._comm_trial)
        assert numpy.array_equal(e.get_coords(), x0.astype(numpy.float32).astype(numpy.float64))
        assert e.stress_history().size == 0
        run_iterations(e, k, lr, world)
    else:
        run_iterations(e, k, lr, world)
        assert e._comm_state == comm, (e._comm_state, comm)
        if comm == "rccl":
            assert e.comm_world() == 1                  # what RCCL itself reports
            e.sync_timeout(10000)
    out[name] = (e.get_coords(), e.stress_history()); e.close()
  assert numpy.array_equal(out["torch"][0], out["exchange"][0])      # same kernels, same sums
  assert numpy.array_equal(out["peer"][0], out["exchange"][0])       # one rank: nothing to reorder
  assert numpy.array_equal(out["auto"][0], out["exchange"][0])
  assert numpy.array_equal(out["peer"][1], out["exchange"][1])
  assert len(_lib.hip_runtimes_loaded()) == 1, _lib.hip_runtimes_loaded()
  assert numpy.abs(out["fused"][0] - out["exchange"][0]).max() < 1e-5 * numpy.abs(out["fused"][0]).max()
  assert numpy.abs(out["fused"][1] / out["exchange"][1] - 1).max() < 1e-5
  if factors is None: plain = out["fused"][0]
assert numpy.abs(out["fused"][0] - plain).max() > 1e-3 * numpy.abs(plain).max()   # the factors did act
dist.destroy_process_group()
print("NCCL_PATH_OK")
"""


def test_rccl_single_rank_device_path():
    script = _NCCL_SCRIPT % {"root": ROOT, "port": _free_port()}
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True,
                       timeout=600, env=env)
    assert "NCCL_PATH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("form", ["one launch", "two launches"])
def test_peer_exchange_contributor_mask_changes_no_bit(form, monkeypatch):
    """A rank sends a block's partial only if its units touch the block (the table every rank
    derives from the tile list and the partition): the zeros it used to push are neither sent
    nor waited for nor added -- and not one bit of the result changes (BB_PEER_MASK=0 is the
    form that sends everything).  Three ranks on a dense map, where rank 0 touches the first
    tile columns only, and on a blocked-sparse one."""
    monkeypatch.setenv("BB_PEER_TIMEOUT_MS", "20000")
    monkeypatch.setenv("BB_PEER_FUSED", "1" if form == "one launch" else "0")
    from blueberry_amd.solver import tiles_from_blocks
    from tests import _oracle
    import ctypes
    from blueberry_amd import _lib
    from blueberry_amd.solver import HipEngine
    lib = _lib.load()
    world, k = 3, 5
    for n, tiles in ((2600, None), (3100, tiles_from_blocks(3100, [0, 1500, 2300, 3100], 100, "float32")[0])):
        xs = _oracle.random_walk(n)
        x0 = _oracle.noisy_init(xs)
        lr = 1.0 / (2 * n)
        out = {}
        for setting in ("1", "0"):
            monkeypatch.setenv("BB_PEER_MASK", setting)
            engs = [HipEngine(n, "float32", rank=r, world=world, tiles=tiles) for r in range(world)]
            blobs = []
            for e in engs:
                buf = ctypes.create_string_buffer(_lib.BB_PEER_HANDLE_BYTES)
                _lib.check(lib.bb_solver_peer_export(e._h, buf), "export")
                blobs.append(buf.raw)
            for e in engs:
                _lib.check(lib.bb_solver_peer_connect(e._h, b"".join(blobs)), "connect")
                e.set_wish_from_coords(xs)
                e.set_coords(x0)
                e.set_momentum(0.3)
            for _ in range(k):
                for e in engs:
                    e.iterate_peer(1, lr)
            for e in engs:
                assert e.peer_status() == 0
            out[setting] = [(e.get_coords(), e.stress_history()) for e in engs]
            for e in engs:
                e.close()
        for (Xa, ha), (Xb, hb) in zip(out["1"], out["0"]):
            assert numpy.array_equal(Xa, Xb) and numpy.array_equal(ha, hb)
        assert all(numpy.array_equal(X, out["1"][0][0]) for X, _ in out["1"][1:])


def test_eight_ranks_over_the_peer_exchange_in_one_process():
    """World size 8 -- what the scaling run uses -- rehearsed on one GPU: eight ranks of ONE
    process (more than the six processes a box allows on its GPU), arenas connected directly,
    random call sequences of tools/peer_sequence_fuzz.py (iterate_peer, the spectral start,
    per-bin steps, the two-call path) against the one-rank model; ranks bit-identical after
    every call.  Needs more hardware queues than the HIP runtime's default of four (streams
    that share a queue deadlock when one waits for the other), hence a process of its own
    with GPU_MAX_HW_QUEUES set; the one-launch form keeps to a size whose waiting workgroups
    of all eight ranks fit one chip (the tool says why)."""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="32", BB_FUZZ_WORLDS="8,5", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "peer_sequence_fuzz.py"), "12", "7"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "FAILURES: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "world=8" in r.stdout or "world=5" in r.stdout


def test_world_size_eight_at_full_sizes_on_one_gpu():
    """The scaling run's world size at BASELINE's sizes: eight ranks of one process, each with
    its 1/8 share of the units of the headline map (N = 50,000 dense fp32) and of config 5
    (N = 309,568 block-sparse), iterate over the peer exchange and end where ONE rank ends
    (1e-5; measured 6e-8), bit-identical among themselves (tools/world8_rehearsal.py; a
    process of its own for the hardware queues)."""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="32", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "world8_rehearsal.py")],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "world-8 rehearsal ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_peer_exchange_call_sequence_errors():
    """The peer entry points refuse to be used out of order or with handles that do
    not belong to this job (status codes + bb_last_error, no crash)."""
    import ctypes
    from blueberry_amd import _lib
    from blueberry_amd.solver import HipEngine
    from tests import _oracle
    lib = _lib.load()
    n = 300
    xs = _oracle.random_walk(n)
    a = HipEngine(n, "float32", rank=0, world=2)
    b = HipEngine(n, "float32", rank=1, world=2)
    other = HipEngine(n + 600, "float32", rank=1, world=2)          # another problem size
    a.set_wish_from_coords(xs)
    a.set_coords(xs)
    buf = lambda: ctypes.create_string_buffer(_lib.BB_PEER_HANDLE_BYTES)
    ha, hb, ho = buf(), buf(), buf()
    st = ctypes.c_int()
    assert lib.bb_solver_iterate_peer(a._h, 1, 0.1) == _lib.BB_ERR_STATE      # not connected
    assert lib.bb_solver_peer_status(a._h, st) == _lib.BB_ERR_STATE
    assert lib.bb_solver_peer_connect(a._h, ha) == _lib.BB_ERR_STATE          # export first
    assert lib.bb_solver_peer_export(a._h, ha) == _lib.BB_OK
    assert lib.bb_solver_peer_export(a._h, ha) == _lib.BB_ERR_STATE           # once only
    assert lib.bb_solver_peer_export(b._h, hb) == _lib.BB_OK
    assert lib.bb_solver_peer_export(other._h, ho) == _lib.BB_OK
    assert lib.bb_solver_peer_connect(a._h, hb.raw + ha.raw) == _lib.BB_ERR_INVALID   # rank order
    assert b"does not match" in lib.bb_last_error()
    assert lib.bb_solver_peer_connect(a._h, ha.raw + ho.raw) == _lib.BB_ERR_INVALID   # other n_bins
    assert lib.bb_solver_peer_connect(a._h, ha.raw + hb.raw) == _lib.BB_OK
    assert lib.bb_solver_peer_connect(a._h, ha.raw + hb.raw) == _lib.BB_ERR_STATE     # once only
    assert lib.bb_solver_peer_status(a._h, st) == _lib.BB_OK and st.value == 0
    assert lib.bb_solver_peer_export(None, ha) == _lib.BB_ERR_INVALID
    for e in (a, b, other):
        e.close()


class _CountingLib(object):
    """Wraps the ctypes library of one engine and counts the calls that move an (N, 3) array
    between host and device."""
    moving = ("bb_solver_get_coords", "bb_solver_set_coords", "bb_solver_matvec_sq",
              "bb_solver_read_exchange", "bb_solver_write_exchange")

    def __init__(self, lib):
        self._lib, self.counts = lib, {}

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in self.moving:
            def counted(*a, **k):
                self.counts[name] = self.counts.get(name, 0) + 1
                return fn(*a, **k)
            return counted
        return fn


@pytest.mark.parametrize("form", ["one launch", "two launches"])
@pytest.mark.parametrize("dtype,tol,world", [("float64", 1e-12, 2), ("float32", 1e-5, 3)])
def test_block_steps_on_several_ranks(dtype, tol, world, form, monkeypatch):
    """bb_solver_set_block_steps on several ranks: the gradient is scaled where it leaves each
    rank's reduce, so the peer exchange (both forms) and the two-call path with a host sum
    step exactly as one rank does; replicas bit-identical."""
    monkeypatch.setenv("BB_PEER_TIMEOUT_MS", "20000")
    monkeypatch.setenv("BB_PEER_FUSED", "1" if form == "one launch" else "0")
    from blueberry_amd.solver import HipEngine
    from tests import _oracle
    n, k = 2600, 5
    lr = 1.0 / (2 * n)
    xs = _oracle.random_walk(n)
    w = _oracle.wish_from_coords(xs)
    x0 = _oracle.noisy_init(xs)
    one = HipEngine(n, dtype)
    scale = numpy.random.default_rng(2).uniform(0.4, 1.6, one.layout()["n_blocks"])
    one.set_wish_dense(w, "wish", 3.0)
    one.set_block_steps(scale)
    one.set_momentum(0.3)
    one.set_coords(x0)
    one.iterate(k, lr)
    X1, h1 = one.get_coords(), one.stress_history()
    one.close()
    engs = _peer_engines(world, n, dtype, w, x0, mu=0.3)
    assert all(e.peer_form() == form for e in engs)
    for e in engs:
        e.set_block_steps(scale)
    for _ in range(k):                                # ranks of one process: step by step
        for e in engs:
            e.iterate_peer(1, lr)
    Xs = [e.get_coords() for e in engs]
    hs = [e.stress_history() for e in engs]
    for e in engs:
        assert e.peer_status() == 0
    for X, h in zip(Xs, hs):
        assert numpy.abs(X - X1).max() < tol * numpy.abs(X1).max()
        assert numpy.abs(h / h1 - 1).max() < tol
        assert numpy.array_equal(X, Xs[0]) and numpy.array_equal(h, hs[0])
    # the two-call path: grad, sum of the exchange buffers on the host, apply
    for e in engs:
        e.set_coords(x0)
    for _ in range(k):
        for e in engs:
            e.grad()
        total = sum(e.read_exchange() for e in engs)
        for e in engs:
            e.write_exchange(total)
            e.apply(lr)
    for e in engs:
        assert numpy.abs(e.get_coords() - X1).max() < tol * numpy.abs(X1).max()
        e.close()


@pytest.mark.parametrize("form", ["one launch", "two launches"])
@pytest.mark.parametrize("dtype,tol,world", [("float64", 1e-12, 2), ("float32", 1e-5, 3)])
def test_multi_rank_spectral_start_stays_on_the_device(dtype, tol, world, form, monkeypatch):
    """init='spectral' on several ranks.  `world` ranks of this process, their
    arenas connected, each driven by a thread of its own (the start synchronises at its end;
    the ranks' kernels meet in the exchange): the per-rank products of the block power
    iteration are summed through the peer exchange ON THE DEVICE -- no (N, 3) array crosses
    PCIe inside the loop -- and every rank ends with the start one rank computes alone
    (same Ritz sign rule), bit-identical among the ranks."""
    import threading
    monkeypatch.setenv("BB_PEER_TIMEOUT_MS", "20000")
    monkeypatch.setenv("BB_PEER_FUSED", "1" if form == "one launch" else "0")
    from blueberry_amd.solver import HipEngine
    from tests import _oracle
    n = 2600                                      # above the row-owner switch for world = 1 too
    xs = _oracle.random_walk(n)
    w = _oracle.wish_from_coords(xs)
    v0 = numpy.random.default_rng(0).standard_normal((n, 3))
    one = HipEngine(n, dtype)
    one.set_wish_dense(w, "wish", 3.0)
    one.spectral_init_device(40, v0)
    X1 = one.get_coords()
    # (the stopping rule: every rank must leave the loop at the same product -- they hold
    # the same V and Z bit for bit -- and at the product one rank alone leaves at)
    hole = numpy.triu(numpy.random.default_rng(5).random((n, n)) < 0.1, 1)
    wm = w.copy()
    wm[hole | hole.T] = 0.0
    one.set_wish_dense(wm, "wish", 3.0)
    done1 = one.spectral_init_device(80, v0, tol=1e-3)[0]
    X1m = one.get_coords()
    one.close()
    engs = _peer_engines(world, n, dtype, w, v0)
    assert all(e.peer_form() == form for e in engs)
    for e in engs:
        e._lib = _CountingLib(e._lib)
    errs = []

    def run(e):
        try:
            e.spectral_init_device(40, v0)
        except Exception as exc:                  # noqa: BLE001
            errs.append(exc)

    threads = [threading.Thread(target=run, args=(e,)) for e in engs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errs, errs
    for e in engs:
        assert e.peer_status() == 0
        assert e._lib.counts == {}, e._lib.counts   # nothing of size N moved during the start
    Xs = [e.get_coords() for e in engs]
    for X in Xs:
        assert numpy.abs(X - X1).max() < tol * numpy.abs(X1).max()
    for X in Xs[1:]:
        assert numpy.array_equal(X, Xs[0])
    # the start is a start: the solver goes on from it over the same exchange
    lr = 1.0 / (2 * n)
    for e in engs:
        e.iterate_peer(3, lr)
    hs = [e.stress_history() for e in engs]
    for e in engs:
        assert e.peer_status() == 0
        e.close()
    d = _oracle.wish_from_coords(Xs[0])
    assert numpy.abs(d - w).max() < (1e-6 if dtype == "float64" else 1e-3) * w.max()
    assert all(numpy.array_equal(h, hs[0]) for h in hs) and hs[0].shape == (3,)
    engs = _peer_engines(world, n, dtype, wm, v0)
    dones = [None] * world

    def run_tol(r):
        try:
            dones[r] = engs[r].spectral_init_device(80, v0, tol=1e-3)[0]
        except Exception as exc:                  # noqa: BLE001
            errs.append(exc)

    threads = [threading.Thread(target=run_tol, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errs, errs
    assert 1 < done1 < 80 and all(abs(d - done1) <= (0 if dtype == "float64" else 1) for d in dones)
    assert len(set(dones)) == 1
    Xm = [e.get_coords() for e in engs]
    for e in engs:
        assert e.peer_status() == 0
        e.close()
    assert all(numpy.array_equal(X, Xm[0]) for X in Xm[1:])
    if dones[0] == done1:
        assert numpy.abs(Xm[0] - X1m).max() < 1e3 * tol * numpy.abs(X1m).max()


def _degree_map(n):
    """Two dense blocks (5 : 1) joined by a band: the bins' degrees differ by a factor of 5."""
    from tests import _oracle
    xs = _oracle.random_walk(n)
    w = _oracle.wish_from_coords(xs)
    i, j = numpy.indices((n, n))
    cut = (5 * n) // 6
    keep = ((i < cut) & (j < cut)) | ((i >= cut) & (j >= cut)) | (numpy.abs(i - j) <= 40)
    return numpy.where(keep, w, 0.0), _oracle.noisy_init(xs)


def _worker_degree_fit(rank, world, port, n, dtype, comm, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["BB_COMM"] = comm
        os.environ["BB_PEER_TIMEOUT_MS"] = "20000"
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import blueberry_amd as bb
        w, x0 = _degree_map(n)
        s = bb.StructureSolver(n_iter=6, dtype=dtype, kind="wish", device=0, degree_steps=True,
                               momentum=0.3).fit(w, init=x0)
        q.put((rank, s.structure_, s.stress_, s.exchange_, s.lr_))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, traceback.format_exc(), None, None, None))


@pytest.mark.parametrize("dtype,tol,world,comm", [("float64", 1e-11, 2, "peer"),
                                                   ("float32", 1e-5, 3, "peer"),
                                                   ("float64", 1e-11, 2, "host")])
def test_fit_with_degree_steps_across_processes(dtype, tol, world, comm):
    """fit(degree_steps=True) on real processes sharing the test GPU: every rank counts the
    degrees of its own units on the device, the counts are summed over the ranks, the same
    per-bin factors go to every rank's solver, and the scaled gradients travel through the
    peer exchange (arenas over HIP IPC) or the host-staged sum.  Equal to one process."""
    import torch.multiprocessing as mp
    import blueberry_amd as bb
    n = 2300
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_degree_fit, args=(r, world, port, n, dtype, comm, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert not isinstance(r[1], str), r[1]
        assert r[3] == comm
    w, x0 = _degree_map(n)
    one = bb.StructureSolver(n_iter=6, dtype=dtype, kind="wish", degree_steps=True, momentum=0.3,
                             distributed=False).fit(w, init=x0)
    deg = (w > 0).sum(axis=0)
    assert one.lr_ == 1.0 / (2 * (deg.max() + 1)) and deg.max() > 4 * deg.min()
    for rank, X, h, _, lr in results:
        assert lr == one.lr_
        assert numpy.abs(h / one.stress_ - 1).max() < tol
        assert numpy.abs(X - one.structure_).max() < tol * numpy.abs(one.structure_).max()
    for r in results[1:]:
        assert numpy.array_equal(r[1], results[0][1])


def _worker_spectral_fit(rank, world, port, n, dtype, comm, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["BB_COMM"] = comm
        os.environ["BB_PEER_TIMEOUT_MS"] = "20000"
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import blueberry_amd as bb
        from blueberry_amd import solver
        from tests import _oracle
        calls = {"host_form": 0}
        host_form = solver.spectral_init

        def counted(*a, **k):
            calls["host_form"] += 1
            return host_form(*a, **k)

        solver.spectral_init = counted
        xs = _oracle.random_walk(n)
        w = _oracle.wish_from_coords(xs)
        s = bb.StructureSolver(n_iter=3, dtype=dtype, kind="wish", device=0, init="spectral",
                               seed=0).fit(w)
        q.put((rank, s.structure_, s.stress_, s.exchange_, calls["host_form"]))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, traceback.format_exc(), None, None, None))


@pytest.mark.parametrize("dtype,tol,world,comm", [("float64", 1e-9, 2, "peer"),
                                                   ("float32", 1e-3, 3, "peer"),
                                                   ("float64", 1e-9, 2, "host")])
def test_fit_with_spectral_start_across_processes(dtype, tol, world, comm):
    """The whole Python path on real processes (sharing the test box's GPU, arenas mapped
    through HIP IPC): `fit(init='spectral')` chooses its exchange first, then computes the
    classical-MDS start ON THE DEVICE on every rank when the exchange sums there (peer), the
    per-rank products going through it -- and through the host-driven loop when it does not
    (gloo, host-staged).  Every rank ends with the start one process computes alone and the
    same three iterations from it."""
    import torch.multiprocessing as mp
    import blueberry_amd as bb
    from tests import _oracle
    n = 2300
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_spectral_fit, args=(r, world, port, n, dtype, comm, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert not isinstance(r[1], str), r[1]
        assert r[3] == comm
        assert r[4] == (0 if comm == "peer" else 1)        # device form vs host-driven form
    w = _oracle.wish_from_coords(_oracle.random_walk(n))
    one = bb.StructureSolver(n_iter=3, dtype=dtype, kind="wish", init="spectral", seed=0,
                             distributed=False).fit(w)
    scale = numpy.abs(one.structure_).max()
    for rank, X, hist, _, _ in results:
        assert numpy.abs(X - one.structure_).max() < tol * scale
        # exact recovery on a complete map: the stress is at rounding level on both sides
        assert hist[0] < 1e-6 * (w ** 2).sum() and one.stress_[0] < 1e-6 * (w ** 2).sum()
    for r in results[1:]:
        assert numpy.array_equal(results[0][1], r[1])


def _worker_fit_many(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import blueberry_amd as bb
        from tests import _oracle
        sizes = [900, 300, 1500, 640, 1100]
        mats = [_oracle.wish_from_coords(_oracle.random_walk(n, seed=50 + m)) for m, n in enumerate(sizes)]
        s = bb.StructureSolver(n_iter=5, dtype="float32", kind="wish", device=0, seed=1).fit_many(mats)
        q.put((rank, s.structures_, s.stresses_, s.ranks_of_maps_))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, traceback.format_exc(), None, None))


def test_fit_many_across_processes(oracle):
    """Several ranks: the maps are dealt to the ranks, every rank solves its own in one solver on
    the device, all ranks get all results; each map against the oracle's solve of it (fp32)."""
    import torch.multiprocessing as mp
    from tests import _oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fit_many, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert not isinstance(r[1], str), r[1]
    assert results[0][3] == results[1][3] and set(results[0][3]) == {0, 1}
    sizes = [900, 300, 1500, 640, 1100]
    for m, n in enumerate(sizes):
        w = _oracle.wish_from_coords(_oracle.random_walk(n, seed=50 + m))
        x0 = numpy.random.default_rng(1).standard_normal((n, 3))
        X, h = oracle.solve(w, x0, 5, 1.0 / (2 * n), f64=False)
        for r in results:
            assert numpy.abs(r[1][m] - X).max() < 1e-5 * numpy.abs(X).max()
            assert numpy.abs(r[2][m] / h - 1).max() < 1e-5
        assert numpy.array_equal(results[0][1][m], results[1][1][m])
