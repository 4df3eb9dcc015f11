"""CPU, world_size 2 over gloo: the N>1 path of the solver and of the band
count, with one rank's compute played by the oracle (tests/_engines.py) and
everything else -- unit partition, exchange-buffer layout, the per-iteration
all-reduce, the identical update on every rank -- the product's own code."""
import os
import socket
import sys
import traceback

import numpy
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, k, dtype, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import blueberry_amd as bb
        from blueberry_amd.band import allreduce_count, band_row_share
        from tests import _oracle
        from tests._engines import OracleEngine

        xs = _oracle.random_walk(n)
        w = _oracle.wish_from_coords(xs)
        x0 = _oracle.noisy_init(xs)
        s = bb.StructureSolver(n_iter=k, dtype=dtype, kind="wish", engine=OracleEngine)
        s.fit(w, init=x0)                      # distributed=None: picks up the gloo job
        # spectral start: the per-rank matvec shares are summed over the ranks
        sp = bb.StructureSolver(n_iter=1, dtype=dtype, kind="wish", engine=OracleEngine,
                                init="spectral").fit(w)
        assert sp.stress_[0] < 1e-6 * s.stress_[0], (sp.stress_[0], s.stress_[0])

        r = _oracle.golden("band_count")["in_gappy_n1000"]
        a, b = band_row_share(r.shape[0], rank, world)
        total = allreduce_count(_oracle.load().count_band_regions_rows(r, a, b))
        q.put((rank, s.structure_, s.stress_, total))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, traceback.format_exc(), None, None))


@pytest.mark.parametrize("world,dtype", [(2, "float32"), (2, "float64"), (3, "float32")])
def test_sharded_solve_equals_single_process_oracle(oracle, world, dtype):
    import torch.multiprocessing as mp
    from tests import _oracle
    n, k = 700, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, k, dtype, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert not isinstance(r[1], str), r[1]

    xs = _oracle.random_walk(n)
    w = _oracle.wish_from_coords(xs)
    X_ref, h_ref = oracle.solve(w, _oracle.noisy_init(xs), k, 1.0 / (2 * n))
    band_ref = int(_oracle.golden("band_count")["out_gappy_n1000"])
    results.sort(key=lambda t: t[0])
    for rank, X, hist, band_total in results:
        assert numpy.abs(X - X_ref).max() < 1e-11 * numpy.abs(X_ref).max()
        assert numpy.abs(hist / h_ref - 1).max() < 1e-11
        assert band_total == band_ref                      # integer all-reduce: exact
    # every rank applied the identical update: replicas are bit-identical
    for rank, X, hist, _ in results[1:]:
        assert numpy.array_equal(X, results[0][1]) and numpy.array_equal(hist, results[0][2])


# ---- degree_steps on several ranks: the degrees are summed over the ranks ----------------
def _degree_worker(rank, world, port, n, k, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import blueberry_amd as bb
        from tests import _oracle
        from tests._engines import OracleEngine
        w, x0 = _holey_map(n)
        s = bb.StructureSolver(n_iter=k, dtype="float64", kind="wish", engine=OracleEngine,
                               degree_steps=True).fit(w, init=x0)
        q.put((rank, s.structure_, s.stress_, s.lr_))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, traceback.format_exc(), None, None))


def _holey_map(n):
    """Two dense blocks of unequal size (5 : 1) joined by a narrow band."""
    from tests import _oracle
    xs = _oracle.random_walk(n)
    w = _oracle.wish_from_coords(xs)
    i, j = numpy.indices((n, n))
    cut = (5 * n) // 6
    keep = ((i < cut) & (j < cut)) | ((i >= cut) & (j >= cut)) | (numpy.abs(i - j) <= 30)
    return numpy.where(keep, w, 0.0), _oracle.noisy_init(xs)


def test_degree_steps_on_two_ranks_equal_one_process(oracle):
    """StructureSolver(degree_steps=True) on 2 gloo ranks: every rank counts the degrees of
    its own units, the counts are summed over the ranks, every rank sets the same factors
    and step; the result is the one-process run and the oracle loop with those factors."""
    import torch.multiprocessing as mp
    import blueberry_amd as bb
    from blueberry_amd.solver import degree_step_factors
    from tests._engines import OracleEngine
    n, k, world = 600, 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_degree_worker, args=(r, world, port, n, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert not isinstance(r[1], str), r[1]
    w, x0 = _holey_map(n)
    deg = (w > 0).sum(axis=0)
    lr, scale = degree_step_factors(deg)
    assert scale.max() > 3.0
    X, hist = x0.copy(), []
    for _ in range(k):
        s_, g = oracle.stress_grad(w, X)
        X = X - lr * scale[:, None] * g
        hist.append(s_)
    one = bb.StructureSolver(n_iter=k, dtype="float64", kind="wish", engine=OracleEngine,
                             distributed=False, degree_steps=True).fit(w, init=x0)
    plain = bb.StructureSolver(n_iter=k, dtype="float64", kind="wish", engine=OracleEngine,
                               distributed=False).fit(w, init=x0)
    assert one.lr_ == lr and numpy.abs(one.stress_ / numpy.array(hist) - 1).max() < 1e-12
    assert one.stress_[-1] < plain.stress_[-1]
    for rank, Xr, hr, lr_r in results:
        assert lr_r == lr
        assert numpy.abs(hr / numpy.array(hist) - 1).max() < 1e-11
        assert numpy.abs(Xr - X).max() < 1e-11 * numpy.abs(X).max()
    assert numpy.array_equal(results[0][1], results[1][1])


# ---- init='spectral' on several ranks: which form runs, and that the ranks agree --------
def _spectral_worker(rank, world, port, n, fail_rank, q):
    """fit(init='spectral') on `world` gloo ranks with an engine that HAS a device-resident
    start and a transport that sums on the device (played: _comm_state = 'peer').  The
    device form must be tried after the exchange is chosen, and if it is refused anywhere
    (RankDeficient on `fail_rank`) EVERY rank must take the host-driven start."""
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import blueberry_amd as bb
        from blueberry_amd import solver
        from tests import _oracle
        from tests._engines import OracleEngine

        log = []
        host_form = solver.spectral_init

        def counted_host_form(*a, **k):
            log.append("host")
            return host_form(*a, **k)

        class Eng(OracleEngine):
            def spectral_init_device(self, n_iter, v0, tol=0.0):
                assert self._comm_state == "peer", "the exchange must be chosen first"
                log.append("device")
                x = host_form(self, self.n_bins, self.world, n_iter=n_iter, seed=0, tol=tol)   # collective
                if rank == fail_rank:
                    raise solver.RankDeficient("the iterate lost rank (scripted)")
                self.set_coords(x)

            def peer_status(self):
                return 0

            def iterate_peer(self, k, lr):
                for _ in range(k):
                    self.grad()
                    solver.allreduce_exchange_host(self)
                    self.apply(lr)

        def fake_select(eng, lr, trial=False):
            eng._comm_state = "peer"
            return "peer"

        solver.select_exchange = fake_select
        solver.spectral_init = counted_host_form
        xs = _oracle.random_walk(n)
        w = _oracle.wish_from_coords(xs)
        s = bb.StructureSolver(n_iter=2, dtype="float64", kind="wish", engine=Eng,
                               init="spectral").fit(w)
        q.put((rank, log, s.structure_, s.stress_))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, traceback.format_exc(), None, None))


@pytest.mark.parametrize("fail_rank", [None, 1])
def test_multi_rank_spectral_start_control_flow(fail_rank):
    import torch.multiprocessing as mp
    from tests import _oracle
    n, world = 300, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_spectral_worker, args=(r, world, port, n, fail_rank, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=240) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert not isinstance(r[1], str), r[1]
        # the device form is tried once everywhere; refused anywhere -> host form everywhere
        assert r[1] == (["device"] if fail_rank is None else ["device", "host"])
    w = _oracle.wish_from_coords(_oracle.random_walk(n))
    for rank, _, X, hist in results:
        assert numpy.abs(_oracle.wish_from_coords(X) - w).max() < 1e-6 * w.max()
    assert numpy.array_equal(results[0][2], results[1][2])         # replicas identical


# ---- fit_many on several ranks: every rank solves maps of its own ----------------------------
def _many_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import blueberry_amd as bb
        from tests import _oracle
        from tests._engines import OracleEngine
        sizes = [300, 90, 640, 200, 515]
        mats = [_oracle.wish_from_coords(_oracle.random_walk(n, seed=q_)) for q_, n in enumerate(sizes)]
        s = bb.StructureSolver(n_iter=3, dtype="float64", kind="wish", seed=2,
                               engine=OracleEngine).fit_many(mats)      # picks up the gloo job
        # the options travel to the per-rank solvers: a step per bin from each map's degrees
        holes = []
        for q_, m in enumerate(mats[:3]):
            keep = numpy.triu(numpy.random.default_rng(q_).random(m.shape) < 0.3, 1)
            keep |= numpy.triu(numpy.ones(m.shape, dtype=bool), 1) & ~numpy.triu(numpy.ones(m.shape, dtype=bool), 2)
            holes.append(numpy.where(keep | keep.T, m, 0.0))
        d = bb.StructureSolver(n_iter=3, dtype="float64", kind="wish", seed=2, engine=OracleEngine,
                               degree_steps=True).fit_many(holes)
        q.put((rank, s.structures_, s.stresses_, s.ranks_of_maps_, d.lrs_, d.stresses_))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, traceback.format_exc(), None, None, None, None))


def test_fit_many_deals_the_maps_to_the_ranks(oracle):
    """Several GPUs: the maps are independent, so every rank solves maps of its own (dealt by
    size) and all ranks get all results -- equal to the oracle's solve of each map."""
    import torch.multiprocessing as mp
    from tests import _oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_many_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=240) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert not isinstance(r[1], str), r[1]
    sizes = [300, 90, 640, 200, 515]
    owners = results[0][3]
    assert owners == results[1][3] and set(owners) == {0, 1}
    assert owners[2] != owners[4]                    # the two largest maps go to different ranks
    for m, n in enumerate(sizes):
        w = _oracle.wish_from_coords(_oracle.random_walk(n, seed=m))
        x0 = numpy.random.default_rng(2).standard_normal((n, 3))
        X, h = oracle.solve(w, x0, 3, 1.0 / (2 * n))
        for r in results:
            assert numpy.array_equal(r[1][m], X) and numpy.array_equal(r[2][m], h)
    # degree_steps reached the ranks' own solvers: every map's step is its best-connected bin's
    import blueberry_amd as bb
    from tests._engines import OracleEngine
    for m in range(3):
        w = _oracle.wish_from_coords(_oracle.random_walk(sizes[m], seed=m))
        keep = numpy.triu(numpy.random.default_rng(m).random(w.shape) < 0.3, 1)
        keep |= numpy.triu(numpy.ones(w.shape, dtype=bool), 1) & ~numpy.triu(numpy.ones(w.shape, dtype=bool), 2)
        hm = numpy.where(keep | keep.T, w, 0.0)
        one = bb.StructureSolver(n_iter=3, dtype="float64", kind="wish", seed=2, engine=OracleEngine,
                                 distributed=False, degree_steps=True).fit(hm)
        for r in results:
            assert r[4][m] == one.lr_ == 1.0 / (2 * ((hm > 0).sum(axis=0).max() + 1))
            assert numpy.abs(r[5][m] / one.stress_ - 1).max() < 1e-12


# ---- select_exchange: the decision logic, with a scripted engine -----------------
class _ScriptedEngine(object):
    """Stands in for HipEngine: records what select_exchange does to it and lets a
    test script how each transport behaves (no GPU, no RCCL)."""

    def __init__(self, peer_ok=True, rccl_ok=True, peer_shift=0.0, peer_raises=False,
                 peer_sleep=0.0, rccl_sleep=0.0):
        self.world, self.rank = 1, 0
        self.x = numpy.arange(12.0).reshape(4, 3)
        self.calls = []
        self._comm_state = None
        self._comm_trial = None
        self.peer_ok, self.rccl_ok = peer_ok, rccl_ok
        self.peer_shift, self.peer_raises = peer_shift, peer_raises
        self.peer_sleep, self.rccl_sleep = peer_sleep, rccl_sleep
        self._peer_error = "scripted"

    def peer_setup(self):
        return self.peer_ok

    def comm_setup(self):
        return self.rccl_ok

    def get_coords(self):
        return self.x.copy()

    def set_coords(self, x):
        self.x = numpy.array(x, dtype=float)
        self.calls.append("set_coords")

    def sync(self):
        pass

    def peer_status(self):
        if self.peer_raises:
            raise RuntimeError("peer exchange: time limit")
        return 0

    def iterate_dist(self, k, lr):
        import time
        time.sleep(self.rccl_sleep * k)
        self.x = self.x - lr * k
        self.calls.append("rccl")

    def iterate_peer(self, k, lr):
        import time
        time.sleep(self.peer_sleep * k)
        self.x = self.x - lr * k * (1.0 + self.peer_shift)
        self.calls.append("peer")


class _ScriptedTorchEngine(_ScriptedEngine):
    """The same with the torch.distributed path played too (exchange_tensor / grad / apply):
    what the trial holds the peer exchange against when the library's communicator cannot
    be made."""

    def exchange_tensor(self):
        import torch
        return torch.zeros(3)

    def grad(self):
        import time
        time.sleep(self.rccl_sleep)

    def apply(self, lr):
        self.x = self.x - lr
        self.calls.append("torch")


@pytest.fixture
def one_rank_group(monkeypatch):
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(port))
    monkeypatch.delenv("BB_COMM", raising=False)
    dist.init_process_group("gloo", rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("kwargs,expect", [
    (dict(peer_sleep=0.0, rccl_sleep=0.002), "peer"),          # agrees and is faster
    (dict(peer_sleep=0.002, rccl_sleep=0.0), "rccl"),          # agrees but is slower
    (dict(peer_shift=0.1, rccl_sleep=0.002), "rccl"),          # faster but WRONG
    (dict(peer_raises=True, rccl_sleep=0.002), "rccl"),        # a wait timed out
    (dict(peer_ok=False), "rccl"),                             # arenas could not be mapped
    (dict(peer_ok=False, rccl_ok=False), "torch"),             # neither
])
def test_select_exchange_auto_is_a_measured_validated_choice(one_rank_group, monkeypatch, kwargs,
                                                             expect):
    from blueberry_amd import solver
    monkeypatch.setattr(one_rank_group, "get_backend", lambda *a, **k: "nccl")
    e = _ScriptedEngine(**kwargs)
    x0 = e.get_coords()
    assert solver.select_exchange(e, 0.5, trial=True) == expect
    assert e._comm_state == expect
    assert numpy.array_equal(e.get_coords(), x0)               # the start is restored
    if kwargs.get("peer_ok", True) and kwargs.get("rccl_ok", True):
        agree = not kwargs.get("peer_shift") and not kwargs.get("peer_raises")
        assert e._comm_trial["agree"] == agree
    assert solver.select_exchange(e, 0.5) == expect            # decided once


@pytest.mark.parametrize("kwargs,expect", [
    (dict(rccl_ok=False, peer_sleep=0.0, rccl_sleep=0.002), "peer"),     # agrees with torch, faster
    (dict(rccl_ok=False, peer_shift=0.1, rccl_sleep=0.002), "torch"),    # faster but WRONG
    (dict(rccl_ok=False, peer_sleep=0.002, rccl_sleep=0.0), "torch"),    # agrees but slower
])
def test_trial_against_torch_when_the_library_communicator_cannot_be_made(one_rank_group,
                                                                           monkeypatch, kwargs, expect):
    """Round 3 gave the peer exchange up together with the library's RCCL communicator: without
    it there was no trial.  Now the peer exchange is held against torch.distributed's
    all-reduce instead, and the loser of THAT comparison is what runs."""
    from blueberry_amd import solver
    monkeypatch.setattr(one_rank_group, "get_backend", lambda *a, **k: "nccl")
    e = _ScriptedTorchEngine(**kwargs)
    x0 = e.get_coords()
    assert solver.select_exchange(e, 0.5, trial=True) == expect
    assert e._comm_trial["reference"] == "torch" and "torch" in e.calls and "rccl" not in e.calls
    assert numpy.array_equal(e.get_coords(), x0)               # the start was restored


def test_select_exchange_without_trial_is_rccl(one_rank_group, monkeypatch):
    """fit() never picks a transport by timing: auto = the library's RCCL communicator,
    torch.distributed if that cannot be made; BB_COMM_TRIAL=1 opts in to the trial."""
    from blueberry_amd import solver
    monkeypatch.setattr(one_rank_group, "get_backend", lambda *a, **k: "nccl")
    e = _ScriptedEngine(peer_sleep=0.0, rccl_sleep=0.002)
    assert solver.select_exchange(e, 0.5) == "rccl" and e.calls == []
    assert solver.select_exchange(_ScriptedEngine(rccl_ok=False), 0.5) == "torch"
    monkeypatch.setenv("BB_COMM_TRIAL", "1")
    assert solver.select_exchange(_ScriptedEngine(rccl_sleep=0.002), 0.5) == "peer"


def _trial_worker(rank, world, port, fail_rank, fail_leg, q):
    """One rank of a world-2 trial whose `fail_rank` raises at the start of `fail_leg`,
    BEFORE it joins that transport's collective -- the other rank is already inside."""
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["BB_TRIAL_SYNC_TIMEOUT_MS"] = "1500"
        os.environ.pop("BB_COMM", None)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        data = dist.new_group(backend="gloo")        # the transport's own communicator
        dist.get_backend = lambda *a, **k: "nccl"    # take select_exchange's RCCL branch
        from blueberry_amd import solver
        from tests._engines import ScriptedRankEngine
        e = ScriptedRankEngine(rank, world, data,
                               fail_leg=fail_leg if rank == fail_rank else None)
        x0 = e.get_coords()
        state = solver.select_exchange(e, 0.5, trial=True)
        q.put((rank, state, e._comm_trial, e.aborted, bool(numpy.array_equal(e.get_coords(), x0))))
    except Exception:
        q.put((rank, "FAILED " + traceback.format_exc(), None, None, None))
    q.close()
    q.join_thread()      # the result must have left before the hard exit
    os._exit(0)          # a collective nobody will ever complete may still be pending


@pytest.mark.parametrize("fail_leg,expect,aborted", [("rccl", "torch", True),
                                                     ("peer", "rccl", False),
                                                     (None, "peer", False)])
def test_trial_with_a_rank_that_fails_before_its_collective(fail_leg, expect, aborted):
    """a rank whose warm-up raised used to skip the timed collective
    while the others entered it.  Now every stage ends with an agreement, waits on the
    device are bounded, and a communicator with a collective in flight is aborted: both
    ranks come out with the SAME transport, and nobody hangs."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trial_worker, args=(r, 2, port, 1, fail_leg, q))
             for r in range(2)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=30)
    for r in results:
        assert not r[1].startswith("FAILED"), r[1]
    assert [r[1] for r in results] == [expect, expect]
    for rank, state, trial, was_aborted, restored in results:
        assert restored
        assert was_aborted == aborted
        assert trial["agree"] == (fail_leg is None)
        if fail_leg and rank == 1:                 # the rank that failed says why
            assert fail_leg in trial["error"]


def _reuse_worker(rank, world, port, q):
    """Three 'fits' of one job: select_exchange on three fresh engines whose communicator
    cache is a dict that plays the library's (bb_comm_cached / _attach / _detach).  Before the
    third, rank 1 loses its cached communicator."""
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.pop("BB_COMM", None)
        os.environ.pop("BB_COMM_TRIAL", None)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from blueberry_amd import solver
        from tests._engines import ScriptedRankEngine

        cache = {}                      # the library's process-wide cache, played

        class Engine(ScriptedRankEngine):
            setups = 0

            def comm_setup(self):
                Engine.setups += 1
                cache["comm"] = "made by fit %d" % Engine.setups
                self.comm = cache["comm"]
                return True

            def _comm_cached(self):
                return "comm" in cache

            def _comm_attach(self):
                self.comm = cache["comm"]
                return True

            def _comm_detach(self):
                self.comm = None

        # select_exchange must take its RCCL branch; the agreement (all_gather_object) runs
        # over the real gloo group
        dist.get_backend = lambda *a, **k: "nccl"
        out = []
        for fit in range(3):
            if fit == 2 and rank == 1:
                cache.clear()
            e = Engine(rank, world, None)
            state = solver.select_exchange(e, 0.5)
            out.append((state, Engine.setups, e.comm))
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "FAILED " + traceback.format_exc()))


def test_two_fits_make_one_communicator():
    """every multi-rank fit() made a fresh RCCL communicator.  Now
    the first fit of a job makes it (comm_setup), later ones borrow it from the library's
    cache (solver.comm_reuse) -- and when one rank has lost its copy, EVERY rank makes a
    new one: the ranks agree before anybody attaches."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reuse_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    for rank in (0, 1):
        assert not isinstance(results[rank], str), results[rank]
        (s1, n1, c1), (s2, n2, c2), (s3, n3, c3) = results[rank]
        assert (s1, s2, s3) == ("rccl", "rccl", "rccl")
        assert (n1, n2) == (1, 1) and c2 == c1 == "made by fit 1"     # the second fit reused it
        assert n3 == 2 and c3 == "made by fit 2"                       # both ranks, not just rank 1


def _generation_worker(rank, world, port, q):
    """every rank answers 'cached and free', but the communicators are of
    DIFFERENT generations -- rank 0 replaced its stale entry while rank 1's earlier solver
    still held the old one.  Attaching them would hang the first all-reduce: comm_reuse
    compares the generations, nobody attaches, every rank makes a fresh one."""
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.pop("BB_COMM", None)
        os.environ.pop("BB_COMM_TRIAL", None)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from blueberry_amd import solver
        from tests._engines import ScriptedRankEngine

        cache = {}
        attached = []

        class Engine(ScriptedRankEngine):
            setups = 0

            def comm_setup(self):
                Engine.setups += 1
                cache["comm"], cache["gen"] = "made by setup %d" % Engine.setups, 100 + Engine.setups
                self.comm = cache["comm"]
                return True

            def _comm_cached(self):
                return "comm" in cache

            def _comm_generation(self):
                return cache.get("gen", 0)

            def _comm_attach(self):
                attached.append(cache["gen"])
                self.comm = cache["comm"]
                return True

            def _comm_detach(self):
                self.comm = None

        dist.get_backend = lambda *a, **k: "nccl"
        out = []
        e = Engine(rank, world, None)
        out.append((solver.select_exchange(e, 0.5), Engine.setups, e.comm))     # fit 1: made
        e = Engine(rank, world, None)
        out.append((solver.select_exchange(e, 0.5), Engine.setups, e.comm))     # fit 2: reused
        if rank == 0:                     # rank 0 now holds a communicator of another generation
            cache["comm"], cache["gen"] = "someone else's", 7
        e = Engine(rank, world, None)
        out.append((solver.select_exchange(e, 0.5), Engine.setups, e.comm))     # fit 3: fresh
        q.put((rank, out, attached))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "FAILED " + traceback.format_exc(), None))


def test_cached_communicators_of_different_generations_are_not_attached():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_generation_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {r[0]: r for r in (q.get(timeout=120) for _ in procs)}
    for p in procs:
        p.join(timeout=30)
    for rank in (0, 1):
        assert not isinstance(results[rank][1], str), results[rank][1]
        (s1, n1, c1), (s2, n2, c2), (s3, n3, c3) = results[rank][1]
        assert (s1, s2, s3) == ("rccl", "rccl", "rccl")
        assert (n1, n2) == (1, 1) and c2 == c1                     # same generation: reused
        assert n3 == 2 and c3 == "made by setup 2"                 # mismatch: fresh on BOTH ranks
        assert results[rank][2] == [101]                           # the only attach was fit 2's


def test_select_exchange_overrides(one_rank_group, monkeypatch):
    from blueberry_amd import solver
    monkeypatch.setattr(one_rank_group, "get_backend", lambda *a, **k: "nccl")
    for want, kwargs, expect in (("peer", {}, "peer"), ("rccl", {}, "rccl"),
                                 ("rccl", dict(rccl_ok=False), "torch"), ("torch", {}, "torch"),
                                 ("host", {}, "host")):
        monkeypatch.setenv("BB_COMM", want)
        e = _ScriptedEngine(**kwargs)
        assert solver.select_exchange(e, 0.5) == expect
        assert e.calls == []                                   # no trial when the choice is forced
    monkeypatch.setenv("BB_COMM", "peer")
    with pytest.raises(RuntimeError, match="could not be set up"):
        solver.select_exchange(_ScriptedEngine(peer_ok=False), 0.5)
    monkeypatch.setenv("BB_COMM", "bogus")
    with pytest.raises(ValueError):
        solver.select_exchange(_ScriptedEngine(), 0.5)


def test_select_exchange_on_gloo_is_host_staged(one_rank_group):
    from blueberry_amd import solver
    e = _ScriptedEngine()
    assert solver.select_exchange(e, 0.5) == "host" and e.calls == []
