#!/usr/bin/env python
"""bench.py -- pair-updates/s of the 3D-structure solver on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bins 50000] [--dtype float32]
    python bench.py --workload genome10kb [--gpus N] ...      BASELINE config 5 (see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` without a launcher starts its own N ranks -- as child
processes, before this process has made a single GPU call -- and relays rank 0's
line and the children's exit code.

A "step" is one solver iteration (stress + gradient over every bin pair, then
the coordinate update) on the workload BASELINE.json's metric is quoted on:
a dense synthetic N = 50,000-bin wish-distance matrix, fp32, resident in HBM
(generated on the device from a seeded random walk).
N GPUs split the SAME matrix (strong scaling): each rank owns a contiguous
1/N of the packed units and the ranks exchange one all-reduce of the
(3*n_pad+2)-element gradient buffer per step (RCCL over xGMI).

--workload genome10kb is BASELINE config 5: the whole hg19 genome at 10 kb bins (N = 309,568)
as blocked-sparse 512 x 512 tiles -- one block per chromosome + a 10-Mb band, 10,013 of 183,315
upper tiles, 2.54 G stored pairs (the reference's dense matrix, blueberry/datatypes.pyx:99, would
be 720 GB) -- with pairs = stored pairs, the roofline on the stored-pair bytes, and the CPU
baseline on the same block structure at 50 kb.

Order of the measurements: convergence legs, read sweep, --settle-ms of untimed iterations
(the chip's clocks need load to settle), W warm-up steps, the K timed
steps (`value`, `ms_per_step`, `roofline`), then --reps more K-step blocks for the spread.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel
(stress_grad_kernel) by its ALGORITHMIC bytes -- 4 B (one fp32 wish distance)
per pair-update, SURVEY.md 8(d) -- over its HIP-event duration measured on the
solver's own stream inside the timed region.  `cpu_baseline` times this
repository's CPU oracle (the reference has no solver to time) on a bounded
sample, one core.  reference parity: N/A -- path absent in reference.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bins", type=int, default=None,
                    help="bins of the map (default: 50000 dense, 309568 genome10kb; another "
                         "value rescales the genome's chromosome proportions)")
    ap.add_argument("--workload", default="dense", choices=["dense", "genome10kb"],
                    help="dense: BASELINE's headline, a dense N-bin wish matrix (configs 2-4). "
                         "genome10kb: BASELINE config 5, the whole hg19 genome at 10 kb bins "
                         "(N = 309,568) as blocked-sparse tiles -- every chromosome's own block "
                         "plus a band of --band-bins around the diagonal; pairs = stored pairs")
    ap.add_argument("--band-bins", type=int, default=1000,
                    help="genome10kb: tiles holding a pair of bins at most this far apart are "
                         "kept across chromosome borders too (1000 bins = 10 Mb = the reference's "
                         "HIGH_FITHIC_CUTOFF, blueberry/utils.py:25)")
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-bins", type=int, default=None,
                    help="bins of the cpu_baseline sample (default 10000 dense; genome10kb: "
                         "61914 = the same genome at 50 kb bins)")
    ap.add_argument("--cpu-iters", type=int, default=100)
    ap.add_argument("--converge-steps", type=int, default=60)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong (default, what the headline is): the same --bins matrix split "
                         "over the ranks; weak: bins grow with sqrt(ranks) so that every rank "
                         "keeps the 1-GPU number of pairs")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the multi-rank path with ranks sharing GPUs "
                         "(host-staged all-reduce); the driver's runs use nccl = RCCL")
    ap.add_argument("--momentum", type=float, default=0.5,
                    help="heavy-ball coefficient of the second time-to-converged-stress leg")
    ap.add_argument("--relax", type=float, default=1.8,
                    help="third leg: step = relax / (2 N), an over-relaxed majorisation step")
    ap.add_argument("--relax-momentum", type=float, default=0.4,
                    help="heavy-ball coefficient of the third leg")
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="untimed iterations run for this long right before the W warm-up "
                         "steps, so that the timed block sees settled clocks (0 = none)")
    ap.add_argument("--event-stride", type=int, default=8,
                    help="HIP events around the kernels of every k-th timed step (0 = none: "
                         "then roofline.kernel_ms is 0)")
    ap.add_argument("--reps", type=int, default=5,
                    help="further repetitions of the --steps block after the timed one, for "
                         "the spread of ms_per_step (0 = none)")
    ap.add_argument("--check-every", type=int, default=1,
                    help="measured time-to-converged-stress legs: the stress history is read "
                         "back (one sync + D2H, ~30 us) every this many steps, as fit(tol=..., "
                         "check_every=...) does; at N=50k a step is 0.8 ms, so checking every "
                         "step costs less than stopping up to k-1 steps late")
    a = ap.parse_args()
    if a.bins is None:
        a.bins = 50000 if a.workload == "dense" else 309568
    if a.cpu_bins is None:
        a.cpu_bins = 10000 if a.workload == "dense" else 61914
    return a


def self_launch(a):
    """--gpus N with no launcher around us: become the launcher.  The N ranks are
    child processes of a `torch.distributed.run` started HERE, before this process
    has touched the GPU (bench.py imports neither torch nor the HIP library up to this
    point), so nothing that holds a GPU context is ever re-executed.  Rank 0's JSON
    line is relayed on stdout, everything else goes to stderr, and the exit code is
    the launcher's."""
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line:
        print(line)
        sys.stdout.flush()
    if rc == 0 and not line:
        sys.stderr.write("bench.py: the ranks exited cleanly but printed no result line\n")
        rc = 1
    sys.exit(rc)


def random_walk(n, seed=0):
    x = numpy.cumsum(numpy.random.default_rng(seed).standard_normal((n, 3)), axis=0)
    return x - x.mean(axis=0)


def host_cores():
    """Host cores this process should use for the CPU baseline: its affinity mask,
    cut to the cgroup's CPU quota if there is one, and to 16 -- the CPU share of a
    one-GPU box of this pool (more threads than that only time-slice)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(cores, 16))


def cpu_baseline(n, iters):
    """The oracle's C loop on a bounded sample of the same workload shape, on the
    host cores of this box: oracle/bb_oracle_mt.c (OpenMP, every core this process
    may use), with the scalar oracle/bb_oracle.c bbo_solve timed next to it on a
    shorter sample.  kind = "port": the reference has no solver, so there is no
    reference binary to time."""
    from tests import _oracle
    o = _oracle.load()
    xs = random_walk(n)
    w = numpy.empty((n, n))
    for a in range(0, n, 1000):               # chunked: no (n,n,3) temporary
        d = xs[a:a + 1000, None, :] - xs[None, :, :]
        w[a:a + 1000] = numpy.sqrt((d * d).sum(-1))
    x0 = xs + 0.5 * numpy.random.default_rng(1).standard_normal(xs.shape)
    pairs = n * (n - 1) // 2
    it1 = max(1, iters // 4)
    t0 = time.perf_counter()
    o.solve(w, x0, it1, 1.0 / (2 * n), f64=False)
    dt1 = time.perf_counter() - t0
    one = pairs * it1 / dt1 / 1e9
    out = {"value": one, "unit": "Gpair-updates/s", "cores": 1, "kind": "port",
           "sample": "oracle bbo_solve, N=%d dense, %d iterations, %.1f s, gcc -O2, 1 thread"
                     % (n, it1, dt1)}
    cores = host_cores()
    if cores > 1:
        try:
            _oracle.solve_mt(w, x0, 2, 1.0 / (2 * n), cores, f64=False)     # threads, pages
            t0 = time.perf_counter()
            _oracle.solve_mt(w, x0, 4, 1.0 / (2 * n), cores, f64=False)     # sizes the sample
            per_iter = (time.perf_counter() - t0) / 4
            itm = int(min(max(8.0 / per_iter, 4), 100 * iters))            # about 8 s
            t0 = time.perf_counter()
            _oracle.solve_mt(w, x0, itm, 1.0 / (2 * n), cores, f64=False)
            dtm = time.perf_counter() - t0
            out = {"value": pairs * itm / dtm / 1e9, "unit": "Gpair-updates/s", "cores": cores,
                   "kind": "port",
                   "sample": "oracle bbo_solve_mt (OpenMP, rows dealt cyclically), N=%d dense, "
                             "%d iterations, %.1f s, gcc -O2 -fopenmp, %d threads"
                             % (n, itm, dtm, cores),
                   "single_core": {"value": one, "sample": out["sample"]}}
        except (OSError, MemoryError, subprocess.CalledProcessError):
            pass                                  # no libgomp here: the scalar figure stands
    return out


def cpu_baseline_genome(n, band_bins, iters):
    """The config-5 workload on the host cores: the SAME block structure (hg19 chromosome
    proportions, one block per chromosome + a band) at a bounded size -- by default the
    genome at 50 kb bins, N = 61,914, band 10 Mb -- through the oracle's tile-list loop
    (oracle/bb_oracle_mt.c bbo_solve_gen_mt: wish distances formed on the fly from the
    generating walk, only stored pairs are visited and counted).  kind = "port"."""
    from tests import _oracle
    from blueberry_amd.solver import max_degree, tiles_from_blocks
    from blueberry_amd.utils import genome_boundaries
    tiles, pairs = tiles_from_blocks(n, genome_boundaries(n), band_bins, "float32")
    xs = random_walk(n)
    x0 = xs + 0.5 * numpy.random.default_rng(1).standard_normal(xs.shape)
    lr = 1.0 / (2 * max_degree(n, tiles, "float32"))
    what = ("hg19 chromosome blocks + %d-bin band at N=%d (%d tiles of 512, %d stored pairs)"
            % (band_bins, n, len(tiles[0]), pairs))
    t0 = time.perf_counter()
    _oracle.solve_gen_mt(xs, x0, 1, lr, 1, tiles=tiles, f64=False)
    per1 = time.perf_counter() - t0
    it1 = int(min(max(3.0 / per1, 1), iters))
    t0 = time.perf_counter()
    _oracle.solve_gen_mt(xs, x0, it1, lr, 1, tiles=tiles, f64=False)
    dt1 = time.perf_counter() - t0
    one = pairs * it1 / dt1 / 1e9
    out = {"value": one, "unit": "Gpair-updates/s", "cores": 1, "kind": "port",
           "sample": "oracle bbo_solve_gen_mt, %s, %d iterations, %.1f s, gcc -O2, 1 thread"
                     % (what, it1, dt1)}
    cores = host_cores()
    if cores > 1:
        _oracle.solve_gen_mt(xs, x0, 2, lr, cores, tiles=tiles, f64=False)
        t0 = time.perf_counter()
        _oracle.solve_gen_mt(xs, x0, 4, lr, cores, tiles=tiles, f64=False)
        per_iter = (time.perf_counter() - t0) / 4
        itm = int(min(max(8.0 / per_iter, 4), 100 * iters))
        t0 = time.perf_counter()
        _oracle.solve_gen_mt(xs, x0, itm, lr, cores, tiles=tiles, f64=False)
        dtm = time.perf_counter() - t0
        out = {"value": pairs * itm / dtm / 1e9, "unit": "Gpair-updates/s", "cores": cores,
               "kind": "port",
               "sample": "oracle bbo_solve_gen_mt (OpenMP, tiles dealt cyclically), %s, "
                         "%d iterations, %.1f s, gcc -O2 -fopenmp, %d threads"
                         % (what, itm, dtm, cores),
               "single_core": {"value": one, "sample": out["sample"]}}
    return out


PMC_TABLE = "profiles/pmc_latest.json"


def pmc_traffic(n_bins, dtype, world, workload="dense"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC
    passes on this configuration (profiles/pmc_latest.json: one entry per problem
    size, written by tools/tools_pmc.sh), or None.  It is replayed from the profile,
    not measured in this run: counters need their own rocprofv3 passes."""
    p = os.path.join(ROOT, PMC_TABLE)
    try:
        with open(p) as fh:
            d = json.load(fh)
        for e in d.get("entries", [d]):
            if (e.get("bins") == n_bins and e.get("dtype") == dtype and e.get("gpus", 1) == world
                    and e.get("workload", "dense") == workload):
                return e.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


class stdout_to_stderr(object):
    """RCCL prints a version banner on fd 1 when its first communicator is made;
    keep fd 1 for the one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1 and "WORLD_SIZE" not in os.environ:
            self_launch(a)                     # does not return
        a.gpus = world

    torch = dist = None
    # BB_BENCH_FORCE_DIST=1: run the grad / all-reduce / apply path even with one
    # rank (rehearses the RCCL plumbing on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("BB_BENCH_FORCE_DIST") == "1"
    if use_dist:                               # torch is plumbing for the collective only
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        n_dev = torch.cuda.device_count()      # counting does not initialise the GPU
        if world > 5 * max(1, n_dev):
            # a rehearsal with ranks sharing GPUs: the boxes of this pool allow six processes
            # per GPU and the launcher is one of them -- stop before any rank opens the device
            raise SystemExit("bench.py: %d ranks on %d GPU(s): at most 5 ranks may share a GPU "
                             "here (tools/peer_sequence_fuzz.py rehearses 8 ranks in ONE process)"
                             % (world, n_dev))
        if a.backend == "nccl" and world > max(1, n_dev):
            # more ranks than GPUs (a rehearsal on a smaller box): RCCL refuses two ranks
            # on one device, so the sum over ranks goes through gloo and host memory
            a.backend = "gloo"
        local_rank = local_rank % max(1, n_dev)
        torch.cuda.set_device(local_rank)
        with stdout_to_stderr():
            if a.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local_rank))
                warm = torch.zeros(1, device="cuda")
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
                warm = torch.zeros(1)
            dist.all_reduce(warm)             # creates the communicator now
            torch.cuda.synchronize()

    from blueberry_amd.solver import HipEngine, run_iterations, select_exchange

    n = a.bins if a.scaling == "strong" else int(round(a.bins * world ** 0.5))
    xs = random_walk(n, 0)
    x0 = xs + 0.5 * numpy.random.default_rng(1).standard_normal(xs.shape)
    tiles = None
    pairs = n * (n - 1) // 2
    lr = 1.0 / (2 * n)
    degree_legs = False
    if a.workload == "genome10kb":
        # BASELINE config 5: the dense (n_bins+1)^2 float64 matrix of the reference
        # (blueberry/datatypes.pyx:99) would be 720 GB; the tiles a Hi-C map populates -- each
        # chromosome's own block and a band along the diagonal -- are 10.5 GB in fp32
        from blueberry_amd.solver import max_degree, tiles_from_blocks
        from blueberry_amd.utils import genome_boundaries
        if a.scaling != "strong":
            raise SystemExit("bench.py: --workload genome10kb is a fixed map (strong scaling)")
        tiles, pairs = tiles_from_blocks(n, genome_boundaries(n), a.band_bins, a.dtype)
        lr = 1.0 / (2 * max_degree(n, tiles, a.dtype))
        degree_legs = True                    # SPEC 2.4.1: a step per bin in the convergence legs
    eng = HipEngine(n, a.dtype, rank=rank, world=world, device=local_rank, tiles=tiles)
    eng.set_wish_from_coords(xs)          # delta_ij = |x*_i - x*_j| generated in HBM
    eng.set_coords(x0)
    if use_dist:
        # the transport of the per-iteration sum is chosen here, once: the peer
        # exchange inside the solver's own kernels if it validates against RCCL and
        # is faster, else the library's RCCL communicator (BB_COMM overrides)
        with stdout_to_stderr():
            select_exchange(eng, lr, trial=True)

    def steps(k, step=None):
        # world 1: one C call enqueues k fused iterations.  world > 1: grad ->
        # reduce -> sum over ranks -> update per iteration, also enqueued from C
        run_iterations(eng, k, lr if step is None else step, 2 if use_dist else 1)

    def fence():
        eng.sync()
        if eng._comm_state == "peer":
            eng.peer_status()                  # a timed-out exchange must not pass as a result
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    # Order of the measurements: the convergence legs and the read sweep run FIRST, then
    # --settle-ms of untimed iterations, then the W warm-up steps and the timed block.
    # After an idle spell (input generation, host-side set-up) the chip needs a while
    # under load before its clocks settle (DVFS, MI355X_MICROARCH.md "DVFS give-back"): a
    # block timed right after idle ran 4-19 % slower than the same block a moment later
    # (ms_per_step_reps).  A fit() runs hundreds of iterations back to back, so the
    # settled figure is the one that counts.
    # BASELINE metric, second half: wall-clock from a resident matrix and the
    # noisy start X0 to S_k / S_0 <= 1e-3 (the synthetic matrix has a zero-stress
    # solution).  Run a fixed number of steps, then read k* off the history.
    def converge_leg(mu, relax=1.0):
        eng.set_timing(False)
        eng.set_coords(x0)
        eng.set_momentum(mu)
        fence()
        t1 = time.perf_counter()
        steps(a.converge_steps, relax * lr)
        fence()
        dtc = time.perf_counter() - t1
        h2 = eng.stress_history()
        below = numpy.nonzero(h2 <= 1e-3 * h2[0])[0]
        if below.size:
            kstar = int(below[0])            # S_k is the stress BEFORE step k: k steps were needed
            return {"iterations": kstar, "ms": kstar * dtc / a.converge_steps * 1e3,
                    "stress_ratio": float(h2[kstar] / h2[0]), "momentum": mu,
                    "lr_times_2N": relax}
        return {"iterations": None, "ms": None, "stress_ratio": float(h2[-1] / h2[0]),
                "momentum": mu, "lr_times_2N": relax}

    def converge_measured(mu, relax=1.0, spectral=False, spectral_tol=0.0):
        """The same leg TIMED END TO END the way the product stops early
        (StructureSolver.fit(tol=...), blueberry_amd/solver.py): --check-every steps are
        enqueued, the stress history is read back (a sync + a D2H), and the loop ends at the
        first read that shows S_k <= 1e-3 S_0.  The wall clock includes every read-back and,
        with `spectral`, the classical-MDS start computed on the device; S_0 is the stress of
        the noisy start X0 either way."""
        eng.set_timing(False)
        eng.set_coords(x0)
        eng.set_momentum(mu)
        s0 = None
        if spectral:
            s0 = eng.stress()                    # S(X0): the yardstick, before the clock starts
            # one untimed call first, as the iteration legs come after --warmup steps: the
            # first call loads the start's kernels and allocates its buffers (about 9 ms)
            eng.spectral_init_device(2, numpy.random.default_rng(0).standard_normal((n, 3)),
                                     tol=spectral_tol)
            eng.set_coords(x0)
        fence()
        t1 = time.perf_counter()
        products = None
        if spectral:
            products = eng.spectral_init_device(
                40, numpy.random.default_rng(0).standard_normal((n, 3)), tol=spectral_tol)[0] + 1
        done, reads, hit = 0, 0, None
        while done < a.converge_steps:
            k = min(a.check_every, a.converge_steps - done)
            steps(k, relax * lr)
            done += k
            h2 = eng.stress_history()            # synchronises: the read-back is in the figure
            reads += 1
            if s0 is None:
                s0 = h2[0]
            if h2[-1] <= 1e-3 * s0:
                hit = int(numpy.nonzero(h2 <= 1e-3 * s0)[0][0])
                break
        fence()
        dtc = time.perf_counter() - t1
        return {"measured_ms": dtc * 1e3 if hit is not None else None, "iterations_run": done,
                "first_iteration_below": hit, "stress_reads": reads,
                "check_every": a.check_every, "momentum": mu, "lr_times_2N": relax,
                "start": ("spectral (block power iteration on the device, at most 41 products, %s; "
                          "inside the clock, after one untimed warm-up call)" % ("ended by spectral_tol=%g" % spectral_tol
                                                 if spectral_tol else "all of them made"))
                         if spectral else "noisy X0",
                **({"spectral_products": products} if spectral else {})}

    conv = conv_mu = conv_relaxed = conv_spectral = None
    if a.converge_steps > 0:
        conv = converge_leg(0.0)             # the plain step the throughput figure is timed on
        conv_mu = converge_leg(a.momentum)   # heavy-ball, SPEC 2.4
        # over-relaxed majorisation step lr = omega / 2N (omega < 2) + heavy-ball, SPEC 2.4
        conv_relaxed = converge_leg(a.relax_momentum, a.relax)
        # ... and the same three with the early stop really running (wall clock incl. read-backs)
        conv.update(converge_measured(0.0))
        conv_mu.update(converge_measured(a.momentum))
        conv_relaxed.update(converge_measured(a.relax_momentum, a.relax))
        if degree_legs:
            # a blocked-sparse map: the same three legs with a step per bin from the map's own
            # degrees (bb_solver_degrees + bb_solver_set_bin_steps = StructureSolver(
            # degree_steps=True); SPEC 2.4.1) -- every chromosome's bins step by their own
            # 1 / (2 (degree + 1)), not by the largest one's.  The degree pass and the upload
            # of the factors are timed too (once per map).
            from blueberry_amd.solver import degree_step_factors, _sum_over_ranks
            setup_ms = []
            for _ in range(2):                # the first call of the process, then a warm one
                fence()
                t1 = time.perf_counter()
                deg = _sum_over_ranks(eng.degrees(), eng, world)
                lr_deg, factors = degree_step_factors(deg)
                eng.set_bin_steps(factors)
                fence()
                setup_ms.append((time.perf_counter() - t1) * 1e3)
            lr_uniform, lr = lr, lr_deg
            for leg, (mu_, rx_) in ((conv, (0.0, 1.0)), (conv_mu, (a.momentum, 1.0)),
                                    (conv_relaxed, (a.relax_momentum, a.relax))):
                per_bin = converge_leg(mu_, rx_)
                per_bin.update(converge_measured(mu_, rx_))
                per_bin["degree_pass_and_factors_ms"] = {"first_call": setup_ms[0], "warm": setup_ms[1]}
                per_bin["degree_min_max"] = [int(deg.min()), int(deg.max())]
                leg["degree_steps"] = per_bin
            lr = lr_uniform
            eng.set_bin_steps(None)
        if world == 1 and not use_dist:
            # the product's default (StructureSolver.spectral_tol = 1e-3); the fixed 41
            # products of rounds 3-4 beside it
            conv_spectral = converge_measured(0.0, spectral=True, spectral_tol=1e-3)
            conv_spectral["all_41_products"] = converge_measured(0.0, spectral=True)
        eng.set_momentum(0.0)
    read_ms = eng.stream_read_ms(10) if a.dtype == "float32" else None
    eng.set_coords(x0)                       # the timed block starts where the legs did
    eng.set_momentum(0.0)
    eng.set_timing(1)                        # creates the HIP events now (host work), ...
    eng.set_timing(False)                    # ... not between the warm-up and the timed block
    # settle: the number of blocks is agreed between the ranks (every step of a multi-rank
    # run is a collective: a loop that runs "for 300 ms" on each rank's own clock would
    # leave the ranks with different step counts, i.e. inside different collectives)
    blk = max(10, a.steps)
    fence()
    t_settle = time.perf_counter()
    steps(blk)
    fence()
    t_blk = time.perf_counter() - t_settle
    if use_dist:
        t = torch.tensor([t_blk], dtype=torch.float64,
                         device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_blk = float(t.item())
    n_settle = int(min(200, max(0, round(a.settle_ms * 1e-3 / max(t_blk, 1e-6)) - 1)))
    for _ in range(n_settle if a.settle_ms > 0 else 0):
        steps(blk)                           # untimed; speed does not depend on the state
        eng.sync()                           # (tools/state_probe.py), clocks do on the load
    steps(a.warmup)
    fence()
    # HIP events on a sample of the steps of the timed region: three records cost ~10 us of
    # stream time, too much to put on every step once a step is ~0.1 ms (8 GPUs).  At least
    # 8 launches are sampled whatever --steps is (4 on several ranks, where a step is short),
    # so that the sampled launches are the average ones: with --steps 20 and a fixed stride
    # of 8 round 2 timed 3 launches, and kernel + reduce came out above the device step.
    want = 8 if world == 1 else 4
    stride = max(1, min(a.event_stride, a.steps // want)) if a.event_stride > 0 else 0
    eng.set_timing(stride)
    t0 = time.perf_counter()
    steps(a.steps)
    fence()
    dt = time.perf_counter() - t0
    tim = eng.timing()
    # For the record: what two HIP events recorded back to back cost on this stream.  Every
    # event-timed interval of a sampled step contains a part of that (rocprofv3's dispatch
    # durations of the same kernels come out 2-4 us shorter), which is why kernel_ms +
    # reduce_update_ms of the sampled steps can exceed device_step_ms -- the start-to-start
    # time averaged over ALL steps between two samples -- by a few microseconds.  The figures
    # are reported as the events have them: nothing is subtracted.
    gap_ms = eng.event_gap_ms(16) if tim["launches"] > 0 else 0.0
    dt_mine = dt
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64,
                         device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # what every rank saw: kernel, reduce, whole device-side step, its own wall clock
    mine = {"rank": rank, "kernel_ms": tim["grad_ms"], "reduce_ms": tim["reduce_ms"],
            "device_step_ms": tim["step_ms"],
            "exchange_update_gaps_ms": max(0.0, tim["step_ms"] - tim["grad_ms"] - tim["reduce_ms"])
            if tim["step_ms"] > 0 else None,
            "wall_ms_per_step": dt_mine / a.steps * 1e3}
    if eng._comm_state == "peer" and eng.peer_form() == "one launch":
        # the exchange (push, wait for the peers, sum, update) is INSIDE the reduce launch
        mine["reduce_includes_exchange"] = True
    per_rank = [mine]
    if use_dist and world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    # spread: the same K-step block a few more times (the timed one above is `value`)
    eng.set_timing(False)
    reps = []
    for _ in range(max(0, a.reps)):
        fence()
        t1 = time.perf_counter()
        steps(a.steps)
        fence()
        r = time.perf_counter() - t1
        if use_dist:
            t = torch.tensor([r], dtype=torch.float64,
                             device="cuda" if a.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            r = float(t.item())
        reps.append(r / a.steps * 1e3)

    hist = eng.stress_history()
    traffic = eng.traffic()
    path, waves_per_row = eng.iteration_path()
    eng_comm = {"peer": "peer exchange (one-shot, in-kernel, IPC arenas over xGMI)",
                "rccl": "library-owned RCCL communicator", "torch": "torch.distributed (RCCL)",
                "host": "gloo, host-staged (rehearsal)", None: "none"}[eng._comm_state]
    comm_trial = eng._comm_trial
    rccl_world = eng.comm_world() if eng._comm_state == "rccl" else None
    if eng._comm_state == "peer":
        # "one launch": reduce + push + wait + sum + update per workgroup; "two launches"
        # where ranks share a GPU (include/blueberry_hip.h)
        eng_comm += ", " + eng.peer_form()

    eng.close()
    if rank == 0:
        es = 4 if a.dtype == "float32" else 8
        value = pairs * a.steps / dt / 1e9
        # dominant kernel: algorithmic bytes this rank's launch streams / its duration
        alg_bytes = pairs * es / float(world)
        achieved = alg_bytes / (tim["grad_ms"] * 1e-3) / 1e9 if tim["grad_ms"] > 0 else 0.0
        tname = "float" if es == 4 else "double"
        kernel = ("stress_grad_kernel<%s>" % tname if path == "units" else
                  "row_owner_kernel<%s, %d>: one launch per iteration over BOTH triangles, which "
                  "stay in L2 / the Infinity Cache -- launch-bound, the HBM roofline does not apply"
                  % (tname, waves_per_row))
        out = {
            "metric": ("Gpair-updates/s per stress iteration, N=%d blocked-sparse tiles "
                       "(whole genome at 10 kb bins, stored pairs)" % n)
                      if tiles is not None else
                      "Gpair-updates/s per stress iteration, N=50k" if n == 50000 else
                      "Gpair-updates/s per stress iteration, N=%d" % n,
            "value": value,
            "unit": "Gpair-updates/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            # the same K-step block repeated --reps more times after the timed one
            "ms_per_step_reps": {"n": len(reps), "min": min(reps), "median": sorted(reps)[len(reps) // 2],
                                 "max": max(reps)} if reps else None,
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "f32" if a.dtype == "float32" else "f64",
            "data": "synthetic",
            "config": {"workload": ("genome10kb: %d bins = hg19 at 10 kb, blocked-sparse 512 x 512 "
                                    "tiles (each chromosome's own block + a %d-bin band: %d of %d "
                                    "upper tiles), wish distances from a seeded 3-D random walk "
                                    "generated in HBM, %s; one stress+gradient+update iteration "
                                    "over the stored pairs per step (BASELINE config 5)"
                                    % (n, a.band_bins, len(tiles[0]),
                                       (-(-n // 512)) * (-(-n // 512) + 1) // 2, a.dtype))
                                   if tiles is not None else
                                   "dense %d-bin wish-distance matrix from a seeded 3-D random "
                                   "walk, upper triangle packed in HBM, %s; one stress+gradient+"
                                   "update iteration per step" % (n, a.dtype),
                       "bins": n, "pairs_per_step": pairs,
                       "stored_pairs": pairs if tiles is not None else None,
                       "stored_tiles": len(tiles[0]) if tiles is not None else None,
                       "lr": lr,
                       "parallelism": "unit-range sharding x%d + all-reduce(3*n_pad+2) via %s"
                                      % (world, eng_comm) if world > 1 else "1 gpu",
                       "exchange": eng._comm_state, "exchange_trial": comm_trial,
                       "backend": a.backend if use_dist else None,
                       "rccl_reported_world": rccl_world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(n, a.dtype, world, a.workload),
                         # counters need rocprofv3 passes of their own: replayed, not measured here
                         "traffic_source": ("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                            "this configuration, replayed)" % PMC_TABLE)
                         if pmc_traffic(n, a.dtype, world, a.workload) is not None else None,
                         "kernel": kernel,
                         "kernel_ms": tim["grad_ms"], "reduce_update_ms": tim["reduce_ms"],
                         # two events back to back on this stream: an upper bound of what an
                         # event-timed interval holds besides its kernel (nothing is subtracted)
                         "event_pair_gap_ms": gap_ms,
                         # start-to-start of consecutive steps on the device: what is
                         # left after the two figures above is exchange + update + gaps
                         "device_step_ms": tim["step_ms"],
                         "timed_launches": tim["launches"],
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "resident_bytes_streamed_per_launch": traffic["unit_bytes"],
                         # a read-only sweep of the same units by the same grid on this box
                         "stream_read_ms": read_ms,
                         "stream_read_GBs": (traffic["unit_bytes"] / (read_ms * 1e-3) / 1e9
                                             if read_ms else None),
                         "frac_of_stream_read": (read_ms / tim["grad_ms"]
                                                 if read_ms and tim["grad_ms"] > 0 else None)},
            "ranks": per_rank if world > 1 else None,
            "rank_skew_ms_per_step": (max(r["wall_ms_per_step"] for r in per_rank) -
                                      min(r["wall_ms_per_step"] for r in per_rank))
            if world > 1 else None,
            "stress_first_last": [float(hist[0]), float(hist[-1])] if hist.size else None,
            "time_to_stress_1e-3": conv,
            "time_to_stress_1e-3_momentum": conv_mu,
            "time_to_stress_1e-3_relaxed": conv_relaxed,
            "time_to_stress_1e-3_spectral_start": conv_spectral,
            "reference_parity": "N/A - path absent in reference; parity is against this "
                                "repo's CPU oracle (tests/)",
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = (cpu_baseline(a.cpu_bins, a.cpu_iters) if tiles is None else
                                   cpu_baseline_genome(a.cpu_bins, max(1, a.band_bins * a.cpu_bins // n),
                                                       a.cpu_iters))
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        # A rank that fails must END, at once: the launcher then stops the other ranks.  Left
        # to the interpreter's shutdown it can sit in a communicator's destructor while its
        # peers wait in a collective it will never join.
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            os._exit(1)
        raise
