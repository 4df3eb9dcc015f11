"""Differential fuzz of CALL SEQUENCES on SEVERAL RANKS that share one solver state over the
peer exchange (ranks of this process, arenas connected directly -- the kernels, flags and
arenas are the ones of a multi-process job): random orders of the collective entry points --
bb_solver_iterate_peer in calls of different lengths, the spectral start (whose products go
through the same exchange and sequence numbers), coordinates reset, momentum switched, per-bin
steps set and cleared, the two-call path (grad, host sum, apply) in between -- against the
one-rank numpy / oracle model of tools/api_sequence_fuzz.py.  After every call all ranks must
hold the SAME bits and agree with the model; a time-out (peer_status != 0) is a finding.
Both forms of the exchange, 2, 3 and 8 ranks, both dtypes.  Test infrastructure: uses the oracle.

    python tools/peer_sequence_fuzz.py [n_sequences] [seed]"""
import ctypes
import os
import sys
import threading

import numpy

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from blueberry_amd import _lib                        # noqa: E402
from blueberry_amd.solver import HipEngine            # noqa: E402
from tests import _oracle                             # noqa: E402
from tools.api_sequence_fuzz import Model, close, close_stress      # noqa: E402

DONE = {}


def connect(engs):
    lib = _lib.load()
    blobs = []
    for e in engs:
        buf = ctypes.create_string_buffer(_lib.BB_PEER_HANDLE_BYTES)
        _lib.check(lib.bb_solver_peer_export(e._h, buf), "export")
        blobs.append(buf.raw)
    for e in engs:
        _lib.check(lib.bb_solver_peer_connect(e._h, b"".join(blobs)), "connect")


def one_sequence(rng):
    # (BB_FUZZ_WORLDS=8: more ranks than the HIP runtime has hardware queues by default --
    # GPU_MAX_HW_QUEUES=4 -- deadlock when they are streams of ONE process: a kernel that
    # waits for a peer sits in front of that peer's kernel in the same queue.  Run such a
    # rehearsal with GPU_MAX_HW_QUEUES=16 in the environment.)
    world = int(rng.choice([int(v) for v in os.environ.get("BB_FUZZ_WORLDS", "2,3").split(",")]))
    n = int(rng.choice([700, 1100, 2600]))           # 2,600: above the row-owner switch for the model's twin too
    dtype = str(rng.choice(["float64", "float32"]))
    fused = bool(rng.integers(2))
    if fused and world > 3:
        # ranks that share ONE GPU: the one-launch form's workgroups wait inside the kernel
        # for their peers.  8 ranks x 72 workgroups x 16 waves (n = 2,600) are more waves than
        # the chip holds -- the ranks that are not resident can never deliver -- and already
        # 4-6 ranks' waiting workgroups (16 waves and 16 KB of LDS each) can keep a peer's
        # sweep (150 KB of LDS per workgroup) off every CU: what bb_solver_peer_connect
        # avoids by itself by choosing the two-launch form when ranks share a GPU.  (On 8 GPUs
        # each holds its own 72.)  Forced here, the one-launch form keeps to a size whose
        # waiting workgroups are few.
        n = 700
    os.environ["BB_PEER_FUSED"] = "1" if fused else "0"
    os.environ["BB_PEER_TIMEOUT_MS"] = "20000"
    tol = 1e-10 if dtype == "float64" else 2e-4
    log = ["world=%d n=%d %s %s" % (world, n, dtype, "one launch" if fused else "two launches")]
    engs = [HipEngine(n, dtype, rank=r, world=world) for r in range(world)]
    m = Model(n, dtype)
    lr = 1.0 / (2 * n)
    try:
        connect(engs)
        xs = _oracle.random_walk(n, seed=int(rng.integers(1 << 30)))
        w = _oracle.wish_from_coords(xs)
        if rng.random() < 0.5:
            hole = numpy.triu(rng.random((n, n)) < 0.3, 1)
            w[hole | hole.T] = 0.0
        x0 = _oracle.noisy_init(xs, seed=int(rng.integers(1 << 30)))
        for e in engs:
            e.set_wish_dense(w, "wish", 3.0)
            e.set_coords(x0)
        m.set_wish(w)
        m.set_coords(x0)
        ops = ["iterate", "iterate", "iterate", "coords", "momentum", "bin_steps", "two_call", "spectral",
               "wish", "check"]
        for _ in range(int(rng.integers(5, 14))):
            op = str(rng.choice(ops))
            log.append(op)
            if op == "iterate":
                k = int(rng.integers(1, 5))
                for e in engs:
                    e.iterate_peer(k, lr)
                m.iterate(k, lr)
            elif op == "coords":
                x = rng.standard_normal((n, 3)) * float(rng.choice([1.0, 30.0]))
                for e in engs:
                    e.set_coords(x)
                m.set_coords(x)
            elif op == "momentum":
                mu = float(rng.choice([0.0, 0.3, 0.9]))
                for e in engs:
                    e.set_momentum(mu)
                m.mu = mu
            elif op == "bin_steps":
                sc = None if rng.random() < 0.3 else rng.uniform(0.3, 1.7, n)
                for e in engs:
                    e.set_bin_steps(sc)
                m.per_bin = None if sc is None else sc[:, None]
            elif op == "wish":
                xs = _oracle.random_walk(n, seed=int(rng.integers(1 << 30)))
                w = _oracle.wish_from_coords(xs)
                for e in engs:
                    e.set_wish_dense(w, "wish", 3.0)
                m.set_wish(w)
            elif op == "two_call":
                for e in engs:
                    e.grad()
                total = sum(e.read_exchange() for e in engs)
                for e in engs:
                    e.write_exchange(total)
                    e.apply(lr)
                m.grad()
                m.apply(lr)
            elif op == "spectral":
                if not (m.w[numpy.triu_indices(n, 1)] > 0).all():
                    continue
                v0 = rng.standard_normal((n, 3))
                errs = []

                def run(e):
                    try:
                        e.spectral_init_device(40, v0, tol=1e-3)
                    except Exception as exc:          # noqa: BLE001
                        errs.append(exc)

                ts = [threading.Thread(target=run, args=(e,)) for e in engs]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join(timeout=120)
                if errs:
                    raise AssertionError("spectral start: %r" % errs[0])
                x = engs[0].get_coords()
                if not close(_oracle.wish_from_coords(x), m.w, 1e-6 if dtype == "float64" else 2e-3, m.w.max()):
                    raise AssertionError("spectral start is not the map's embedding")
                m.set_coords(x)
            # after every call: no time-out, ranks bit-identical, equal to the model
            Xs, hs = [], []
            for e in engs:
                if e.peer_status() != 0:
                    raise AssertionError("peer_status != 0 after %s" % op)
                Xs.append(e.get_coords())
                hs.append(e.stress_history())
            for X, h in zip(Xs[1:], hs[1:]):
                if not (numpy.array_equal(X, Xs[0]) and numpy.array_equal(h, hs[0])):
                    raise AssertionError("ranks differ after %s" % op)
            scale = max(1.0, numpy.abs(m.X).max())
            if not close(Xs[0], m.X, tol * 50, scale):
                raise AssertionError("coordinates differ from the model by %g (scale %g) after %s"
                                     % (numpy.abs(Xs[0] - m.X).max(), scale, op))
            if not close_stress(hs[0], numpy.array(m.hist), tol * 50, m):
                raise AssertionError("history differs from the model after %s" % op)
            DONE[op] = DONE.get(op, 0) + 1
        return True, log
    except (AssertionError, RuntimeError) as exc:
        return False, log + ["FAIL: %s" % exc]
    finally:
        for e in engs:
            e.close()


def main():
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = numpy.random.default_rng(seed)
    bad = 0
    for case in range(n_seq):
        ok, log = one_sequence(rng)
        if not ok:
            bad += 1
            print("sequence %d: %s" % (case, " | ".join(log)), flush=True)
        elif case % 10 == 0:
            print("sequence %d ok (%s, %d calls)" % (case, log[0], len(log) - 1), flush=True)
    print("calls checked:", ", ".join("%s %d" % kv for kv in sorted(DONE.items())))
    print("%d sequences, FAILURES: %d" % (n_seq, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
