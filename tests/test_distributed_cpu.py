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
