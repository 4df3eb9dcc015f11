"""count_band_regions -- host side of K1 (reference: blueberry/blueberry.pyx:77-91)."""
import os

import numpy

from . import _lib
from .utils import HIGH_FITHIC_CUTOFF, LOW_FITHIC_CUTOFF


def _as_regions(regions_ndarray):
    """The reference casts `regions_ndarray.data` to `double*` unchecked
    (pyx:80): anything but a C-contiguous float64 vector is read as garbage.
    Deviation: convert instead, and reject non-1-D input."""
    r = numpy.ascontiguousarray(regions_ndarray, dtype=numpy.float64)
    if r.ndim != 1:
        raise ValueError("count_band_regions: regions must be one-dimensional")
    return r


def count_band_regions(regions_ndarray, device=None, distributed=False):
    """Calculate the number of regions in the band.

    Same contract as `blueberry.count_band_regions` (pyx:77-91): the number of
    bin pairs (i, j < i) with LOW_FITHIC_CUTOFF <= regions[i] - regions[j] <=
    HIGH_FITHIC_CUTOFF, as a Python int, computed exactly on the GPU.

    distributed=True (inside an initialised torch.distributed job): rows are
    sharded over the ranks and the shares summed with an integer all-reduce;
    every rank returns the full count.

    device: HIP device index.  None = device 0, or, with distributed=True, this
    rank's own GPU (LOCAL_RANK, as StructureSolver picks it) -- an RCCL job must not
    put every rank's kernel and all-reduce tensor on cuda:0.
    """
    device = pick_device(device, distributed)
    r = _as_regions(regions_ndarray)
    lib = _lib.load()
    n = r.shape[0]
    out = _lib.c_i64(0)
    if not distributed:
        _lib.check(lib.bb_band_count(_lib.as_f64_ptr(r), n, LOW_FITHIC_CUTOFF, HIGH_FITHIC_CUTOFF,
                                     device, out), "bb_band_count")
        return int(out.value)

    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    i_begin, i_end = band_row_share(n, rank, world)
    _lib.check(lib.bb_band_count_rows(_lib.as_f64_ptr(r), n, LOW_FITHIC_CUTOFF,
                                      HIGH_FITHIC_CUTOFF, i_begin, i_end, device, out),
               "bb_band_count_rows")
    return int(allreduce_count(int(out.value), device))


def pick_device(device, distributed):
    """An explicit index wins; else LOCAL_RANK in a distributed call, 0 otherwise."""
    if device is not None:
        return int(device)
    return int(os.environ.get("LOCAL_RANK", "0")) if distributed else 0


def band_row_share(n, rank, world):
    """Rows [i_begin, i_end) of rank `rank`: row i costs i pairs, so cut the
    triangle into `world` bands of equal area (i ~ n * sqrt(r / world))."""
    cut = lambda r: int(round(n * (float(r) / world) ** 0.5))
    i_begin = cut(rank) if rank > 0 else 0
    i_end = cut(rank + 1) if rank + 1 < world else n
    return i_begin, max(i_begin, i_end)


def allreduce_count(local_count, device=0):
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", device) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([local_count], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
