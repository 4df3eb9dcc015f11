"""benjamini_hochberg and downsample -- host side of the two remaining numeric
Cython helpers of the reference (`blueberry/blueberry.pyx:40-75`, `:93-104`)."""
import ctypes

import numpy

from . import _lib


def benjamini_hochberg(p_values, n, device=0):
    """Run the Benjamini-Hochberg procedure on a vector of -sorted- p-values.

    Same contract as `blueberry.benjamini_hochberg` (pyx:40-75): returns the
    q-value of each point, q[i] = max(q[i-1], min(p[i] * n / (i+1), 1)), as a
    float64 array shaped like `p_values`."""
    p = numpy.ascontiguousarray(numpy.asarray(p_values).astype("float64"))   # pyx:60
    if p.ndim != 1:
        raise ValueError("benjamini_hochberg: p_values must be one-dimensional")
    q = numpy.zeros_like(p)
    _lib.check(_lib.load().bb_benjamini_hochberg(_lib.as_f64_ptr(p), p.shape[0], int(n),
                                                 _lib.as_f64_ptr(q), int(device)),
               "bb_benjamini_hochberg")
    return q


def downsample(yp1, yp5, yp5i, device=0):
    """5x5 max-pool of `yp1` into `yp5i`, in place on top of `yp5i`'s contents,
    for the first n5-1 rows/columns (n5 = yp5.shape[0]), returning a copy of it
    -- `blueberry.downsample(yp1, yp5, yp5i)` (pyx:93-104; `yp5` only gives n5)."""
    f32p = ctypes.POINTER(ctypes.c_float)
    a = numpy.ascontiguousarray(yp1, dtype=numpy.float32)
    n5 = int(numpy.asarray(yp5).shape[0])
    out = yp5i if (isinstance(yp5i, numpy.ndarray) and yp5i.dtype == numpy.float32
                   and yp5i.flags.c_contiguous) else numpy.ascontiguousarray(yp5i, numpy.float32)
    if a.ndim != 2 or a.shape[0] != a.shape[1] or out.shape != (n5, n5):
        raise ValueError("downsample: yp1 must be square and yp5i shaped like yp5")
    _lib.check(_lib.load().bb_downsample(a.ctypes.data_as(f32p), a.shape[0],
                                         out.ctypes.data_as(f32p), n5, int(device)),
               "bb_downsample")
    return numpy.array(out)
