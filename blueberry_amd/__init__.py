"""blueberry_amd -- MI355X-native replacement for the O(N^2) bin-pair hot path
of jmschrei/blueberry.

Drop-in names (same meaning as in `blueberry.*`, reference
`blueberry/__init__.py:38-43` star-exports):
    count_band_regions            blueberry/blueberry.pyx:77-91
    ContactMap                    blueberry/datatypes.pyx:31-272
    FithicContactMap              blueberry/datatypes.pyx:274-388
    benjamini_hochberg            blueberry/blueberry.pyx:40-75
    downsample                    blueberry/blueberry.pyx:93-104
    Q_LOWER_BOUND, Q_UPPER_BOUND, HIGH_FITHIC_CUTOFF, LOW_FITHIC_CUTOFF
                                  blueberry/utils.py:23-26
Net-new (the reference has no solver; docs/SPEC.md):
    StructureSolver               contact matrix -> 3D coordinates

All compute runs in libblueberry_hip.so (hand-written HIP for gfx950) behind
the C-ABI of include/blueberry_hip.h.  Importing this package needs neither
the library nor a GPU; the first compute call does, and raises if either is
missing -- there is no CPU fallback.
"""
from .utils import (HIGH_FITHIC_CUTOFF, LOW_FITHIC_CUTOFF, Q_LOWER_BOUND,  # noqa: F401
                    Q_UPPER_BOUND)
from .band import count_band_regions  # noqa: F401
from .datatypes import ContactMap, EigenNoConvergence, FithicContactMap  # noqa: F401
from .solver import HipEngine, RankDeficient, StructureSolver  # noqa: F401
from .stats import benjamini_hochberg, downsample  # noqa: F401

__version__ = "0.1.0"
