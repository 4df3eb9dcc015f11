"""Constants shared by the package.

Mirrors the module constants of the reference (`blueberry/utils.py:23-28`);
`count_band_regions` reads the two Fit-Hi-C cutoffs exactly as the Cython
function does (`blueberry/blueberry.pyx:82`).
"""

Q_LOWER_BOUND = 0.01
Q_UPPER_BOUND = 0.50
HIGH_FITHIC_CUTOFF = 10000000
LOW_FITHIC_CUTOFF = 25000
