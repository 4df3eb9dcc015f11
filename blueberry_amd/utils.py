"""Constants shared by the package.

Mirrors the module constants of the reference (`blueberry/utils.py:23-28`);
`count_band_regions` reads the two Fit-Hi-C cutoffs exactly as the Cython
function does (`blueberry/blueberry.pyx:82`).
"""

Q_LOWER_BOUND = 0.01
Q_UPPER_BOUND = 0.50
HIGH_FITHIC_CUTOFF = 10000000
LOW_FITHIC_CUTOFF = 25000

# hg19 chromosome lengths in bp, chr1..chr22, chrX, chrY: the genome the reference's
# data paths name (`blueberry/datatypes.pyx:25-29`, Rao et al. 2014 maps).  Their sum
# at 10 kb per bin is the 309,568 bins of BASELINE config 5.
HG19_CHROM_SIZES = (
    249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663, 146364022,
    141213431, 135534747, 135006516, 133851895, 115169878, 107349540, 102531392, 90354753,
    81195210, 78077248, 59128983, 63025520, 48129895, 51304566, 155270560, 59373566)


def genome_boundaries(n_bins=None, resolution=10000, sizes=HG19_CHROM_SIZES):
    """Bin offsets [0, b_1, ..., n] of the chromosomes laid end to end at `resolution` bp
    per bin (cumulative length rounded up once, so the blocks sum to the genome's bin count:
    309,568 at 10 kb, 61,914 at 50 kb).  `n_bins`: rescale the same proportions to that many
    bins (reduced-size tests and CPU samples of the whole-genome workload)."""
    import numpy
    cum = numpy.concatenate([[0], numpy.cumsum(numpy.asarray(sizes, dtype=numpy.int64))])
    b = -(-cum // int(resolution))                       # ceil
    if n_bins is not None:
        b = numpy.rint(b * (float(n_bins) / float(b[-1]))).astype(numpy.int64)
        b[-1] = int(n_bins)
    return b
