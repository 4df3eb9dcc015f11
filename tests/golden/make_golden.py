#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run once, in the build container (needs /root/reference, Cython, gcc):

    python tests/golden/make_golden.py

What it does: copies `blueberry.pyx` and `datatypes.pyx` from the read-only
reference checkout into a throw-away temp directory (never into this repo),
puts a two-constant stub `utils.py` beside them (the reference's own
`utils.py` is Python 2 and does not import; the two values are those at
reference `blueberry/utils.py:25-26`), compiles them with pyximport, calls the
real Cython functions on seeded synthetic inputs and stores inputs + outputs
as small .npz fixtures.  The fixtures are data only; the GPU box and the test
suite never see the reference.

Functions captured (reference file:line):
  count_band_regions      blueberry/blueberry.pyx:77-91
  benjamini_hochberg      blueberry/blueberry.pyx:40-75
  downsample              blueberry/blueberry.pyx:93-104
  ContactMap.__init__     blueberry/datatypes.pyx:88-120
  ContactMap.normalize    blueberry/datatypes.pyx:143-171
  ContactMap.filter       blueberry/datatypes.pyx:122-141
  FithicContactMap        blueberry/datatypes.pyx:274-388 (__init__, contacts, to_matrix)
  FithicContactMap.decimate  blueberry/datatypes.pyx:317-339 (midpoints that are multiples of
                          the target resolution: there Python 2's integer division, which the
                          line was written for, and Python 3's true division give the same bins)
"""
import os
import shutil
import sys
import tempfile

import numpy

REF = "/root/reference/blueberry"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference(tmp):
    import pyximport
    for f in ("blueberry.pyx", "datatypes.pyx"):
        shutil.copy(os.path.join(REF, f), tmp)
    with open(os.path.join(tmp, "utils.py"), "w") as fh:
        fh.write("HIGH_FITHIC_CUTOFF = 10000000\nLOW_FITHIC_CUTOFF = 25000\n"
                 "Q_LOWER_BOUND = 0.01\n")
    open(os.path.join(tmp, "__init__.py"), "w").close()
    pkg = os.path.basename(tmp)
    sys.path.insert(0, os.path.dirname(tmp))   # `from .utils import *` needs a package
    sys.path.insert(0, tmp)                    # datatypes.pyx does `from blueberry import *`
    pyximport.install(language_level=2, build_dir=os.path.join(tmp, "_bld"),
                      setup_args={"include_dirs": numpy.get_include()})
    bb = __import__(pkg + ".blueberry", fromlist=["x"])
    sys.modules["blueberry"] = bb
    dt = __import__(pkg + ".datatypes", fromlist=["x"])
    return bb, dt


def band_cases():
    rng = numpy.random.default_rng(0)
    cases = {}
    for res, tag in ((50000, "50kb"), (10000, "10kb"), (5000, "5kb")):
        for n in (1, 2, 1000, 4096):
            cases["uniform_%s_n%d" % (tag, n)] = numpy.arange(n) * float(res) + res / 2.0
    for n in (257, 1000, 5000):
        gaps = rng.integers(1, 40, size=n) * 5000.0
        gaps[rng.random(n) < 0.02] += 3.0e6         # centromere-like holes
        cases["gappy_n%d" % n] = numpy.cumsum(gaps) + 2500.0
    # both bounds are inclusive (pyx:88)
    cases["edge_exact_low"] = numpy.array([0.0, 25000.0, 50000.0])
    cases["edge_exact_high"] = numpy.array([0.0, 10000000.0, 10000001.0, 9999999.0])
    cases["edge_just_outside"] = numpy.array([0.0, 24999.0, 10000001.0, 10024999.0])
    # unsorted: only (i, j<i) with regions[i]-regions[j] in band count
    cases["unsorted"] = rng.permutation(numpy.arange(300) * 40000.0)
    cases["descending"] = numpy.arange(200)[::-1] * 50000.0
    cases["fractional"] = numpy.sort(rng.random(500) * 3.0e7)
    cases["empty"] = numpy.zeros(0)
    return cases


def make_band(bb):
    out = {}
    for name, r in band_cases().items():
        r = numpy.ascontiguousarray(r, dtype=numpy.float64)
        out["in_" + name] = r
        out["out_" + name] = numpy.int64(bb.count_band_regions(r))
    numpy.savez_compressed(os.path.join(OUT, "band_count.npz"), **out)
    print("band_count:", {k[4:]: int(v) for k, v in out.items() if k.startswith("out_")})


def make_bh_downsample(bb):
    rng = numpy.random.default_rng(2)
    out = {}
    for k, (d, n) in enumerate(((4, 10), (100, 1000), (1000, 179900), (50, 50))):
        p = numpy.sort(rng.random(d) ** 3)
        out["bh_p_%d" % k] = p
        out["bh_n_%d" % k] = numpy.int64(n)
        out["bh_q_%d" % k] = bb.benjamini_hochberg(p, n)
    # NaN p-values (round 2; own generator, the cases above keep their bits): Cython's
    # min(q, 1) / max(q, prev) are `1 < q ? 1 : q` and `prev > q ? prev : q`
    # (pyx:67-68), so a NaN comes out as NaN AND restarts the running maximum
    rng_nan = numpy.random.default_rng(12)
    for k, (d, n, where) in enumerate(((6, 10, (2,)), (40, 100, (0, 17, 18)), (2500, 9000, (1023, 1024, 2047)),
                                       (5, 5, (4,))), start=4):
        p = numpy.sort(rng_nan.random(d) ** 3)
        p[list(where)] = numpy.nan
        out["bh_p_%d" % k] = p
        out["bh_n_%d" % k] = numpy.int64(n)
        with numpy.errstate(all="ignore"):
            out["bh_q_%d" % k] = bb.benjamini_hochberg(p, n)
    for k, n5 in enumerate((2, 7, 20)):
        yp1 = rng.random((n5 * 5, n5 * 5)).astype(numpy.float32)
        yp5i = (rng.random((n5, n5)) * 0.9).astype(numpy.float32)
        out["ds_yp1_%d" % k] = yp1
        out["ds_yp5i_%d" % k] = yp5i.copy()
        out["ds_out_%d" % k] = bb.downsample(yp1, numpy.zeros_like(yp5i), yp5i.copy())
    numpy.savez_compressed(os.path.join(OUT, "bh_downsample.npz"), **out)
    print("bh/downsample cases written")


def make_contactmap(dt, tmp):
    """Drive the real ContactMap ctor + normalize on synthetic Rao-format files."""
    rng = numpy.random.default_rng(3)
    data_dir = os.path.join(tmp, "data")
    os.makedirs(data_dir)
    dt.RAW_DIR = os.path.join(data_dir, "{0}_chr{1}_{2}kb.RAWobserved")
    dt.KR_NORM = os.path.join(data_dir, "{0}_chr{1}_{2}kb.KRnorm")
    dt.KR_EXP = os.path.join(data_dir, "{0}_chr{1}_{2}kb.KRexpected")
    out = {}
    for k, (n_bins, res) in enumerate(((6, 50000), (64, 10000), (300, 5000))):
        # sparse upper-triangle triples (pos_i <= pos_j), some duplicates (last wins)
        nnz = max(8, n_bins * 6)
        bi = rng.integers(0, n_bins, size=nnz)
        bj = numpy.minimum(n_bins - 1, bi + rng.geometric(0.15, size=nnz) - 1)
        counts = rng.integers(1, 500, size=nnz).astype(numpy.float64)
        triples = numpy.stack([bi * float(res), bj * float(res), counts], axis=1)
        kr = 0.5 + rng.random(n_bins)
        kr[rng.random(n_bins) < 0.08] = numpy.nan     # unmappable bins, as in Rao KRnorm files
        krexp = 50.0 / (1.0 + numpy.arange(n_bins)) + 0.1
        tag = (("cell%d" % k), k + 1, res / 1000)     # py3: resolution/1000 is a float
        numpy.savetxt(dt.RAW_DIR.format(*tag), triples, delimiter="\t", fmt="%.1f")
        numpy.savetxt(dt.KR_NORM.format(*tag), kr)
        numpy.savetxt(dt.KR_EXP.format(*tag), krexp)
        cm = dt.ContactMap("cell%d" % k, k + 1, res)
        assert cm.n_bins == n_bins
        out["cm%d_triples" % k] = triples
        out["cm%d_resolution" % k] = numpy.int64(res)
        out["cm%d_krnorm" % k] = numpy.loadtxt(dt.KR_NORM.format(*tag))
        out["cm%d_krexp" % k] = numpy.loadtxt(dt.KR_EXP.format(*tag))
        out["cm%d_matrix_raw" % k] = cm.matrix.copy()
        out["cm%d_regions" % k] = cm.regions.copy()
        with numpy.errstate(all="ignore"):
            cm.normalize()
        out["cm%d_matrix_norm" % k] = cm.matrix.copy()
        # A4 (round 2): the real ContactMap.filter (pyx:122-141) on the raw and on the
        # normalised map, threshold 0 and the median marginal; n_bins / regions are
        # stored as the reference leaves them (stale)
        for tagf, do_norm, med in (("raw_t0", False, False), ("raw_tmed", False, True),
                                   ("norm_t0", True, False), ("norm_tmed", True, True)):
            cf = dt.ContactMap("cell%d" % k, k + 1, res)
            if do_norm:
                with numpy.errstate(all="ignore"):
                    cf.normalize()
            thr = float(numpy.median(cf.matrix.sum(axis=0))) if med else 0.0
            cf.filter(thr)
            out["cm%d_filter_%s_thr" % (k, tagf)] = numpy.float64(thr)
            out["cm%d_filter_%s_matrix" % (k, tagf)] = numpy.ascontiguousarray(cf.matrix)
            out["cm%d_filter_%s_n_bins" % (k, tagf)] = numpy.int64(cf.n_bins)
            out["cm%d_filter_%s_regions" % (k, tagf)] = cf.regions.copy()
    # A zero KR entry: Cython's default cdivision=False turns the C division
    # at datatypes.pyx:168 into a checked one, so the reference raises.
    numpy.savetxt(dt.KR_NORM.format("cell0", 1, 50.0), numpy.array([1.0, 0.0, 1.0, 1.0, 1.0, 1.0]))
    cm = dt.ContactMap("cell0", 1, 50000)
    try:
        cm.normalize()
        raised = False
    except ZeroDivisionError:
        raised = True
    out["cm_zero_kr_raises_zerodivision"] = numpy.bool_(raised)
    numpy.savez_compressed(os.path.join(OUT, "contactmap.npz"), **out)
    print("contactmap cases written")


def make_fithic(dt, tmp):
    """FithicContactMap (datatypes.pyx:274-388) on a synthetic Fit-Hi-C output
    file (format: fithic.py:411 header, 7 tab-separated columns, gzip)."""
    import gzip
    rng = numpy.random.default_rng(4)
    data_dir = os.path.join(tmp, "fithic")
    os.makedirs(data_dir)
    dt.DATA_DIR = os.path.join(data_dir, "{0}.chr{1}.res{2}.significances.txt.gz")
    dt.KR_NORM = os.path.join(data_dir, "{0}_chr{1}_{2}kb.KRnorm")
    n_bins, res, n = 40, 5000, 300
    b1 = rng.integers(0, n_bins, n)
    b2 = numpy.minimum(n_bins - 1, b1 + rng.integers(1, 12, n))
    key = numpy.unique(b1 * n_bins + b2)
    b1, b2 = key // n_bins, key % n_bins
    mid1, mid2 = b1 * res + res // 2, b2 * res + res // 2
    cc = rng.integers(1, 90, b1.size)
    pv = rng.random(b1.size) ** 3
    qv = numpy.minimum(1.0, pv * 4)
    path = dt.DATA_DIR.format("cellF", 7, res)
    with gzip.open(path, "wt") as fh:
        fh.write("chr1\tfragmentMid1\tchr2\tfragmentMid2\tcontactCount\tp-value\tq-value\n")
        for k in range(b1.size):
            fh.write("7\t%d\t7\t%d\t%d\t%.10e\t%.10e\n" % (mid1[k], mid2[k], cc[k], pv[k], qv[k]))
    numpy.savetxt(dt.KR_NORM.format("cellF", 7, res / 1000), numpy.ones(n_bins))
    fm = dt.FithicContactMap("cellF", 7, res)
    out = {"fh_map": fm.map.copy(), "fh_regions": fm.regions.copy(), "fh_contacts": fm.contacts(),
           "fh_resolution": numpy.int64(res), "fh_n_bins": numpy.int64(n_bins)}
    for stat in ("count", "p", "q"):
        out["fh_matrix_" + stat] = fm.to_matrix(stat)
    numpy.savez_compressed(os.path.join(OUT, "fithic_map.npz"), **out)
    print("fithic map cases written", fm.map.shape, out["fh_contacts"].shape)


def make_fithic_decimate(dt, tmp):
    """The real FithicContactMap.decimate (datatypes.pyx:317-339).  Its rounding line
    `(mid.astype('int') + r) / r * r - r/2` means floor division under Python 2 and true
    division here; for midpoints that are exact multiples of r both give mid + r/2, so on
    such a map what this captures IS the reference's aggregation: counts summed, p-values
    multiplied in file order, q-values minimised, one row per (mid1, mid2) in the order of
    first occurrence (dict insertion order)."""
    import gzip
    rng = numpy.random.default_rng(8)
    data_dir = os.path.join(tmp, "fithic_dec")
    os.makedirs(data_dir)
    dt.DATA_DIR = os.path.join(data_dir, "{0}.chr{1}.res{2}.significances.txt.gz")
    r, n_keys, n_rows = 5000, 40, 130
    k1 = rng.integers(0, 60, n_keys)
    k2 = k1 + rng.integers(1, 30, n_keys)
    pick = numpy.concatenate([numpy.arange(n_keys), rng.integers(0, n_keys, n_rows - n_keys)])
    pick = rng.permutation(pick)
    mid1, mid2 = k1[pick] * r, k2[pick] * r                  # exact multiples of r
    cc = rng.integers(1, 50, n_rows)
    pv = rng.random(n_rows) ** 2
    qv = numpy.minimum(1.0, pv * 3)
    path = dt.DATA_DIR.format("cellG", 9, 1000)
    with gzip.open(path, "wt") as fh:
        fh.write("chr1\tfragmentMid1\tchr2\tfragmentMid2\tcontactCount\tp-value\tq-value\n")
        for k in range(n_rows):
            fh.write("9\t%d\t9\t%d\t%d\t%.10e\t%.10e\n" % (mid1[k], mid2[k], cc[k], pv[k], qv[k]))
    fm = dt.FithicContactMap("cellG", 9, 1000)
    out = {"dec_in_map": fm.map.copy(), "dec_in_resolution": numpy.int64(1000),
           "dec_resolution": numpy.int64(r)}
    assert fm.decimate(r) is None
    out["dec_map"], out["dec_regions"] = fm.map.copy(), fm.regions.copy()
    out["dec_resolution_attr"] = numpy.int64(fm.resolution)
    n_distinct = numpy.unique(numpy.stack([mid1, mid2], axis=1), axis=0).shape[0]
    assert out["dec_map"].shape[0] == n_distinct < n_rows == out["dec_in_map"].shape[0]
    numpy.savez_compressed(os.path.join(OUT, "fithic_decimate.npz"), **out)
    print("fithic decimate case written", out["dec_in_map"].shape, "->", out["dec_map"].shape)


def main():
    tmp = tempfile.mkdtemp(prefix="bbref_")
    try:
        bb, dt = import_reference(tmp)
        make_band(bb)
        make_bh_downsample(bb)
        make_contactmap(dt, tmp)
        make_fithic(dt, tmp)
        make_fithic_decimate(dt, tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        shutil.rmtree(os.path.expanduser("~/.pyxbld"), ignore_errors=True)


if __name__ == "__main__":
    main()
