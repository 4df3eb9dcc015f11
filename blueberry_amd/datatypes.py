"""ContactMap -- the solver's input datatype.

Mirrors the public surface of the reference's Cython class
(`blueberry/datatypes.pyx:31-272`): same constructor arguments, public
attributes (pyx:78-86), in-place / return-None methods and the
`(n_bins+1, n_bins+1)` float64 matrix.  The two Cython loops of the class --
the sparse-triple scatter (pyx:110-116) and KR + observed/expected
normalisation (pyx:166-169, nan_to_num :171) -- run on the GPU through
libblueberry_hip.so and are bit-exact against golden vectors captured from
the real reference (tests/golden/contactmap.npz).  Everything the reference
does with numpy / scipy / pandas stays numpy / scipy / pandas here.

Deviations from the reference, all deliberate:
  * `from_arrays` works: the reference's calls the file-loading constructor
    and has no `return` (pyx:264-272), so it needs the lab's NFS files and
    yields None.  Here it builds the map from the arrays alone, and takes
    optional KRnorm / KRexpected so that `normalize()` is usable.
  * path templates use `resolution // 1000` (the Python 2 meaning of the
    reference's `resolution/1000`, pyx:90-95).
  * `filter` also updates `n_bins` (= rows kept) and `regions` (the reference
    leaves them stale, pyx:140-141) and drops the KR vectors, so a later
    `normalize()` raises instead of indexing out of bounds --
    `keep_stale=True` restores the old behaviour.  The matrix it leaves is bit
    for bit the reference's (goldens `cm*_filter_*`).
  * `matrix` is a lazily fetched attribute, read-only while the matrix lives in HBM
    (class docstring); `to_host()`, `host_matrix()`, `marginals()`, `from_triples()`
    are additions.
  * `normalize` raises ZeroDivisionError up front when a divisor would be 0
    (the reference raises from inside the loop, after modifying part of the
    matrix, because Cython checks float division; golden flag
    `cm_zero_kr_raises_zerodivision`).
"""
import numpy

from . import _lib

RAO = "/net/noble/vol1/data/hic-datasets/"
RAW_DIR = RAO + ("data/Rao-Cell2014/rawFromGEO/{0}/{2}kb_resolution_intrachromosomal/chr{1}/"
                 "MAPQGE30/chr{1}_{2}kb.RAWobserved")
KR_NORM = RAO + ("data/Rao-Cell2014/rawFromGEO/{0}/{2}kb_resolution_intrachromosomal/chr{1}/"
                 "MAPQGE30/chr{1}_{2}kb.KRnorm")
KR_EXP = RAO + ("data/Rao-Cell2014/rawFromGEO/{0}/{2}kb_resolution_intrachromosomal/chr{1}/"
                "MAPQGE30/chr{1}_{2}kb.KRexpected")


def _pick_device(device):
    """An explicit index wins; else this rank's own GPU inside a torch.distributed job
    (LOCAL_RANK, the rule of StructureSolver._pick_device and band.pick_device), else 0.
    torch is only looked at if the process has imported it already."""
    if device is not None:
        return int(device)
    import os
    import sys
    td = sys.modules.get("torch.distributed")
    if td is not None and td.is_available() and td.is_initialized() and td.get_world_size() > 1:
        return int(os.environ.get("LOCAL_RANK", "0"))
    return 0


def scatter_triples(triples, resolution, n_bins, device=0):
    """(n,3) [pos_i, pos_j, count] rows -> dense symmetric (n_bins+1)^2 host matrix.

    GPU form of the loop at `blueberry/datatypes.pyx:110-116`: bin =
    int(pos / resolution), both [j,k] and [k,j] set, later rows win."""
    return _DeviceMatrix.from_triples(triples, resolution, n_bins, device).to_host()


def _assign_last_wins(matrix, rows, cols, vals):
    """matrix[rows[k], cols[k]] = vals[k] for k in order -- of several entries for one cell
    the LAST stays, as in the reference's Python loops (`blueberry/datatypes.pyx:268-271,
    376-386`) -- without a Python loop over the entries: numpy leaves the outcome of a fancy
    assignment with repeated indices open, so the last entry of every cell is picked first."""
    if rows.shape[0] == 0:
        return
    flat = rows.astype(numpy.int64) * matrix.shape[1] + cols.astype(numpy.int64)
    # unique() on the reversed keys returns, per cell, its first index there = its last here
    _, first_rev = numpy.unique(flat[::-1], return_index=True)
    last = flat.shape[0] - 1 - first_rev
    matrix.reshape(-1)[flat[last]] = numpy.asarray(vals)[last]


def _nan_to_num(a):
    """`numpy.nan_to_num` of a float64 view of `a` (pyx:102) -- without the copy and the
    three passes when every value is finite already (10 M triples: 36 ms instead of 0.7 s).
    Host paths only: the device scatter applies nan_to_num to the values as it reads them."""
    a = numpy.asarray(a, dtype=numpy.float64)
    return a if numpy.isfinite(a).all() else numpy.nan_to_num(a)


class _DeviceMatrix(object):
    """Owner of one bb_cm handle: the (d, d) float64 matrix resident in HBM."""

    def __init__(self, d, device):
        self._lib = _lib.load()
        self._h = _lib.c_void_p()
        self.device = int(device)
        _lib.check(self._lib.bb_cm_create(self._h, int(d), self.device), "bb_cm_create")

    @classmethod
    def from_triples(cls, triples, resolution, n_bins, device, want_regions=False):
        """Scatter (n, 3) rows [pos_i, pos_j, count].  With want_regions also returns
        `numpy.union1d(pos_i, pos_j)` (pyx:120): when every position is exactly
        bin * resolution -- Rao's format -- it is the bins the scatter kernel saw times the
        resolution (no sort of 2n doubles on the host); otherwise numpy's own union1d."""
        import ctypes
        # numpy.nan_to_num (pyx:102) is applied by the scatter kernels to the values they read:
        # no pass over the triples on the host
        t = numpy.asarray(triples, dtype=numpy.float64)
        if t.ndim != 2 or t.shape[1] != 3:
            raise ValueError("triples must have shape (n, 3)")
        # C-ordered rows are read in place; the reference's own layout -- pandas hands it
        # an F-ordered array and its pointer arithmetic reads that column-major
        # (pyx:111-113) -- is read in place too; anything else is copied once
        if t.flags.c_contiguous:
            buf, row_major = t, 1
        elif t.flags.f_contiguous:
            buf, row_major = t.T, 0
        else:
            buf, row_major = numpy.ascontiguousarray(t), 1
        d = int(n_bins) + 1
        self = cls(d, device)                                               # zeros, pyx:99
        present = numpy.zeros(d, dtype=numpy.uint8)
        on_grid = _lib.c_i32(0)
        _lib.check(self._lib.bb_cm_scatter_ex(
            self._h, _lib.as_f64_ptr(buf), t.shape[0], int(resolution), row_major,
            present.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.byref(on_grid)),
            "bb_cm_scatter")
        if not want_regions:
            return self
        if on_grid.value:
            regions = numpy.flatnonzero(present).astype(numpy.float64) * float(int(resolution))
        else:
            regions = numpy.union1d(_nan_to_num(t[:, 0]), _nan_to_num(t[:, 1]))
        return self, regions

    @classmethod
    def from_host(cls, matrix, device):
        m = numpy.ascontiguousarray(matrix, dtype=numpy.float64)
        self = cls(m.shape[0], device)
        _lib.check(self._lib.bb_cm_upload(self._h, _lib.as_f64_ptr(m), m.shape[1]), "bb_cm_upload")
        return self

    @property
    def d(self):
        n = _lib.c_i64()
        _lib.check(self._lib.bb_cm_dim(self._h, n), "bb_cm_dim")
        return int(n.value)

    def to_host(self):
        d = self.d
        out = numpy.empty((d, d), dtype=numpy.float64)
        if d:
            _lib.check(self._lib.bb_cm_download(self._h, _lib.as_f64_ptr(out), d), "bb_cm_download")
        return out

    def close(self):
        if self._h:
            self._lib.bb_cm_destroy(self._h)
            self._h = _lib.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EigenNoConvergence(RuntimeError):
    """`ContactMap.eigenvector` used up `max_matvecs` before its residual reached the
    tolerance (the counterpart of scipy's ArpackNoConvergence, which the reference's
    `eigsh` call raises, `blueberry/datatypes.pyx:234`).  `.eigenvalue`, `.eigenvector`:
    the pair found so far."""

    def __init__(self, msg, eigenvalue, eigenvector):
        RuntimeError.__init__(self, msg)
        self.eigenvalue, self.eigenvector = eigenvalue, eigenvector


class ContactMap(object):
    """This is a contact map for Hi-C datasets.

    Same parameters and attributes as the reference class
    (`blueberry/datatypes.pyx:31-76`).

    Parameters
    ----------
    celltype : str
    chromosome : int
    resolution : int

    Attributes
    ----------
    resolution, chromosome, celltype, filename, n_bins
    matrix : numpy.ndarray, shape=(n_bins+1, n_bins+1), float64
    regions : numpy.ndarray -- the midpoints found in this map

    Where the matrix lives.  The reference keeps one host matrix alive across
    `__init__` -> `normalize()` -> `filter()` -> consumer (pyx:97-120, :161-171,
    :140-141).  Here that one matrix lives in HBM (`bb_cm_*`, include/blueberry_hip.h):
    the constructor scatters the triples on the device, `normalize()` and `filter()`
    work on the resident matrix in place, and `StructureSolver.fit(contact_map)` packs
    it device to device -- no (n_bins+1)^2 host transfer anywhere on that path.
    `matrix` is fetched lazily and is a READ-ONLY copy while the matrix lives in HBM: a read
    does not cost the residency.  `to_host()` returns a private writable copy and leaves the
    device copy authoritative; `host_matrix()` / `cm.matrix = m` make a host array the
    authoritative one.  `device=None` means this rank's own GPU inside a torch.distributed
    job (LOCAL_RANK, as `StructureSolver` picks it), else device 0.
    """

    def __init__(self, celltype, chromosome, resolution=1000, device=None):
        import pandas
        self.resolution = int(resolution)
        self.chromosome = chromosome
        self.celltype = celltype
        self.device = _pick_device(device)
        kb = self.resolution // 1000
        self.filename = RAW_DIR.format(celltype, chromosome, kb)
        self._KRnorm = numpy.atleast_1d(numpy.loadtxt(KR_NORM.format(celltype, chromosome, kb)))
        self._KRexpected = numpy.atleast_1d(numpy.loadtxt(KR_EXP.format(celltype, chromosome, kb)))
        self.n_bins = int(self._KRnorm.shape[0])
        data = pandas.read_csv(self.filename, delimiter="\t", engine="c", dtype="float64",
                               header=None).values               # nan_to_num: on the device
        self._host = self._view = None
        self._dev, self.regions = _DeviceMatrix.from_triples(data, self.resolution, self.n_bins,
                                                             self.device, want_regions=True)
        self.regions.sort()

    # -- where the matrix lives ------------------------------------------
    @property
    def matrix(self):
        """The (n_bins+1)^2 float64 matrix.  While the matrix lives in HBM this is a
        READ-ONLY host copy (fetched once, dropped when a device operation changes the
        matrix): looking at `cm.matrix.shape`, `.sum()` or a cell costs one download, not
        the residency.  Round 2 handed out a writable array and, because the caller might
        write into it, gave the device copy up on every read -- 5 GB down and 5 GB up again
        around a `cm.matrix[0, 0]` at chr1@10kb.  To CHANGE the matrix assign a new one
        (`cm.matrix = m`) or ask for the writable host copy (`cm.host_matrix()`); both
        make the host array the authoritative one, as the reference's attribute is
        (`blueberry/datatypes.pyx:85`)."""
        if self._dev is None:
            return self._host
        if self._view is None:
            v = self._dev.to_host()
            v.flags.writeable = False
            self._view = v
        return self._view

    @matrix.setter
    def matrix(self, value):
        self._host = value
        self._view = None
        if self._dev is not None:
            self._dev.close()
            self._dev = None

    def host_matrix(self):
        """The matrix as a writable host array that is from now on the authoritative copy
        (the resident one is given up; the next device operation uploads the array again)."""
        if self._dev is not None:
            self._host = self._dev.to_host()
            self._dev.close()
            self._dev = None
        self._view = None
        return self._host

    def to_host(self):
        """A private copy of the matrix; the resident one stays authoritative."""
        if self._dev is not None:
            return self._dev.to_host()
        return numpy.array(self._host, dtype=numpy.float64)

    @property
    def is_resident(self):
        """True while the authoritative matrix is the one in HBM."""
        return self._dev is not None

    def _resident(self):
        """The device handle of the authoritative matrix (uploading a host one)."""
        if self._dev is None:
            m = numpy.ascontiguousarray(self._host, dtype=numpy.float64)
            if m.ndim != 2 or m.shape[0] != m.shape[1]:
                raise ValueError("matrix must be square")
            if m.shape[0] == 0:
                raise ValueError("the contact map is empty (every bin was filtered out)")
            self._dev = _DeviceMatrix.from_host(m, self.device)
            self._host = None
        self._view = None      # the caller is about to work on (usually: change) the resident matrix
        return self._dev

    # a ContactMap pickles / deep-copies as its host matrix: the device handle is a pointer
    # into this process's HIP context and means nothing anywhere else
    def __getstate__(self):
        state = dict(self.__dict__)
        state["_host"] = self.to_host()
        state["_dev"] = state["_view"] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)

    @property
    def shape(self):
        d = self._dev.d if self._dev is not None else self._host.shape[0]
        return (d, d)

    # ------------------------------------------------------------------
    @classmethod
    def from_arrays(cls, celltype, chromosome, resolution, contacts, n_bins=None, KRnorm=None,
                    KRexpected=None, symmetric=True, device=None):
        """Create a contact map from numpy arrays.  No files required.

        contacts : (n, 3) array of (mid1, mid2, statistic), mid = bin midpoint,
        placed at bin int((mid - resolution/2) / resolution) as at
        `blueberry/datatypes.pyx:268-271`.  `symmetric` also sets the mirrored
        cell (the file constructor's matrices are symmetric; the reference's
        from_arrays would have set one triangle only)."""
        self = cls.__new__(cls)
        self.resolution = int(resolution)
        self.chromosome = chromosome
        self.celltype = celltype
        self.device = _pick_device(device)
        self.filename = ""
        contacts = numpy.asarray(contacts, dtype=numpy.float64)
        if contacts.ndim != 2 or contacts.shape[1] != 3:
            raise ValueError("contacts must have shape (n, 3)")
        b1 = ((contacts[:, 0] - self.resolution / 2.0) / self.resolution).astype(numpy.int64)
        b2 = ((contacts[:, 1] - self.resolution / 2.0) / self.resolution).astype(numpy.int64)
        if n_bins is None:
            n_bins = (KRnorm.shape[0] if KRnorm is not None
                      else int(max(b1.max(initial=-1), b2.max(initial=-1)) + 1))
        self.n_bins = int(n_bins)
        if contacts.shape[0] and (min(b1.min(), b2.min()) < 0 or
                                  max(b1.max(), b2.max()) > self.n_bins):
            raise ValueError("a contact falls outside [0, n_bins]")
        d = self.n_bins + 1
        m = numpy.zeros((d, d), dtype=numpy.float64)
        if symmetric:
            # the reference's order of stores: [b1, b2] then [b2, b1], entry by entry
            rows = numpy.stack([b1, b2], axis=1).reshape(-1)
            cols = numpy.stack([b2, b1], axis=1).reshape(-1)
            _assign_last_wins(m, rows, cols, numpy.repeat(contacts[:, 2], 2))
        else:
            _assign_last_wins(m, b1, b2, contacts[:, 2])     # later rows win, as in the reference
        self._host, self._dev, self._view = m, None, None
        self.regions = numpy.union1d(contacts[:, 0], contacts[:, 1])
        self._KRnorm = None if KRnorm is None else numpy.asarray(KRnorm, dtype=numpy.float64)
        self._KRexpected = (None if KRexpected is None
                            else numpy.asarray(KRexpected, dtype=numpy.float64))
        return self

    @classmethod
    def from_matrix(cls, matrix, resolution=1000, celltype="", chromosome=0, KRnorm=None,
                    KRexpected=None, device=None):
        """Wrap an existing dense symmetric ((n_bins+1)^2) float64 matrix."""
        self = cls.__new__(cls)
        m = numpy.ascontiguousarray(matrix, dtype=numpy.float64)
        if m.ndim != 2 or m.shape[0] != m.shape[1] or m.shape[0] < 1:
            raise ValueError("matrix must be square")
        self.resolution, self.chromosome, self.celltype = int(resolution), chromosome, celltype
        self.device, self.filename = _pick_device(device), ""
        self._host, self._dev, self._view = m, None, None
        self.n_bins = m.shape[0] - 1
        self.regions = numpy.arange(self.n_bins, dtype=numpy.float64) * self.resolution
        self._KRnorm = None if KRnorm is None else numpy.asarray(KRnorm, dtype=numpy.float64)
        self._KRexpected = (None if KRexpected is None
                            else numpy.asarray(KRexpected, dtype=numpy.float64))
        return self

    @classmethod
    def from_triples(cls, triples, resolution, n_bins, KRnorm=None, KRexpected=None, celltype="",
                     chromosome=0, device=None):
        """The file constructor without the files: Rao-format (n, 3) rows
        [pos_i, pos_j, count] (what `__init__` reads, pyx:100-102), scattered on the
        device; the matrix is never built on the host."""
        self = cls.__new__(cls)
        self.resolution, self.chromosome, self.celltype = int(resolution), chromosome, celltype
        self.device, self.filename = _pick_device(device), ""
        data = numpy.asarray(triples, dtype=numpy.float64)         # nan_to_num: on the device
        self.n_bins = int(n_bins)
        self._host = self._view = None
        self._dev, self.regions = _DeviceMatrix.from_triples(data, self.resolution, self.n_bins,
                                                             self.device, want_regions=True)
        self._KRnorm = None if KRnorm is None else numpy.asarray(KRnorm, dtype=numpy.float64)
        self._KRexpected = (None if KRexpected is None
                            else numpy.asarray(KRexpected, dtype=numpy.float64))
        return self

    # ------------------------------------------------------------------
    def marginals(self):
        """`matrix.sum(axis=0)` of the resident matrix (pyx:140), bit for bit."""
        if self.shape[0] == 0:
            return numpy.zeros(0)
        dev = self._resident()
        out = numpy.empty(dev.d, dtype=numpy.float64)
        _lib.check(dev._lib.bb_cm_marginals(dev._h, _lib.as_f64_ptr(out)), "bb_cm_marginals")
        return out

    def filter(self, threshold=0, keep_stale=False):
        """Remove rows and columns whose marginal count is <= threshold.

        In place, returns None (`blueberry/datatypes.pyx:122-141`); runs on the GPU on
        the resident matrix: column sums in numpy's summation order, prefix-sum
        compaction, gather."""
        if self.shape[0] == 0:
            return                     # nothing left to filter (numpy: an empty index list)
        dev = self._resident()
        d = dev.d
        keep = numpy.zeros(d, dtype=numpy.uint8)
        dn = _lib.c_i64()
        import ctypes
        _lib.check(dev._lib.bb_cm_filter(dev._h, float(threshold), dn,
                                         keep.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))),
                   "bb_cm_filter")
        keep = keep.astype(bool)
        if not keep_stale:
            # bins that survive, in the old numbering; the zero padding row
            # (index n_bins) never survives threshold >= 0
            self.n_bins = int(dn.value)
            if self.regions is not None and self.regions.shape[0]:
                bins = (self.regions / self.resolution).astype(numpy.int64)
                ok = (bins >= 0) & (bins < keep.shape[0])
                ok[ok] = keep[bins[ok]]
                self.regions = self.regions[ok]
            # KR vectors describe the unfiltered map: normalize() first, then filter()
            self._KRnorm = None
            self._KRexpected = None

    def normalize(self):
        """KR matrix balancing and observed/expected normalisation, in place.

        m[j, j+i] /= KRnorm[j] * KRnorm[j+i] * KRexpected[i], mirrored, then
        nan_to_num (`blueberry/datatypes.pyx:143-171`); runs on the GPU on the
        resident matrix."""
        if self._KRnorm is None or self._KRexpected is None:
            raise ValueError("normalize() needs KRnorm and KRexpected")
        n = self.n_bins
        if self.shape != (n + 1, n + 1):
            raise ValueError("matrix shape does not match n_bins (was filter() used with "
                             "keep_stale=True?)")
        if self._KRnorm.shape[0] < n or self._KRexpected.shape[0] < n:
            raise ValueError("KRnorm / KRexpected shorter than n_bins")
        kr = numpy.ascontiguousarray(self._KRnorm[:n], dtype=numpy.float64)
        ke = numpy.ascontiguousarray(self._KRexpected[:n], dtype=numpy.float64)
        if n and (numpy.any(kr == 0.0) or numpy.any(ke == 0.0)):
            raise ZeroDivisionError("float division")
        dev = self._resident()
        _lib.check(dev._lib.bb_cm_normalize(dev._h, n, _lib.as_f64_ptr(kr), _lib.as_f64_ptr(ke)),
                   "bb_cm_normalize")

    def correlation(self):
        """Convert the map to a correlation map, in place (pyx:173-188:
        `numpy.corrcoef(matrix)`), on the resident matrix: rows centred, Gram matrix on
        the fp64 matrix cores, scaled by the diagonal and clipped to [-1, 1] in numpy's
        order of operations (`bb_cm_correlation`).  Equal to numpy to rounding.
        `correlation_tflops_` holds the rate of the Gram kernel afterwards."""
        if self.shape[0] == 0:
            return
        dev = self._resident()
        tf = _lib.c_dbl()
        _lib.check(dev._lib.bb_cm_correlation(dev._h, tf), "bb_cm_correlation")
        self.correlation_tflops_ = float(tf.value)

    def plot(self, arcsinh=True, **kwargs):
        """Plot the contact map onto the current palette (pyx:190-214)."""
        import matplotlib.pyplot as plt
        plt.title("{} chr{} at {}kb resolution".format(self.celltype, self.chromosome,
                                                       self.resolution // 1000), fontsize=16)
        plt.xlabel("Genomic Coordinate (kb)", fontsize=14)
        plt.ylabel("Genomic Coordinate (kb)", fontsize=14)
        plt.xticks(fontsize=14)
        plt.yticks(fontsize=14)
        m = self.to_host()
        plt.imshow(numpy.arcsinh(m) if arcsinh else m, **kwargs)

    def eigenvector(self, tol=1e-13, max_matvecs=2000):
        """First eigenvector of the matrix (pyx:216-235).

        The reference calls `scipy.sparse.linalg.eigsh(matrix, k=1)`: ARPACK's
        restarted Lanczos for the eigenpair of largest magnitude.  Here the same
        pair comes from a restarted Lanczos over the RESIDENT matrix (`bb_cm_eigenvector`:
        one HBM sweep per matrix-vector product, basis on the device).  ARPACK's sign is
        arbitrary; this one makes the largest-magnitude component positive.
        `eigenvalue_` holds the eigenvalue afterwards."""
        dev = self._resident()
        d = dev.d
        vec = numpy.empty(d, dtype=numpy.float64)
        lam, used, res = _lib.c_dbl(), _lib.c_i64(), _lib.c_dbl()
        _lib.check(dev._lib.bb_cm_eigenvector(dev._h, _lib.as_f64_ptr(vec), lam, float(tol),
                                              int(max_matvecs), used, res), "bb_cm_eigenvector")
        self.eigenvalue_, self.eigen_matvecs_, self.eigen_residual_ = (
            float(lam.value), int(used.value), float(res.value))
        if self.eigen_residual_ > max(float(tol), 2.3e-16) * abs(self.eigenvalue_):
            # scipy's eigsh raises ArpackNoConvergence here; the unconverged pair travels
            # with the exception as ARPACK's does
            raise EigenNoConvergence(
                "eigenvector: residual %.3e after %d matrix-vector products (tol %.1e x |%.6e|)"
                % (self.eigen_residual_, self.eigen_matvecs_, tol, self.eigenvalue_),
                self.eigenvalue_, vec)
        return vec


DATA_DIR = RAO + ("results/Rao-Cell2014/fixedWindowSize/fithic/afterICE/{2}/"
                  "{0}.chr{1}.spline_pass1.res{2}.significances.txt.gz")


class FithicContactMap(object):
    """A contact map which has been processed with Fit-Hi-C.

    Same surface as the reference class (`blueberry/datatypes.pyx:274-388`):
    `map` is an (n_contacts, 5) float64 array of (mid1, mid2, contactCount, p, q)
    read from columns [1, 3, 4, 5, 6] of a `.significances.txt.gz` file (format
    written at `blueberry/fithic.py:411`), `regions` the midpoints seen.  It is
    the on-disk adapter that lets real Fit-Hi-C output feed `StructureSolver`:
    `to_sparse()` gives the scipy matrix `fit()` takes without ever building the
    dense one.

    Deviations: `decimate` uses the integer arithmetic the
    Python 2 reference meant (`(int + res) // res * res - res // 2`) and emits
    rows sorted by (mid1, mid2) instead of in dict order; `to_matrix` takes an
    optional `n_bins` instead of always reading the KRnorm file; `from_array`
    builds a map without a file.
    """

    def __init__(self, celltype, chromosome, resolution=1000):
        import pandas
        self.resolution = int(resolution)
        self.filename = DATA_DIR.format(celltype, chromosome, self.resolution)
        self.chromosome = chromosome
        self.celltype = celltype
        self.map = pandas.read_csv(self.filename, sep="\t", usecols=[1, 3, 4, 5, 6], engine="c",
                                   dtype="float64").values
        self.regions = numpy.union1d(self.map[:, 0], self.map[:, 1])

    @classmethod
    def from_array(cls, map_array, resolution, celltype="", chromosome=0):
        self = cls.__new__(cls)
        m = numpy.array(map_array, dtype=numpy.float64)
        if m.ndim != 2 or m.shape[1] != 5:
            raise ValueError("map must have shape (n_contacts, 5): mid1, mid2, count, p, q")
        self.map, self.resolution = m, int(resolution)
        self.filename, self.celltype, self.chromosome = "", celltype, chromosome
        self.regions = numpy.union1d(m[:, 0], m[:, 1])
        return self

    def decimate(self, resolution=5000):
        """Decimate the map to a lower resolution: midpoints are rounded to the
        coarser grid, counts summed, p-values multiplied, q-values minimised
        (`blueberry/datatypes.pyx:317-339`).  In place, returns None."""
        resolution = int(resolution)
        self.resolution = resolution
        mids = (self.map[:, :2].astype(numpy.int64) + resolution) // resolution * resolution \
            - resolution // 2
        key, first, inv = numpy.unique(mids, axis=0, return_index=True, return_inverse=True)
        # one row per (mid1, mid2) in the order of FIRST occurrence: the reference collects
        # them in a dict (`datatypes.pyx:330-336`), whose order that is
        order = numpy.argsort(first, kind="stable")
        rank = numpy.empty_like(order)
        rank[order] = numpy.arange(order.shape[0])
        key, inv = key[order], rank[inv.ravel()]
        count = numpy.zeros(key.shape[0])
        numpy.add.at(count, inv, self.map[:, 2])
        p = numpy.ones(key.shape[0])
        numpy.multiply.at(p, inv, self.map[:, 3])
        q = numpy.ones(key.shape[0])
        numpy.minimum.at(q, inv, self.map[:, 4])
        self.map = numpy.column_stack([key.astype(numpy.float64), count, p, q])
        self.regions = numpy.union1d(self.map[:, 0], self.map[:, 1])

    def contacts(self):
        """All contacts with a q-value <= Q_LOWER_BOUND, as (mid1, mid2) rows
        (`blueberry/datatypes.pyx:341-350`)."""
        from .utils import Q_LOWER_BOUND
        return self.map[self.map[:, 4] <= Q_LOWER_BOUND, :2]

    def _bins(self):
        res = self.resolution
        b1 = ((self.map[:, 0] - res / 2.0) / res).astype(numpy.int64)
        b2 = ((self.map[:, 1] - res / 2.0) / res).astype(numpy.int64)
        return b1, b2

    def _statistic(self, statistic):
        col = {"count": 2, "p": 3, "q": 4}.get(statistic)
        if col is None:
            raise ValueError                     # as the reference (pyx:386)
        return self.map[:, col]

    def to_matrix(self, statistic="count", n_bins=None):
        """Convert the map from column format to a 2d (n_bins+1)^2 matrix holding
        `statistic` at [bin(mid1), bin(mid2)] (one triangle, as the reference
        fills it; `blueberry/datatypes.pyx:352-388`)."""
        vals = self._statistic(statistic)
        if n_bins is None:
            kr = numpy.loadtxt(KR_NORM.format(self.celltype, self.chromosome,
                                              self.resolution // 1000))
            n_bins = numpy.atleast_1d(kr).shape[0]
        d = int(n_bins) + 1
        b1, b2 = self._bins()
        matrix = numpy.zeros((d, d))
        _assign_last_wins(matrix, b1, b2, vals)  # later rows win, as in the reference
        return matrix

    def to_sparse(self, statistic="count", n_bins=None):
        """The same content as a scipy.sparse COO matrix of shape (n_bins+1)^2 --
        what `StructureSolver.fit` takes for blocked-sparse input."""
        import scipy.sparse
        vals = self._statistic(statistic)
        b1, b2 = self._bins()
        if n_bins is None:
            n_bins = int(max(b1.max(initial=0), b2.max(initial=0)))
        d = int(n_bins) + 1
        return scipy.sparse.coo_matrix((vals, (b1, b2)), shape=(d, d))
