"""Differential fuzz of CALL SEQUENCES on one ContactMap: random orders of the reference's
mutators (`normalize`, `filter`, `correlation`, `eigenvector` -- blueberry/datatypes.pyx:122-235)
and of the ways the matrix is read, replaced and handed to the solver, against numpy doing
what the reference's lines do (`m / (KR KR KRexp)` through the oracle's restatement,
`m[:, sums > t][sums > t]`, `numpy.corrcoef`, `eigh`).  normalize / filter / marginals must stay
BIT-exact through any sequence, NaNs included (a zero row -- the reference's padding row --
turns into NaNs in `corrcoef`, and everything after it must treat them as numpy does);
correlation and what follows it to 1e-9.  Maps start resident (`from_triples`) or on the host
(`from_matrix`).  Test infrastructure: uses the oracle.

    python tools/contactmap_sequence_fuzz.py [n_sequences] [seed]"""
import os
import sys
import warnings

import numpy

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import blueberry_amd as bb                            # noqa: E402
from tests import _oracle                            # noqa: E402

DONE = {}


def same(a, b, tol):
    a, b = numpy.asarray(a), numpy.asarray(b)
    if a.shape != b.shape:
        return False
    if tol == 0.0:
        return bool(numpy.array_equal(a, b, equal_nan=True))
    if not numpy.array_equal(numpy.isnan(a), numpy.isnan(b)):
        return False
    fin = ~numpy.isnan(b)
    if not fin.any():
        return True
    return bool(numpy.abs(a[fin] - b[fin]).max() <= tol * max(1.0, numpy.abs(b[fin]).max()))


def one_sequence(rng, oracle):
    n = int(rng.choice([4, 20, 65, 129, 300, 520]))
    res = int(rng.choice([1000, 5000]))
    nnz = int(rng.integers(n, 6 * n * max(1, n // 8)))
    i, j = rng.integers(0, n, nnz), rng.integers(0, n, nnz)
    if rng.random() < 0.5:                            # some bins without any contact at all
        dead = rng.integers(0, n, max(1, n // 10))
        ok = ~numpy.isin(i, dead) & ~numpy.isin(j, dead)
        i, j = i[ok], j[ok]
    cnt = rng.integers(1, 500, i.size).astype(float)
    triples = numpy.stack([i * float(res), j * float(res), cnt], axis=1)
    kr = 0.5 + rng.random(n)
    ke = 10.0 / (1.0 + numpy.arange(n)) + 0.2
    M = oracle.contactmap_scatter(triples, res, n)
    resident = bool(rng.integers(2))
    if resident:
        cm = bb.ContactMap.from_triples(triples, res, n, KRnorm=kr, KRexpected=ke)
    else:
        cm = bb.ContactMap.from_matrix(M.copy(), res, KRnorm=kr, KRexpected=ke)
    log = ["n=%d %s" % (n, "resident" if resident else "host")]
    have_kr, n_bins, tol = True, n, 0.0
    ops = ["normalize", "filter", "filter", "correlation", "eigenvector", "marginals", "read", "read",
           "assign", "host_write", "fit"]

    def count(op, refused=False):
        k = op + (" (refused)" if refused else "")
        DONE[k] = DONE.get(k, 0) + 1

    try:
        for _ in range(int(rng.integers(4, 14))):
            op = str(rng.choice(ops))
            log.append(op)
            d = M.shape[0]
            if op == "normalize":
                should = have_kr and M.shape == (n_bins + 1, n_bins + 1)
                try:
                    cm.normalize()
                    went = True
                except ValueError:
                    went = False
                if went != should:
                    raise AssertionError("normalize went through: %s, expected: %s" % (went, should))
                if should:
                    M = oracle.contactmap_normalize(M.copy(), kr[:n_bins], ke[:n_bins])
                count(op, not should)
            elif op == "filter":
                if d == 0:
                    cm.filter()
                    continue
                with numpy.errstate(invalid="ignore"):
                    sums = M.sum(axis=0)
                fin = numpy.sort(sums[numpy.isfinite(sums)])
                thr = 0.0
                if fin.size > 3 and rng.random() < 0.5:              # a threshold in a gap between sums
                    k = int(rng.integers(0, fin.size - 1))
                    if fin[k + 1] - fin[k] > 1e-6 * max(1.0, abs(fin[k])):
                        thr = 0.5 * (fin[k] + fin[k + 1])
                if tol and numpy.any(numpy.abs(sums[numpy.isfinite(sums)] - thr) <
                                     1e-6 * max(1.0, numpy.abs(fin).max(initial=1.0))):
                    continue                          # (inexact sums at the threshold: either side is right)
                cm.filter(thr)
                with numpy.errstate(invalid="ignore"):
                    keep = sums > thr
                # (numpy hands the reference's expression back Fortran-ordered; ours is the same
                # matrix in C order, which is what cm.matrix gives and marginals() sums)
                M = numpy.ascontiguousarray(M[keep][:, keep])
                n_bins, have_kr = int(keep.sum()), False
                count(op)
            elif op == "correlation":
                if d == 1:
                    continue                          # (numpy returns a 0-d array here)
                cm.correlation()
                if d:
                    with warnings.catch_warnings(), numpy.errstate(all="ignore"):
                        warnings.simplefilter("ignore")
                        M = numpy.corrcoef(M)
                    tol = max(tol, 1e-9)
                count(op)
            elif op == "eigenvector":
                if d == 0 or not numpy.isfinite(M).all():
                    continue
                lam, vec = numpy.linalg.eigh(0.5 * (M + M.T))
                order = numpy.argsort(-numpy.abs(lam))
                if d > 1 and abs(abs(lam[order[0]]) - abs(lam[order[1]])) < 1e-6 * abs(lam[order[0]]):
                    continue                          # (no single dominant pair)
                if not numpy.abs(M).any():
                    continue
                v = cm.eigenvector()
                ref = vec[:, order[0]]
                ref = ref * numpy.sign(ref[numpy.argmax(numpy.abs(ref))])
                if abs(cm.eigenvalue_ - lam[order[0]]) > 1e-8 * abs(lam[order[0]]) + 100 * tol:
                    raise AssertionError("eigenvalue %r vs %r" % (cm.eigenvalue_, lam[order[0]]))
                gap = abs(abs(lam[order[0]]) - abs(lam[order[1]])) / abs(lam[order[0]]) if d > 1 else 1.0
                if numpy.abs(v - ref).max() > (1e-7 + 100 * tol) / gap:
                    raise AssertionError("eigenvector differs by %g (gap %g)" % (numpy.abs(v - ref).max(), gap))
                count(op)
            elif op == "marginals":
                with numpy.errstate(invalid="ignore"):
                    ref = M.sum(axis=0) if d else numpy.zeros(0)
                if not same(cm.marginals(), ref, tol * d):
                    raise AssertionError("marginals differ")
                count(op)
            elif op == "read":
                how = int(rng.integers(3))
                got = cm.matrix if how == 0 else cm.to_host() if how == 1 else None
                if got is None:
                    if cm.shape != M.shape:
                        raise AssertionError("shape %r vs %r" % (cm.shape, M.shape))
                elif not same(got, M, tol):
                    raise AssertionError("matrix differs (%s)" % ("cm.matrix" if how == 0 else "to_host"))
                count(op)
            elif op == "assign":
                if d == 0:
                    continue
                a = rng.random((d, d)) * (rng.random((d, d)) < 0.3)
                M = numpy.triu(a, 1) + numpy.triu(a, 1).T + numpy.diag(rng.random(d))
                cm.matrix = M.copy()
                tol = 0.0
                count(op)
            elif op == "host_write":
                if d == 0:
                    continue
                h = cm.host_matrix()
                a, b, v = int(rng.integers(d)), int(rng.integers(d)), float(rng.integers(1, 50))
                h[a, b] = h[b, a] = v
                M = M.copy()
                M[a, b] = M[b, a] = v
                count(op)
            elif op == "fit":
                if d < 2 or not numpy.isfinite(M).all():
                    continue
                x0 = rng.standard_normal((d, 3))
                s = bb.StructureSolver(n_iter=2, dtype="float64", lr=1.0 / (2 * d)).fit(cm, init=x0)
                with numpy.errstate(all="ignore"):
                    X_ref, h_ref = oracle.solve(oracle.counts_to_wish(M, 3.0), x0, 2, 1.0 / (2 * d))
                t = 1e-10 + 1e3 * tol
                if not (same(s.stress_, h_ref, t) and same(s.structure_, X_ref, t)):
                    raise AssertionError("fit differs: stress %r vs %r" % (s.stress_, h_ref))
                count(op)
        return True, log
    except AssertionError as exc:
        return False, log + ["FAIL: %s" % exc]


def main():
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = numpy.random.default_rng(seed)
    oracle = _oracle.load()
    bad = 0
    for case in range(n_seq):
        ok, log = one_sequence(rng, oracle)
        if not ok:
            bad += 1
            print("sequence %d: %s" % (case, " | ".join(log)), flush=True)
        elif case % 25 == 0:
            print("sequence %d ok (%s, %d calls)" % (case, log[0], len(log) - 1), flush=True)
    print("calls that went through:", ", ".join("%s %d" % kv for kv in sorted(DONE.items())))
    print("%d sequences, FAILURES: %d" % (n_seq, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
