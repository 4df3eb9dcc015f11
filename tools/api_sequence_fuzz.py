"""Differential fuzz of CALL SEQUENCES on one solver handle: random orders of the C-ABI's
entry points (through HipEngine) against a numpy / oracle model of what docs/SPEC.md and
include/blueberry_hip.h say each call does -- inputs replaced mid-run, coordinates reset,
momentum switched, bb_solver_iterate mixed with bb_solver_grad / bb_solver_apply, stress and
matvec reads in between, calls made too early.  Every call must either agree with the model
(fp64 1e-10, fp32 1e-4 relative) or fail with the exception the model predicts; a fault, a
hang or a silent difference is a finding.  Both iteration paths (one launch per iteration up
to 4,096 bins; the unit sweep) and both dtypes.  Test infrastructure: uses the oracle.

    python tools/api_sequence_fuzz.py [n_sequences] [seed]"""
import os
import sys

import numpy

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from blueberry_amd.solver import HipEngine           # noqa: E402
from tests import _oracle                            # noqa: E402


class Model(object):
    """What the header promises, in numpy (gradient and stress from the oracle)."""

    def __init__(self, n, dtype):
        self.n, self.f64 = n, dtype == "float64"
        self.oracle = _oracle.load()
        self.w = self.X = self.V = self.G = None
        self.mu, self.hist, self.pending = 0.0, [], False
        self.per_bin = None                          # bb_solver_set_block_steps, per bin

    def _ready(self):
        if self.w is None or self.X is None:
            raise RuntimeError("not ready")

    def set_wish(self, w):
        if not self.f64:                             # below the fp32 wish floor = no constraint
            floor = float(numpy.finfo(numpy.float32).tiny)
            w = numpy.where(w < floor, 0.0, w)
        self.w = numpy.ascontiguousarray(w)

    def set_coords(self, x):
        self.X, self.V, self.hist, self.pending = x.copy(), numpy.zeros_like(x), [], False

    def sg(self):
        s, g = self.oracle.stress_grad(self.w, self.X, f64=self.f64)
        return s, (g if self.per_bin is None else g * self.per_bin)   # scaled where it leaves the sum

    def iterate(self, k, lr):
        self._ready()
        for _ in range(k):
            s, g = self.sg()
            self.V = self.mu * self.V - lr * g
            self.X = self.X + self.V
            self.hist.append(s)

    def grad(self):
        self._ready()
        self.Gs, self.G = self.sg()
        self.pending = True

    def apply(self, lr):
        self._ready()
        if not self.pending:
            raise RuntimeError("nothing pending")
        self.V = self.mu * self.V - lr * self.G
        self.X = self.X + self.V
        self.hist.append(self.Gs)
        self.pending = False

    def stress(self):
        self._ready()
        return self.sg()[0]

    def matvec(self, x):
        if self.w is None or self.pending:
            raise RuntimeError("state")
        return (self.w * self.w) @ x


def close(a, b, tol, scale=None):
    a, b = numpy.asarray(a, dtype=float), numpy.asarray(b, dtype=float)
    if a.shape != b.shape:
        return False
    if a.size == 0:
        return True
    s = scale if scale is not None else max(1e-300, numpy.abs(b).max())
    return bool(numpy.abs(a - b).max() <= tol * s)


def close_stress(a, b, tol, model):
    """Stress values: relative, above a floor of rounding noise -- at a spectral start of an
    exact map the stress IS rounding noise (1e-13 of sum delta^2 in fp64), and two correct
    computations of it agree in order of magnitude only."""
    a, b = numpy.asarray(a, dtype=float), numpy.asarray(b, dtype=float)
    if a.shape != b.shape:
        return False
    if a.size == 0:
        return True
    floor = (1e-10 if model.f64 else 1e-7) * float((model.w * model.w).sum())
    return bool((numpy.abs(a - b) <= tol * numpy.abs(b) + floor).all())


DONE = {}                                            # calls that went through, per kind


def one_sequence(rng, case):
    n = int(rng.choice([3, 17, 64, 130, 513, 700, 1100]))
    dtype = str(rng.choice(["float64", "float32"]))
    sweep = bool(rng.integers(2))
    os.environ["BB_ROW_OWNER_MAX"] = "0" if sweep else "4096"
    tol = 1e-10 if dtype == "float64" else 2e-4
    e, m = HipEngine(n, dtype), Model(n, dtype)
    log = ["n=%d %s %s" % (n, dtype, "sweep" if sweep else "row-owner")]
    lr = 1.0 / (2 * n)
    ops = ["wish_dense", "wish_counts", "wish_sparse", "wish_coords", "wish_block", "coords", "coords",
           "block_steps",
           "momentum", "iterate", "iterate", "grad", "apply", "apply", "stress", "matvec", "read",
           "spectral"]
    # (two sequences in three start from a solver that is ready, so most calls go through)
    start = ["wish_dense", "coords"] if rng.random() < 0.67 else []
    try:
        for step in range(int(rng.integers(6, 22))):
            op = start[step] if step < len(start) else str(rng.choice(ops))
            log.append(op)
            exp_err = got_err = None
            if op in ("wish_dense", "wish_counts", "wish_sparse", "wish_coords"):
                xs = _oracle.random_walk(n, seed=int(rng.integers(1 << 30)))
                w = _oracle.wish_from_coords(xs)
                if op != "wish_coords" and rng.random() < 0.5:          # an incomplete map
                    hole = numpy.triu(rng.random((n, n)) < 0.3, 1)
                    w[hole | hole.T] = 0.0
                if op == "wish_dense":
                    e.set_wish_dense(w, "wish", 3.0)
                elif op == "wish_counts":
                    with numpy.errstate(divide="ignore"):
                        c = numpy.where(w > 0, w ** -3.0, 0.0)
                    e.set_wish_dense(c, "counts", 3.0)
                    w = m.oracle.counts_to_wish(c, 3.0)
                elif op == "wish_sparse":
                    i, j = numpy.nonzero(numpy.triu(w, 1))
                    p = rng.permutation(i.size)
                    flip = rng.random(i.size) < 0.5                       # either triangle
                    r, c = numpy.where(flip, j, i)[p], numpy.where(flip, i, j)[p]
                    e.set_wish_sparse(r.astype(numpy.int64), c.astype(numpy.int64), w[i, j][p], "wish", 3.0)
                else:
                    e.set_wish_from_coords(xs)
                m.set_wish(w)
                m.pending = m.pending                                     # (inputs leave the rest alone)
                continue
            if op == "wish_block":
                # the block setter on a one-map solver: a leading block of the map replaced
                if m.w is None:
                    continue
                vw = e.layout()["vw"]
                n_sub = n if n <= vw or rng.random() < 0.5 else int(rng.integers(1, n))
                sub = _oracle.wish_from_coords(_oracle.random_walk(n_sub, seed=int(rng.integers(1 << 30))))
                e.set_wish_dense_block(sub, 0, "wish", 3.0)
                w = m.w.copy()
                t_end = -(-n_sub // vw) * vw         # the tiles the block touches end here: what
                w[:t_end, :t_end] = 0.0              # they hold outside the block is cleared
                w[:n_sub, :n_sub] = sub
                m.set_wish(w)
                DONE[op] = DONE.get(op, 0) + 1
                continue
            if op == "block_steps":
                lay = e.layout()
                per_bin_form = rng.random() < 0.5                         # either entry point
                sc = None if rng.random() < 0.3 else rng.uniform(
                    0.3, 1.7, n if per_bin_form else lay["n_blocks"])
                try:
                    (e.set_bin_steps if per_bin_form else e.set_block_steps)(sc)
                    went = True
                except RuntimeError:
                    went = False
                if went == m.pending:                 # refused exactly while a gradient is pending
                    raise AssertionError("set_block_steps went through: %s, pending: %s" % (went, m.pending))
                if went:
                    m.per_bin = (None if sc is None else sc[:, None] if per_bin_form
                                 else numpy.repeat(sc, lay["vw"])[:n, None])
                    if m.w is not None and rng.random() < 0.3:           # the degree count, while here
                        if not numpy.array_equal(e.degrees(), (m.w > 0).sum(axis=0)):
                            raise AssertionError("degrees differ")
                    want = "units" if (sweep or n > 4096) else "row_owner"
                    if e.iteration_path()[0] != want:
                        raise AssertionError("iteration path %s, expected %s" % (e.iteration_path()[0], want))
                    DONE[op] = DONE.get(op, 0) + 1
                continue
            if op == "coords":
                x = rng.standard_normal((n, 3)) * float(rng.choice([0.1, 1.0, 30.0]))
                e.set_coords(x)
                m.set_coords(x)
                continue
            if op == "momentum":
                mu = float(rng.choice([0.0, 0.3, 0.9]))
                e.set_momentum(mu)
                m.mu = mu
                continue

            def both(f_e, f_m):
                nonlocal exp_err, got_err
                r_e = r_m = None
                try:
                    r_m = f_m()
                except RuntimeError as exc:
                    exp_err = exc
                try:
                    r_e = f_e()
                except RuntimeError as exc:
                    got_err = exc
                if (exp_err is None) != (got_err is None):
                    raise AssertionError("%s: model %r, library %r" % (op, exp_err, got_err))
                key = op if exp_err is None else op + " (refused)"
                DONE[key] = DONE.get(key, 0) + 1
                return r_e, r_m

            if op == "iterate":
                k = int(rng.integers(0, 4))
                both(lambda: e.iterate(k, lr), lambda: m.iterate(k, lr))
            elif op == "grad":
                both(e.grad, m.grad)
            elif op == "apply":
                both(lambda: e.apply(lr), lambda: m.apply(lr))
            elif op == "stress":
                r_e, r_m = both(e.stress, m.stress)
                if exp_err is None and not close_stress(r_e, r_m, tol, m):
                    raise AssertionError("stress %r vs %r" % (r_e, r_m))
            elif op == "matvec":
                x = rng.standard_normal((n, 3))
                r_e, r_m = both(lambda: e.matvec_sq(x), lambda: m.matvec(x))
                if exp_err is None and not close(r_e, r_m, tol):
                    raise AssertionError("matvec differs by %g" % numpy.abs(r_e - r_m).max())
            elif op == "spectral":
                if m.w is None or m.pending or n < 4 or not (m.w[numpy.triu_indices(n, 1)] > 0).all():
                    continue                         # (complete maps only: the start is then exact)
                e.spectral_init_device(40, rng.standard_normal((n, 3)), tol=1e-3)
                x = e.get_coords()
                if not close(_oracle.wish_from_coords(x), m.w, 1e-6 if dtype == "float64" else 2e-3,
                             m.w.max()):
                    raise AssertionError("spectral start is not the map's embedding")
                m.set_coords(x)
                DONE[op] = DONE.get(op, 0) + 1
            elif op == "read":
                if m.X is None:
                    try:
                        e.get_coords()
                        raise AssertionError("get_coords before set_coords did not raise")
                    except RuntimeError:
                        pass
                    continue
                x, h = e.get_coords(), e.stress_history()
                scale = max(1.0, numpy.abs(m.X).max())
                if not close(x, m.X, tol * 50, scale):
                    raise AssertionError("coordinates differ by %g (scale %g)"
                                         % (numpy.abs(x - m.X).max(), scale))
                if not close_stress(h, numpy.array(m.hist), tol * 50, m):
                    raise AssertionError("history %r vs %r" % (h, m.hist))
        return True, log
    except AssertionError as exc:
        return False, log + ["FAIL: %s" % exc]
    finally:
        e.close()


def main():
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = numpy.random.default_rng(seed)
    bad = 0
    for case in range(n_seq):
        ok, log = one_sequence(rng, case)
        if not ok:
            bad += 1
            print("sequence %d: %s" % (case, " | ".join(log)), flush=True)
        elif case % 20 == 0:
            print("sequence %d ok (%s, %d calls)" % (case, log[0], len(log) - 1), flush=True)
    print("calls that went through:", ", ".join("%s %d" % kv for kv in sorted(DONE.items())))
    print("%d sequences, FAILURES: %d" % (n_seq, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
