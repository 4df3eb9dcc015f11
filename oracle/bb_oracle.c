/*
 * bb_oracle.c -- CPU oracle for the blueberry_amd hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke
 * check in __graft_entry__.py and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  Nothing under
 * blueberry_amd/ imports, links or calls it.
 *
 * Two kinds of function live here:
 *
 *  (1) Restatements of loops that exist in the reference (jmschrei/blueberry).
 *      Each cites the reference file:line it follows and is pinned bit-exactly
 *      by golden vectors captured from the real Cython functions
 *      (tests/golden/make_golden.py -> tests/golden/ .npz files).
 *        bbo_count_band_regions      blueberry/blueberry.pyx:77-91
 *        bbo_contactmap_scatter      blueberry/datatypes.pyx:97-116
 *        bbo_contactmap_normalize    blueberry/datatypes.pyx:161-171
 *        bbo_benjamini_hochberg      blueberry/blueberry.pyx:40-75
 *        bbo_downsample              blueberry/blueberry.pyx:93-104
 *
 *  (2) The 3D-structure solver (stress, gradient, update).  The reference
 *      contains NO such code, so these functions follow
 *      this repository's own specification, docs/SPEC.md, written in the
 *      reference's loop idiom (raw double*, `for i: for j < i`, single
 *      thread).  PARITY UNPINNED against the reference for these: they are
 *      pinned instead by closed-form known-answer tests and a finite-difference
 *      gradient check (tests/test_oracle.py).
 *        bbo_counts_to_wish, bbo_stress_grad, bbo_solve, bbo_stress_grad_units
 *
 * Plain C99, no dependencies beyond libm.  Build: see oracle/Makefile.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BBO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* (1) restatements of reference loops                                        */
/* ------------------------------------------------------------------------ */

/* blueberry/blueberry.pyx:77-91.  Strict lower triangle (j < i), both bounds
 * inclusive, thresholds held as C int (pyx:82) and promoted to double for the
 * comparison, difference formed in fp64, count held in a C long. */
BBO_API long bbo_count_band_regions(const double *regions, int n, int low, int high)
{
    long t = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++) {
            double diff = regions[i] - regions[j];
            if ((double)low <= diff && diff <= (double)high)
                t += 1;
        }
    return t;
}

/* Same count restricted to rows i in [i_begin, i_end): what one rank of a
 * row-sharded run contributes (tests of the N>1 path). */
BBO_API long bbo_count_band_regions_rows(const double *regions, int n, int low, int high,
                                         int i_begin, int i_end)
{
    long t = 0;
    if (i_begin < 0) i_begin = 0;
    if (i_end > n) i_end = n;
    for (int i = i_begin; i < i_end; i++)
        for (int j = 0; j < i; j++) {
            double diff = regions[i] - regions[j];
            if ((double)low <= diff && diff <= (double)high)
                t += 1;
        }
    return t;
}

/* numpy.nan_to_num with default arguments: NaN -> 0, +inf -> DBL_MAX,
 * -inf -> -DBL_MAX (used at datatypes.pyx:102 and :171). */
static double nan_to_num(double v)
{
    if (isnan(v)) return 0.0;
    if (isinf(v)) return v > 0 ? DBL_MAX : -DBL_MAX;
    return v;
}

/* blueberry/datatypes.pyx:97-116.  `data` is the (n,3) triple array as the
 * reference's pointer arithmetic reads it: COLUMN-major (data[i], data[n+i],
 * data[2n+i]; pyx:111-113), already passed through nan_to_num (pyx:102).
 * bin = position / resolution with C double->int truncation (pyx:111-112).
 * `matrix` is (d,d) row-major, d = n_bins+1, zero-filled by the caller
 * (pyx:99).  Later triples overwrite earlier ones (plain stores, pyx:115-116).
 * Deviation: 64-bit flat index (the reference's `j*d + k` is a C int and
 * overflows for d > 46340). */
BBO_API void bbo_contactmap_scatter(const double *data, long n, int resolution,
                                    double *matrix, long d)
{
    for (long i = 0; i < n; i++) {
        int j = (int)(data[i] / resolution);
        int k = (int)(data[n + i] / resolution);
        double contactCount = data[2 * n + i];
        matrix[(long)j * d + k] = contactCount;
        matrix[(long)k * d + j] = contactCount;
    }
}

/* blueberry/datatypes.pyx:161-171.  For diagonal offset i and position j:
 * m[j][j+i] /= KRnorm[j] * KRnorm[j+i] * KRexpected[i]  (left-to-right
 * product, one division), mirrored to m[j+i][j]; then nan_to_num over the
 * WHOLE (d,d) matrix (pyx:171), d = n_bins + 1. */
BBO_API void bbo_contactmap_normalize(double *matrix, long n_bins, const double *KRnorm,
                                      const double *KRexpected)
{
    long d = n_bins + 1;
    for (long i = 0; i < n_bins; i++)
        for (long j = 0; j < n_bins - i; j++) {
            matrix[j * d + j + i] /= KRnorm[j] * KRnorm[j + i] * KRexpected[i];
            matrix[(j + i) * d + j] = matrix[j * d + j + i];
        }
    for (long k = 0; k < d * d; k++)
        matrix[k] = nan_to_num(matrix[k]);
}

/* blueberry/blueberry.pyx:40-75.  p-values already sorted ascending; running
 * maximum of min(p*n/(i+1), 1).  `p*n` is double*long -> double, then /(i+1). */
BBO_API void bbo_benjamini_hochberg(const double *p_values, long d, long n, double *q_values)
{
    double prev_q_value = 0.0;
    for (long i = 0; i < d; i++) {
        double q_value = p_values[i] * (double)n / (double)(i + 1);
        /* Cython's min(q, 1) and max(q, prev) (pyx:67-68) compare this way round: a NaN
         * p-value gives q = NaN and restarts the running maximum (golden bh_*_4..7) */
        q_value = 1.0 < q_value ? 1.0 : q_value;
        q_value = prev_q_value > q_value ? prev_q_value : q_value;
        q_values[i] = q_value;
        prev_q_value = q_value;
    }
}

/* blueberry/blueberry.pyx:93-104.  5x5 max-pool of yp1 (n1,n1) into yp5i
 * (n5,n5), IN PLACE on top of yp5i's existing contents, for the first n5-1
 * rows/cols only (the reference's `range(n5-1)`). */
BBO_API void bbo_downsample(const float *yp1, long n1, float *yp5i, long n5)
{
    for (long i = 0; i < n5 - 1; i++)
        for (long j = 0; j < n5 - 1; j++)
            for (long ni = i * 5; ni < (i + 1) * 5; ni++)
                for (long nj = j * 5; nj < (j + 1) * 5; nj++) {
                    float v = yp1[ni * n1 + nj];
                    if (v > yp5i[i * n5 + j]) yp5i[i * n5 + j] = v;
                }
}

/* ------------------------------------------------------------------------ */
/* (2) solver -- docs/SPEC.md (build-authored; no reference code exists)      */
/* ------------------------------------------------------------------------ */

/* SPEC 2.1: wish distance from a contact count.  delta = c^(-1/alpha) for a
 * finite c > 0; 0 ("no constraint") otherwise.  Dense (n,n), leading
 * dimensions ld_in / ld_out; the diagonal is always 0. */
BBO_API void bbo_counts_to_wish(const double *counts, long n, long ld_in, double alpha,
                                double *wish, long ld_out)
{
    for (long i = 0; i < n; i++)
        for (long j = 0; j < n; j++) {
            double c = counts[i * ld_in + j];
            double w = 0.0;
            if (i != j && c > 0.0 && !isinf(c)) w = pow(c, -1.0 / alpha);
            wish[i * ld_out + j] = w;
        }
}

/* SPEC 2.2: d_ij = sqrt(|x_i - x_j|^2 + eps2): a tiny eps2 under the square
 * root so that coincident points give a finite, zero force. */
#define BBO_EPS2_F64 1e-300
#define BBO_EPS2_F32 1e-30

static inline double pair_term(const double *xi, const double *xj, double delta, double eps2,
                               double *gi, double *gj)
{
    double dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
    double d2 = dx * dx + dy * dy + dz * dz + eps2;
    double d = sqrt(d2);
    double r = d - delta;
    double coef = 2.0 * r / d;
    gi[0] += coef * dx; gi[1] += coef * dy; gi[2] += coef * dz;
    gj[0] -= coef * dx; gj[1] -= coef * dy; gj[2] -= coef * dz;
    return r * r;
}

/* SPEC 2.2-2.3: stress S = sum_{j<i, delta_ij>0} (d_ij - delta_ij)^2 and its
 * gradient g (n,3), over the strict lower triangle in the reference's loop
 * idiom (blueberry.pyx:86-87: `for i: for j in range(i)`), reading wish[i][j]
 * once per pair.  eps2_kind: 0 -> fp32 clamp, 1 -> fp64 clamp. */
BBO_API double bbo_stress_grad(const double *wish, long n, long ld, const double *X,
                               int eps2_kind, double *g)
{
    double eps2 = eps2_kind ? BBO_EPS2_F64 : BBO_EPS2_F32;
    double s = 0.0;
    memset(g, 0, sizeof(double) * 3 * (size_t)n);
    for (long i = 0; i < n; i++)
        for (long j = 0; j < i; j++) {
            double delta = wish[i * ld + j];
            if (!(delta > 0.0)) continue;
            s += pair_term(X + 3 * i, X + 3 * j, delta, eps2, g + 3 * i, g + 3 * j);
        }
    return s;
}

/* SPEC 2.4: K plain gradient steps X <- X - lr * g.  stress_hist[k] is the
 * stress at X_k (before step k), k = 0..iters-1.  `scratch_g` is (n,3). */
BBO_API void bbo_solve(const double *wish, long n, long ld, double *X, long iters, double lr,
                       int eps2_kind, double *stress_hist, double *scratch_g)
{
    for (long k = 0; k < iters; k++) {
        double s = bbo_stress_grad(wish, n, ld, X, eps2_kind, scratch_g);
        if (stress_hist) stress_hist[k] = s;
        for (long e = 0; e < 3 * n; e++) X[e] -= lr * scratch_g[e];
    }
}

/* SPEC 2.4 with heavy-ball momentum: V <- mu V - lr g, X <- X + V, V_0 = 0.
 * `scratch` is (2n,3): gradient then velocity. */
BBO_API void bbo_solve_momentum(const double *wish, long n, long ld, double *X, long iters,
                                double lr, double mu, int eps2_kind, double *stress_hist,
                                double *scratch)
{
    double *g = scratch, *V = scratch + 3 * n;
    memset(V, 0, sizeof(double) * 3 * (size_t)n);
    for (long k = 0; k < iters; k++) {
        double s = bbo_stress_grad(wish, n, ld, X, eps2_kind, g);
        if (stress_hist) stress_hist[k] = s;
        for (long e = 0; e < 3 * n; e++) {
            V[e] = mu * V[e] - lr * g[e];
            X[e] += V[e];
        }
    }
}

/* The pairs owned by a contiguous range of "units" of the device layout
 * (docs/SPEC.md 3; include/blueberry_hip.h bb_layout_*): unit u belongs to
 * tile t = u / units_per_tile with block coordinates (tile_I[t], tile_J[t]),
 * and covers rows  tile_I*vw + (u % units_per_tile)*rpu .. +rpu  (rpu =
 * vw / units_per_tile) and columns tile_J*vw .. +vw, restricted to i < j < n.  Used by the world_size>1 tests
 * to play one rank's share; summing over a partition of all units must give
 * bbo_stress_grad exactly (up to summation order). */
BBO_API double bbo_stress_grad_units(const double *wish, long n, long ld, const double *X,
                                     int eps2_kind, const int32_t *tile_I, const int32_t *tile_J,
                                     long units_per_tile, long vw, long u_begin, long u_end,
                                     double *g)
{
    double eps2 = eps2_kind ? BBO_EPS2_F64 : BBO_EPS2_F32;
    double s = 0.0;
    memset(g, 0, sizeof(double) * 3 * (size_t)n);
    for (long u = u_begin; u < u_end; u++) {
        long t = u / units_per_tile, sub = u % units_per_tile;
        long rpu = vw / units_per_tile;
        long i0 = (long)tile_I[t] * vw + sub * rpu;
        long j0 = (long)tile_J[t] * vw;
        for (long i = i0; i < i0 + rpu && i < n; i++)
            for (long j = j0; j < j0 + vw && j < n; j++) {
                if (j <= i) continue;
                double delta = wish[i * ld + j];
                if (!(delta > 0.0)) continue;
                s += pair_term(X + 3 * i, X + 3 * j, delta, eps2, g + 3 * i, g + 3 * j);
            }
    }
    return s;
}

BBO_API int bbo_version(void) { return 1; }
