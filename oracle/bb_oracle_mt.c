/*
 * bb_oracle_mt.c -- the oracle's solver loop on several host cores (OpenMP).
 *
 * TEST / MEASUREMENT INFRASTRUCTURE, like bb_oracle.c: only tests/ and bench.py's
 * cpu_baseline leg may load it; the product (blueberry_amd/) never does.
 *
 * It exists for ONE purpose: a CPU baseline that uses the host cores of the GPU
 * box instead of one.  The arithmetic per pair is bb_oracle.c's pair_term
 * (docs/SPEC.md 2.2-2.3; the reference has no solver -- SURVEY.md section 0 --
 * so this is kind "port", parity unpinned against the reference, exactly like
 * bbo_solve).  Rows are dealt to the threads cyclically (row i costs i pairs, so
 * a cyclic deal balances the triangle); every thread accumulates a private
 * gradient and stress, and the privates are added in thread order, so a run is
 * reproducible for a given thread count.  tests/test_oracle.py checks it against
 * the scalar bbo_solve (summation order differs: 1e-12 relative).
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define BBO_API __attribute__((visibility("default")))
#define BBO_EPS2_F64 1e-300
#define BBO_EPS2_F32 1e-30

BBO_API int bbo_mt_max_threads(void) { return omp_get_max_threads(); }

/* K plain gradient steps X <- X - lr * g on `threads` cores; stress_hist[k] is the
 * stress at X_k.  Returns 0, or -1 if the scratch memory cannot be allocated. */
BBO_API int bbo_solve_mt(const double *wish, long n, long ld, double *X, long iters, double lr,
                         int eps2_kind, double *stress_hist, int threads)
{
    const double eps2 = eps2_kind ? BBO_EPS2_F64 : BBO_EPS2_F32;
    if (threads < 1) threads = 1;
    double *G = (double *)malloc(sizeof(double) * 3 * (size_t)n * (size_t)threads);
    double *S = (double *)malloc(sizeof(double) * (size_t)threads);
    if (!G || !S) { free(G); free(S); return -1; }
    for (long k = 0; k < iters; k++) {
#pragma omp parallel num_threads(threads)
        {
            const int t = omp_get_thread_num(), T = omp_get_num_threads();
            double *g = G + 3 * (size_t)n * (size_t)t;
            double s = 0.0;
            memset(g, 0, sizeof(double) * 3 * (size_t)n);
            for (long i = t; i < n; i += T) {
                const double *xi = X + 3 * i;
                double gx = 0.0, gy = 0.0, gz = 0.0;
                for (long j = 0; j < i; j++) {
                    const double delta = wish[i * ld + j];
                    if (!(delta > 0.0)) continue;
                    const double *xj = X + 3 * j;
                    const double dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
                    const double d = sqrt(dx * dx + dy * dy + dz * dz + eps2);
                    const double r = d - delta;
                    const double coef = 2.0 * r / d;
                    gx += coef * dx; gy += coef * dy; gz += coef * dz;
                    g[3 * j] -= coef * dx; g[3 * j + 1] -= coef * dy; g[3 * j + 2] -= coef * dz;
                    s += r * r;
                }
                g[3 * i] += gx; g[3 * i + 1] += gy; g[3 * i + 2] += gz;
            }
            S[t] = s;
            /* threads beyond T (a smaller team than asked for) contribute nothing */
#pragma omp barrier
#pragma omp for schedule(static)
            for (long e = 0; e < 3 * n; e++) {
                double a = 0.0;
                for (int q = 0; q < T; q++) a += G[3 * (size_t)n * (size_t)q + e];
                X[e] -= lr * a;
            }
#pragma omp single
            {
                double a = 0.0;
                for (int q = 0; q < T; q++) a += S[q];
                if (stress_hist) stress_hist[k] = a;
            }
        }
    }
    free(G);
    free(S);
    return 0;
}

/* The same loop with heavy-ball momentum (bbo_solve_momentum: V <- mu V - lr g,
 * X <- X + V, V_0 = 0) on `threads` cores: the K = 20 twin of the depth-parity tests
 * (tests/test_gpu_parity.py).  mu = 0 is bbo_solve_mt. */
BBO_API int bbo_solve_momentum_mt(const double *wish, long n, long ld, double *X, long iters,
                                  double lr, double mu, int eps2_kind, double *stress_hist,
                                  int threads)
{
    const double eps2 = eps2_kind ? BBO_EPS2_F64 : BBO_EPS2_F32;
    if (threads < 1) threads = 1;
    double *G = (double *)malloc(sizeof(double) * 3 * (size_t)n * (size_t)threads);
    double *S = (double *)malloc(sizeof(double) * (size_t)threads);
    double *V = (double *)calloc(3 * (size_t)n, sizeof(double));
    if (!G || !S || !V) { free(G); free(S); free(V); return -1; }
    for (long k = 0; k < iters; k++) {
#pragma omp parallel num_threads(threads)
        {
            const int t = omp_get_thread_num(), T = omp_get_num_threads();
            double *g = G + 3 * (size_t)n * (size_t)t;
            double s = 0.0;
            memset(g, 0, sizeof(double) * 3 * (size_t)n);
            for (long i = t; i < n; i += T) {
                const double *xi = X + 3 * i;
                double gx = 0.0, gy = 0.0, gz = 0.0;
                for (long j = 0; j < i; j++) {
                    const double delta = wish[i * ld + j];
                    if (!(delta > 0.0)) continue;
                    const double *xj = X + 3 * j;
                    const double dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
                    const double d = sqrt(dx * dx + dy * dy + dz * dz + eps2);
                    const double r = d - delta;
                    const double coef = 2.0 * r / d;
                    gx += coef * dx; gy += coef * dy; gz += coef * dz;
                    g[3 * j] -= coef * dx; g[3 * j + 1] -= coef * dy; g[3 * j + 2] -= coef * dz;
                    s += r * r;
                }
                g[3 * i] += gx; g[3 * i + 1] += gy; g[3 * i + 2] += gz;
            }
            S[t] = s;
#pragma omp barrier
#pragma omp for schedule(static)
            for (long e = 0; e < 3 * n; e++) {
                double a = 0.0;
                for (int q = 0; q < T; q++) a += G[3 * (size_t)n * (size_t)q + e];
                V[e] = mu * V[e] - lr * a;
                X[e] += V[e];
            }
#pragma omp single
            {
                double a = 0.0;
                for (int q = 0; q < T; q++) a += S[q];
                if (stress_hist) stress_hist[k] = a;
            }
        }
    }
    free(G);
    free(S);
    free(V);
    return 0;
}

/* The solver loop over a TILE LIST with the wish distances formed on the fly from
 * generating coordinates: delta_ij = |x*_i - x*_j| for every
 * pair i < j < n inside a stored vw x vw tile (tile_I[t], tile_J[t]), tile_I <= tile_J;
 * every other pair carries no constraint.  No matrix exists, so the headline size
 * (N = 50,000: 20 GB dense) and BASELINE config 5 (N = 309,568, block-sparse) can be
 * checked against the oracle on the GPU box, and bench.py's cpu_baseline can time the
 * config-5 workload on its own pair set.
 *
 * delta_f32 != 0 rounds delta to float, as the device's fp32 pack does
 * (blueberry_amd/csrc/bb_solver_kernels.h gen_units_kernel: fp64 sqrt, then (T)v, flushed
 * to 0 = "no constraint" below `flush`), so the oracle and the device see the SAME wish
 * distances and what is compared is the iteration, not the input's rounding.
 * Tiles are dealt to the threads cyclically; private gradients, added in thread order.
 * Heavy-ball momentum as bbo_solve_momentum (mu = 0: plain steps).  Returns 0 / -1. */
static int solve_gen_impl(const double *xstar, long n, const int *tile_I, const int *tile_J,
                          long n_tiles, long vw, double *X, long iters, double lr, double mu,
                          int eps2_kind, int delta_f32, double flush, double *stress_hist,
                          int threads, const double *bin_scale)
{
    const double eps2 = eps2_kind ? BBO_EPS2_F64 : BBO_EPS2_F32;
    if (threads < 1) threads = 1;
    double *G = (double *)malloc(sizeof(double) * 3 * (size_t)n * (size_t)threads);
    double *S = (double *)malloc(sizeof(double) * (size_t)threads);
    double *V = (double *)calloc(3 * (size_t)n, sizeof(double));
    if (!G || !S || !V) { free(G); free(S); free(V); return -1; }
    for (long k = 0; k < iters; k++) {
#pragma omp parallel num_threads(threads)
        {
            const int t = omp_get_thread_num(), T = omp_get_num_threads();
            double *g = G + 3 * (size_t)n * (size_t)t;
            double s = 0.0;
            memset(g, 0, sizeof(double) * 3 * (size_t)n);
            for (long q = t; q < n_tiles; q += T) {
                const long i0 = (long)tile_I[q] * vw, j0 = (long)tile_J[q] * vw;
                const long i1 = i0 + vw < n ? i0 + vw : n, j1 = j0 + vw < n ? j0 + vw : n;
                for (long i = i0; i < i1; i++) {
                    const double *xi = X + 3 * i, *si = xstar + 3 * i;
                    double gx = 0.0, gy = 0.0, gz = 0.0;
                    for (long j = (j0 > i + 1 ? j0 : i + 1); j < j1; j++) {
                        const double *sj = xstar + 3 * j;
                        const double ax = si[0] - sj[0], ay = si[1] - sj[1], az = si[2] - sj[2];
                        double delta = sqrt(ax * ax + ay * ay + az * az);
                        if (delta < flush) delta = 0.0;
                        if (delta_f32) delta = (double)(float)delta;
                        if (!(delta > 0.0)) continue;
                        const double *xj = X + 3 * j;
                        const double dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
                        const double d = sqrt(dx * dx + dy * dy + dz * dz + eps2);
                        const double r = d - delta;
                        const double coef = 2.0 * r / d;
                        gx += coef * dx; gy += coef * dy; gz += coef * dz;
                        g[3 * j] -= coef * dx; g[3 * j + 1] -= coef * dy; g[3 * j + 2] -= coef * dz;
                        s += r * r;
                    }
                    g[3 * i] += gx; g[3 * i + 1] += gy; g[3 * i + 2] += gz;
                }
            }
            S[t] = s;
#pragma omp barrier
#pragma omp for schedule(static)
            for (long e = 0; e < 3 * n; e++) {
                double a = 0.0;
                for (int q = 0; q < T; q++) a += G[3 * (size_t)n * (size_t)q + e];
                if (bin_scale) a *= bin_scale[e / 3];
                V[e] = mu * V[e] - lr * a;
                X[e] += V[e];
            }
#pragma omp single
            {
                double a = 0.0;
                for (int q = 0; q < T; q++) a += S[q];
                if (stress_hist) stress_hist[k] = a;
            }
        }
    }
    free(G);
    free(S);
    free(V);
    return 0;
}

BBO_API int bbo_solve_gen_mt(const double *xstar, long n, const int *tile_I, const int *tile_J,
                             long n_tiles, long vw, double *X, long iters, double lr, double mu,
                             int eps2_kind, int delta_f32, double flush, double *stress_hist,
                             int threads)
{
    return solve_gen_impl(xstar, n, tile_I, tile_J, n_tiles, vw, X, iters, lr, mu, eps2_kind,
                          delta_f32, flush, stress_hist, threads, NULL);
}

/* The same with a step per bin (bb_solver_set_bin_steps, docs/SPEC.md 2.4.1): bin i moves
 * by lr * bin_scale[i] * g_i -- the gradient is scaled where it leaves the sum, then the
 * uniform step and the momentum apply, as in the kernels. */
BBO_API int bbo_solve_gen_steps_mt(const double *xstar, long n, const int *tile_I,
                                   const int *tile_J, long n_tiles, long vw, double *X, long iters,
                                   double lr, double mu, int eps2_kind, int delta_f32,
                                   double flush, double *stress_hist, int threads,
                                   const double *bin_scale)
{
    return solve_gen_impl(xstar, n, tile_I, tile_J, n_tiles, vw, X, iters, lr, mu, eps2_kind,
                          delta_f32, flush, stress_hist, threads, bin_scale);
}
