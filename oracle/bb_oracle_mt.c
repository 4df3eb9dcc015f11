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
