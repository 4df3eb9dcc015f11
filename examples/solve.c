/* examples/solve.c -- the C-ABI used from plain C (gcc, no HIP headers):
 * band count + a small solve, checked against closed forms.
 *
 *   gcc -std=c99 -Iinclude examples/solve.c -o /tmp/bb_solve \
 *       -Lblueberry_amd -lblueberry_hip -Wl,-rpath,$PWD/blueberry_amd -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "blueberry_hip.h"

#define CHECK(call)                                                         \
    do {                                                                    \
        int rc_ = (call);                                                   \
        if (rc_ != BB_OK) {                                                 \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, bb_last_error()); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

int main(void) {
    /* K1: uniform 50 kb grid of 1000 bins: 179,900 pairs in the Fit-Hi-C band */
    enum { NB = 1000 };
    static double regions[NB];
    for (int i = 0; i < NB; i++) regions[i] = i * 50000.0 + 25000.0;
    int64_t count = 0;
    CHECK(bb_band_count(regions, NB, 25000, 10000000, 0, &count));
    printf("band count %lld (expect 179900)\n", (long long)count);
    if (count != 179900) return 2;

    /* S0: points on a helix, wish distances = true distances, noisy start */
    enum { N = 600, K = 40 };
    double *xs = malloc(sizeof(double) * 3 * N), *x0 = malloc(sizeof(double) * 3 * N);
    double *w = malloc(sizeof(double) * N * N), hist[K];
    unsigned s = 12345u;
    for (int i = 0; i < N; i++) {
        xs[3 * i] = 10 * cos(0.1 * i); xs[3 * i + 1] = 10 * sin(0.1 * i); xs[3 * i + 2] = 0.3 * i;
        for (int c = 0; c < 3; c++) {
            s = s * 1664525u + 1013904223u;
            x0[3 * i + c] = xs[3 * i + c] + ((s >> 8) / 16777216.0 - 0.5);
        }
    }
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            double dx = xs[3 * i] - xs[3 * j], dy = xs[3 * i + 1] - xs[3 * j + 1],
                   dz = xs[3 * i + 2] - xs[3 * j + 2];
            w[i * N + j] = sqrt(dx * dx + dy * dy + dz * dz);
        }
    bb_solver *sol = NULL;
    int64_t n_hist = 0;
    CHECK(bb_solver_create(&sol, N, BB_F64, 0, 0, 1, NULL, NULL, 0));
    CHECK(bb_solver_set_wish_dense(sol, w, N, BB_KIND_WISH, 3.0));
    CHECK(bb_solver_set_coords(sol, x0));
    CHECK(bb_solver_iterate(sol, K, 1.0 / (2.0 * N)));
    CHECK(bb_solver_get_stress_history(sol, hist, K, &n_hist));
    CHECK(bb_solver_get_coords(sol, x0));
    CHECK(bb_solver_destroy(sol));
    printf("stress %.6e -> %.6e in %lld iterations\n", hist[0], hist[K - 1], (long long)n_hist);
    for (int k = 1; k < K; k++)
        if (!(hist[k] < hist[k - 1])) return 3; /* lr = 1/2N is a majorisation step */
    if (!(hist[K - 1] < 1e-2 * hist[0])) return 4;
    free(xs); free(x0); free(w);
    puts("C-ABI OK");
    return 0;
}
