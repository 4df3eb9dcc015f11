/* examples/solve.c -- the C-ABI used from plain C (gcc, no HIP headers):
 * band count, a small solve, and the ContactMap stage on a resident matrix feeding a
 * solver device to device -- each checked against closed forms.
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

    /* A1-A4 on one resident matrix: 6 bins, a dead bin (no contacts), scatter ->
     * normalize (all KR = 2, expected = 1: every count / 4) -> filter -> solver */
    {
        enum { NBINS = 6, D = NBINS + 1, NT = 5 };
        /* column-major (pos_i[], pos_j[], count[]), resolution 1000; bin 4 never occurs */
        double tr[3 * NT] = {0, 0, 1000, 2000, 3000, /**/ 1000, 2000, 3000, 5000, 5000, /**/ 8, 4, 12, 16, 20};
        double kr[NBINS] = {2, 2, 2, 2, 2, 2}, ke[NBINS] = {1, 1, 1, 1, 1, 1};
        double m[D * D], marg[D];
        unsigned char keep[D];
        int64_t dn = 0;
        bb_cm *cm = NULL;
        CHECK(bb_cm_create(&cm, D, 0));
        CHECK(bb_cm_scatter(cm, tr, NT, 1000));
        CHECK(bb_cm_normalize(cm, NBINS, kr, ke));
        CHECK(bb_cm_marginals(cm, marg));
        if (marg[0] != 2 + 1 || marg[4] != 0 || marg[6] != 0) return 5;   /* 8/4 + 4/4; dead; pad */
        CHECK(bb_cm_filter(cm, 0.0, &dn, keep));
        if (dn != 5 || keep[4] || keep[6] || !keep[5]) return 6;
        CHECK(bb_cm_download(cm, m, dn));
        if (m[0 * dn + 1] != 2 || m[1 * dn + 0] != 2 || m[3 * dn + 4] != 5 || m[0] != 0) return 7;
        double x5[15], h1[1];
        for (int i = 0; i < 15; i++) x5[i] = (i * 7 % 5) * 0.3 + i * 0.01;
        CHECK(bb_solver_create(&sol, dn, BB_F64, 0, 0, 1, NULL, NULL, 0));
        CHECK(bb_solver_set_wish_from_cm(sol, cm, BB_KIND_COUNTS, 3.0));   /* device to device */
        CHECK(bb_solver_set_coords(sol, x5));
        CHECK(bb_solver_iterate(sol, 1, 0.05));
        CHECK(bb_solver_get_stress_history(sol, h1, 1, &n_hist));
        CHECK(bb_solver_destroy(sol));
        CHECK(bb_cm_destroy(cm));
        /* five constraints: (0,1) c=2, (0,2) c=1, (1,3) c=3, (2,4) c=4, (3,4) c=5; delta = c^(-1/3) */
        const int pi[5] = {0, 0, 1, 2, 3}, pj[5] = {1, 2, 3, 4, 4};
        const double pc[5] = {2, 1, 3, 4, 5};
        double want = 0;
        for (int q = 0; q < 5; q++) {
            double dx = x5[3 * pi[q]] - x5[3 * pj[q]], dy = x5[3 * pi[q] + 1] - x5[3 * pj[q] + 1],
                   dz = x5[3 * pi[q] + 2] - x5[3 * pj[q] + 2];
            double r = sqrt(dx * dx + dy * dy + dz * dz) - pow(pc[q], -1.0 / 3.0);
            want += r * r;
        }
        printf("resident ContactMap -> solver: stress %.12e (expect %.12e)\n", h1[0], want);
        if (fabs(h1[0] - want) > 1e-12 * want) return 8;
    }
    /* Several maps in ONE solver (bb_solver_set_maps): two maps of 3 and 2 bins laid end to
     * end, the second starting at bin 128 (a multiple of the fp64 tile edge of a small
     * problem), each with its own step and its own stress; checked against the closed form
     * of each map's stress at the start. */
    {
        enum { TOTAL = 130 };
        const int32_t ti[2] = {0, 1}, tj[2] = {0, 1};         /* the maps' own diagonal tiles */
        const int64_t begin[3] = {0, 128, TOTAL};
        const double lrs[2] = {1.0 / 6.0, 1.0 / 4.0};          /* 1 / (2 n_m) */
        const double wa[9] = {0, 3, 4, 3, 0, 5, 4, 5, 0};      /* a 3-4-5 triangle */
        const double wb[4] = {0, 2, 2, 0};                     /* two points, 2 apart */
        static double xm[3 * TOTAL];
        double per[2], hm[2 * 3];
        for (int i = 0; i < 3 * TOTAL; i++) xm[i] = 0.0;
        xm[0] = 0; xm[3] = 1; xm[6] = 0; xm[7] = 1;            /* map 0: (0,0,0) (1,0,0) (0,1,0) */
        xm[3 * 128] = 0; xm[3 * 129] = 1;                      /* map 1: (0,0,0) (1,0,0) */
        CHECK(bb_solver_create(&sol, TOTAL, BB_F64, 0, 0, 1, ti, tj, 2));
        CHECK(bb_solver_set_maps(sol, 2, begin, lrs));
        CHECK(bb_solver_set_wish_dense_block(sol, wa, 3, 3, 0, BB_KIND_WISH, 3.0));
        CHECK(bb_solver_set_wish_dense_block(sol, wb, 2, 2, 128, BB_KIND_WISH, 3.0));
        CHECK(bb_solver_set_coords(sol, xm));
        CHECK(bb_solver_stress_maps(sol, per, 2));
        /* map 0: (1-3)^2 + (1-4)^2 + (sqrt 2 - 5)^2; map 1: (1-2)^2 */
        const double wantA = 4.0 + 9.0 + (sqrt(2.0) - 5.0) * (sqrt(2.0) - 5.0), wantB = 1.0;
        printf("two maps in one solver: stress %.12f / %.12f (expect %.12f / %.12f)\n", per[0], per[1],
               wantA, wantB);
        if (fabs(per[0] - wantA) > 1e-12 * wantA || fabs(per[1] - wantB) > 1e-12) return 9;
        CHECK(bb_solver_iterate(sol, 3, 1.0));                 /* lr = 1: the maps' own steps */
        CHECK(bb_solver_get_stress_history(sol, hm, 6, &n_hist));
        CHECK(bb_solver_get_coords(sol, xm));
        CHECK(bb_solver_destroy(sol));
        if (n_hist != 6 || !(hm[2] < hm[0]) || !(hm[3] < hm[1]) || !(hm[4] < hm[2])) return 10;
        /* two points under the SMACOF step reach their wish distance in one step */
        if (fabs(fabs(xm[3 * 129] - xm[3 * 128]) - 2.0) > 1e-12 || fabs(hm[3]) > 1e-20) return 11;
        if (xm[3 * 64] != 0.0) return 12;                      /* padding between the maps */
    }
    /* The helix again, from no start at all: the classical-MDS start with its stopping rule
     * (bb_solver_spectral_init_tol: a complete noise-free map has a rank-3 B, two products
     * instead of forty), then a few iterations with a step per block of the layout
     * (bb_solver_set_block_steps; all factors 1 = one step for all, here just the call). */
    {
        bb_layout_info lay;
        int done = -1;
        double res = -1.0, s_start = 0.0, v0[3 * N];
        unsigned r = 777u;
        for (int i = 0; i < 3 * N; i++) {
            r = r * 1664525u + 1013904223u;
            v0[i] = (r >> 8) / 16777216.0 - 0.5;
        }
        CHECK(bb_solver_create(&sol, N, BB_F64, 0, 0, 1, NULL, NULL, 0));
        CHECK(bb_solver_set_wish_dense(sol, w, N, BB_KIND_WISH, 3.0));
        CHECK(bb_solver_spectral_init_tol(sol, 40, 1e-3, v0, &done, &res));
        CHECK(bb_solver_stress(sol, &s_start));
        printf("spectral start: %d orthonormalised product(s), distance %.1e, stress %.3e (noisy start: %.3e)\n",
               done, res, s_start, hist[0]);
        if (done != 1 || !(res >= 0.0 && res < 1e-3) || !(s_start < 1e-9 * hist[0])) return 13;
        CHECK(bb_solver_layout(sol, &lay, NULL, NULL));
        double *factors = malloc(sizeof(double) * (size_t)lay.n_blocks);
        for (int64_t b = 0; b < lay.n_blocks; b++) factors[b] = 1.0;
        CHECK(bb_solver_set_block_steps(sol, factors, lay.n_blocks));
        CHECK(bb_solver_iterate(sol, 2, 1.0 / (2.0 * N)));
        CHECK(bb_solver_set_block_steps(sol, NULL, 0));
        CHECK(bb_solver_iterate(sol, 2, 1.0 / (2.0 * N)));
        CHECK(bb_solver_get_stress_history(sol, hist, 4, &n_hist));
        CHECK(bb_solver_destroy(sol));
        free(factors);
        if (n_hist != 4) return 14;
    }
    free(xs); free(x0); free(w);
    puts("C-ABI OK");
    return 0;
}
