/*
 * oracle/asan_driver.c -- sanitizer harness of the C restatement (TEST INFRASTRUCTURE ONLY).
 *
 * `make -C oracle asan` compiles grape_ref.c together with this driver under
 * -fsanitize=address,undefined ("ASan/UBSan on the CPU restatement") and
 * tests/test_oracle.py runs the binary: every entry point of the restatement on small ragged problems
 * (all Pade branches, both gradient routes, per-trajectory controls, the state running cost, the
 * caller-supplied chi), so that out-of-bounds accesses and undefined behaviour in the checker itself
 * cannot hide behind matching numbers.  Prints one checksum line per case and "asan-driver OK".
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

int grape_ref_eval(int N, int L, int K, int N_T, const double *tlist, const double *H0_, const double *Hc_,
                   int hc_per_traj, const double *psi0_, const double *target_, const double *weights, int functional,
                   int gradient_method, const double *pulsevals, double *J, double *G, double *tau_, double *psiT_,
                   double *tau_grads_, int nthreads);
int grape_ref_eval_b(int N, int L, int K, int N_T, const double *tlist, const double *H0_, const double *Hc_,
                     int hc_per_traj, const double *psi0_, const double *target_, const double *weights, int functional,
                     int gradient_method, const double *pulsevals, double *J, double *G, double *tau_, double *psiT_,
                     double *tau_grads_, int nthreads, const double *D_, int d_per_traj, double lambda_b);
int grape_ref_eval_chi(int N, int L, int K, int N_T, const double *tlist, const double *H0_, const double *Hc_,
                       int hc_per_traj, const double *psi0_, const double *target_, const double *weights,
                       int gradient_method, const double *pulsevals, double *J, double *G, double *tau_, double *psiT_,
                       double *tau_grads_, int nthreads, const double *D_, int d_per_traj, double lambda_b,
                       const double *chi_in_);
int grape_ref_expm(int n, double *A_, double *E_, double *work_, int *squarings);

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static double urand(void) {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (double)((rng_state >> 11) & 0xFFFFFFFFFFFFFull) / 4503599627370496.0 - 0.5;
}

/* exactly-sized heap arrays: an overrun by one element is an ASan report */
static double *rnd(size_t n, double scale) {
    double *p = (double *)malloc(sizeof(double) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) p[i] = scale * urand();
    return p;
}

static void hermitise(double *m, int N) { /* column-major interleaved complex */
    for (int j = 0; j < N; ++j)
        for (int i = 0; i <= j; ++i) {
            double *a = m + 2 * ((size_t)j * N + i), *b = m + 2 * ((size_t)i * N + j);
            if (i == j) { a[1] = 0.0; continue; }
            b[0] = a[0]; b[1] = -a[1];
        }
}

static int run_case(int N, int L, int K, int N_T, double dt, int hc_per_traj, int functional, int method, int herm,
                    int with_d, int with_chi) {
    const size_t nn = (size_t)N * N;
    double *tl = (double *)malloc(sizeof(double) * (N_T + 1));
    tl[0] = 0.0;
    for (int n = 0; n < N_T; ++n) tl[n + 1] = tl[n] + dt * (1.0 + 0.3 * urand()); /* non-uniform grid */
    double *H0 = rnd(2 * nn * K, 1.0 / sqrt((double)N)), *Hc = rnd(2 * nn * L * (hc_per_traj ? K : 1), 1.0 / sqrt((double)N));
    if (herm) {
        for (int k = 0; k < K; ++k) hermitise(H0 + 2 * nn * k, N);
        for (int q = 0; q < L * (hc_per_traj ? K : 1); ++q) hermitise(Hc + 2 * nn * q, N);
    }
    double *psi0 = rnd(2 * (size_t)K * N, 1.0), *tgt = rnd(2 * (size_t)K * N, 1.0), *w = rnd(K, 1.0);
    for (int k = 0; k < K; ++k) {
        double n0 = 0, n1 = 0;
        for (int i = 0; i < 2 * N; ++i) { n0 += psi0[2 * (size_t)k * N + i] * psi0[2 * (size_t)k * N + i]; n1 += tgt[2 * (size_t)k * N + i] * tgt[2 * (size_t)k * N + i]; }
        for (int i = 0; i < 2 * N; ++i) { psi0[2 * (size_t)k * N + i] /= sqrt(n0); tgt[2 * (size_t)k * N + i] /= sqrt(n1); }
        w[k] = 1.0 + w[k];
    }
    double *x = rnd((size_t)L * N_T, 0.4), *G = rnd((size_t)L * N_T, 0.0), *tau = rnd(2 * (size_t)K, 0.0);
    double *psiT = rnd(2 * (size_t)K * N, 0.0), *tg = rnd(2 * (size_t)K * L * N_T, 0.0);
    double *D = with_d ? rnd(2 * nn, 0.5) : NULL, *chi = with_chi ? rnd(2 * (size_t)K * N, 1.0) : NULL;
    if (D) hermitise(D, N);
    double J = 0.0;
    int rc;
    if (with_chi) rc = grape_ref_eval_chi(N, L, K, N_T, tl, H0, Hc, hc_per_traj, psi0, tgt, w, method, x, &J, G, tau, psiT, tg, 1, D, 0, 0.7, chi);
    else if (with_d) rc = grape_ref_eval_b(N, L, K, N_T, tl, H0, Hc, hc_per_traj, psi0, tgt, w, functional, method, x, &J, G, tau, psiT, tg, 1, D, 0, 0.7);
    else rc = grape_ref_eval(N, L, K, N_T, tl, H0, Hc, hc_per_traj, psi0, tgt, w, functional, method, x, &J, G, tau, psiT, tg, 1);
    double cs = J;
    for (int i = 0; i < L * N_T; ++i) cs += G[i] * (1 + i % 7);
    printf("N=%d L=%d K=%d N_T=%d dt=%.2f f=%d m=%d herm=%d D=%d chi=%d rc=%d checksum=%.12e\n", N, L, K, N_T, dt,
           functional, method, herm, with_d, with_chi, rc, cs);
    /* functional-only call: G == NULL, optional outputs NULL */
    int rc2 = grape_ref_eval(N, L, K, N_T, tl, H0, Hc, hc_per_traj, psi0, tgt, NULL, functional, method, x, &J, NULL, tau, NULL, NULL, 1);
    free(tl); free(H0); free(Hc); free(psi0); free(tgt); free(w); free(x); free(G); free(tau); free(psiT); free(tg);
    free(D); free(chi);
    return (rc != 0 || rc2 != 0 || !isfinite(cs)) ? 1 : 0;
}

int main(void) {
    int bad = 0;
    /* every Pade branch of the restated exp (norms 0.005 .. 40), odd sizes */
    const double scales[] = {0.005, 0.1, 0.6, 1.5, 4.0, 40.0};
    for (int c = 0; c < 6; ++c) {
        const int n = 3 + 2 * c;
        double *A = rnd(2 * (size_t)n * n, scales[c]), *E = rnd(2 * (size_t)n * n, 0.0), *wk = rnd(12 * (size_t)n * n, 0.0);
        int s = -1;
        const int order = grape_ref_expm(n, A, E, wk, &s);
        printf("expm n=%d scale=%g order=%d s=%d\n", n, scales[c], order, s);
        if (order < 0) bad++;
        free(A); free(E); free(wk);
    }
    /*           N  L  K N_T  dt  hcpt f  m herm D chi */
    bad += run_case(2, 1, 1, 7, 0.05, 0, 0, 0, 1, 0, 0);
    bad += run_case(3, 2, 2, 5, 0.40, 1, 1, 1, 0, 0, 0);
    bad += run_case(5, 3, 3, 4, 1.00, 0, 2, 0, 1, 0, 0);
    bad += run_case(7, 1, 2, 3, 2.50, 0, 0, 1, 1, 1, 0);
    bad += run_case(6, 2, 2, 4, 1.00, 1, 0, 0, 0, 1, 0);
    bad += run_case(4, 2, 3, 5, 0.70, 0, 0, 0, 1, 0, 1);
    bad += run_case(4, 2, 3, 5, 0.70, 0, 0, 1, 1, 1, 1);
    if (bad) { printf("asan-driver FAILED (%d)\n", bad); return 1; }
    printf("asan-driver OK\n");
    return 0;
}
