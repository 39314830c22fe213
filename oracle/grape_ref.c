/*
 * oracle/grape_ref.c -- CPU restatement (plain C99 + OpenMP) of the GRAPE hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg
 * of bench.py may load this library, and only as the checker / the timed CPU baseline.
 * The product path (grape.jl_amd/csrc) never links or calls it.
 *
 * What it restates (file:line into /root/reference):
 *   forward sweep, storage, tau, J_T ............. src/optimize.jl:696-768
 *   chi boundary, rho, normalize_chis! ........... src/optimize.jl:845-869, 1017-1038
 *   backward sweep with the gradient generator ... src/optimize.jl:873-911
 *                                                  docs/src/background.md:443-497
 *   :taylor gradient + taylor_grad_step! ......... src/optimize.jl:913-994, 604-653
 *   -2 Re sum_k, control-major index ............. src/optimize.jl:574-584
 *   pulsevals layout ............................. src/workspace.jl:159-162
 *
 * Arithmetic the reference delegates to packages that are NOT vendored under
 * /root/reference (Project.toml:21-30, no Manifest): QuantumPropagators `ExpProp`
 * (U = exp(-i H dt), state <- U state), QuantumGradientGenerators `GradGenerator`
 * (block upper-triangular generator, densified by ExpProp) and Julia's stdlib
 * `LinearAlgebra.exp` (Higham 2005 scaling-and-squaring Pade, orders 3/5/7/9 for
 * ||A||_1 <= 2.1 and order 13 with s = ceil(log2(||A||_1 / 5.4)) squarings otherwise,
 * solved with LAPACK gesv = LU with partial pivoting), behind LAPACK's gebal('B') balancing
 * (permutation + power-of-two diagonal scaling, undone on the result) as in Julia's `exp!`.
 * They are restated here from their published algorithms.
 *
 * PARITY: no numeric golden vectors exist in the reference and Julia is absent from this
 * image, so bit-level parity with the Julia run is UNPINNED.  This file is pinned instead
 * against scipy.linalg.expm, the closed-form two-level result, finite differences and the
 * numpy oracle (tests/test_oracle.py).
 *
 * Layout: complex numbers are interleaved (re,im) doubles; matrices are column-major
 * (Julia layout); H0 is K x (N x N), Hc is L x (N x N) (shared) or K x L x (N x N).
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double complex cplx;

enum { GRAPE_REF_SM = 0, GRAPE_REF_SS = 1, GRAPE_REF_RE = 2 };
enum { GRAPE_REF_GRADGEN = 0, GRAPE_REF_TAYLOR = 1, GRAPE_REF_FRECHET = 2 };

/* ---------- dense kernels (column-major) ---------- */

/* Optional BLAS / LAPACK underneath the dense kernels (timed CPU baseline of bench.py: the reference runs Julia's exp! on
 * OpenBLAS, so the stated baseline should too).  grape_ref_set_blas() receives the Fortran-interface entry points
 * zgemm_ and zgesv_ (LP64) of a library the CALLER has loaded -- oracle/grape_ref.py passes the OpenBLAS that scipy
 * bundles, set to one thread: the parallelism stays the OpenMP loop over trajectories, as in the reference
 * (src/optimize.jl:720, 876).  NULL pointers (the default) select the plain C loops below. */
typedef void (*zgemm_fn)(const char *, const char *, const int *, const int *, const int *, const cplx *, const cplx *, const int *,
                         const cplx *, const int *, const cplx *, cplx *, const int *);
typedef void (*zgesv_fn)(const int *, const int *, cplx *, const int *, int *, cplx *, const int *, int *);
static zgemm_fn blas_zgemm = NULL;
static zgesv_fn blas_zgesv = NULL;
void grape_ref_set_blas(void *zgemm, void *zgesv_) {
    blas_zgemm = (zgemm_fn)zgemm;
    blas_zgesv = (zgesv_fn)zgesv_;
}

/* C = A*B (n x n) */
static void zgemm_nn(int n, const cplx *restrict A, const cplx *restrict B, cplx *restrict C) {
    if (blas_zgemm) {
        const cplx one = 1.0, zero = 0.0;
        blas_zgemm("N", "N", &n, &n, &n, &one, A, &n, B, &n, &zero, C, &n);
        return;
    }
    memset(C, 0, sizeof(cplx) * (size_t)n * n);
    for (int j = 0; j < n; ++j) {
        cplx *restrict cj = C + (size_t)j * n;
        int k = 0;
        for (; k + 3 < n; k += 4) {
            const cplx b0 = B[(size_t)j * n + k], b1 = B[(size_t)j * n + k + 1];
            const cplx b2 = B[(size_t)j * n + k + 2], b3 = B[(size_t)j * n + k + 3];
            const cplx *restrict a0 = A + (size_t)k * n, *restrict a1 = a0 + n;
            const cplx *restrict a2 = a1 + n, *restrict a3 = a2 + n;
            for (int i = 0; i < n; ++i) cj[i] += a0[i] * b0 + a1[i] * b1 + a2[i] * b2 + a3[i] * b3;
        }
        for (; k < n; ++k) {
            const cplx b0 = B[(size_t)j * n + k];
            const cplx *restrict a0 = A + (size_t)k * n;
            for (int i = 0; i < n; ++i) cj[i] += a0[i] * b0;
        }
    }
}

/* y = A*x */
static void zgemv_n(int n, const cplx *restrict A, const cplx *restrict x, cplx *restrict y) {
    for (int i = 0; i < n; ++i) y[i] = 0;
    for (int k = 0; k < n; ++k) {
        const cplx xk = x[k];
        const cplx *restrict a = A + (size_t)k * n;
        for (int i = 0; i < n; ++i) y[i] += a[i] * xk;
    }
}

static double norm1(int n, const cplx *A) {
    double m = 0;
    for (int j = 0; j < n; ++j) {
        double s = 0;
        for (int i = 0; i < n; ++i) s += cabs(A[(size_t)j * n + i]);
        if (s > m) m = s;
    }
    return m;
}

/* Solve Q X = P in place (X overwrites P): LU with partial pivoting (LAPACK gesv). */
static int zgesv(int n, cplx *restrict Q, cplx *restrict P) {
    if (blas_zgesv) {
        int info = 0;
        int *ipiv = (int *)malloc(sizeof(int) * (size_t)n);
        blas_zgesv(&n, &n, Q, &n, ipiv, P, &n, &info);
        free(ipiv);
        return info ? -1 : 0;
    }
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = cabs(Q[(size_t)k * n + k]);
        for (int i = k + 1; i < n; ++i) {
            double v = cabs(Q[(size_t)k * n + i]);
            if (v > best) { best = v; p = i; }
        }
        if (best == 0.0) return -1;
        if (p != k) {
            for (int j = 0; j < n; ++j) {
                cplx t = Q[(size_t)j * n + k]; Q[(size_t)j * n + k] = Q[(size_t)j * n + p]; Q[(size_t)j * n + p] = t;
                t = P[(size_t)j * n + k]; P[(size_t)j * n + k] = P[(size_t)j * n + p]; P[(size_t)j * n + p] = t;
            }
        }
        const cplx inv = 1.0 / Q[(size_t)k * n + k];
        for (int i = k + 1; i < n; ++i) Q[(size_t)k * n + i] *= inv;
        for (int j = k + 1; j < n; ++j) {
            const cplx u = Q[(size_t)j * n + k];
            cplx *restrict qj = Q + (size_t)j * n;
            const cplx *restrict lk = Q + (size_t)k * n;
            for (int i = k + 1; i < n; ++i) qj[i] -= lk[i] * u;
        }
    }
    for (int j = 0; j < n; ++j) {
        cplx *restrict x = P + (size_t)j * n;
        for (int k = 0; k < n; ++k) { /* L y = b (unit lower) */
            const cplx xk = x[k];
            const cplx *restrict lk = Q + (size_t)k * n;
            for (int i = k + 1; i < n; ++i) x[i] -= lk[i] * xk;
        }
        for (int k = n - 1; k >= 0; --k) { /* U x = y */
            x[k] /= Q[(size_t)k * n + k];
            const cplx xk = x[k];
            const cplx *restrict uk = Q + (size_t)k * n;
            for (int i = 0; i < k; ++i) x[i] -= uk[i] * xk;
        }
    }
    return 0;
}

static const double PADE3[] = {120., 60., 12., 1.};
static const double PADE5[] = {30240., 15120., 3360., 420., 30., 1.};
static const double PADE7[] = {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.};
static const double PADE9[] = {17643225600., 8821612800., 2075673600., 302702400., 30270240.,
                               2162160., 110880., 3960., 90., 1.};
static const double PADE13[] = {64764752532480000., 32382376266240000., 7771770303897600.,
                                1187353796428800., 129060195264000., 10559470521600.,
                                670442572800., 33522128640., 1323241920., 40840800., 960960.,
                                16380., 182., 1.};

/* ---- balancing: LAPACK zgebal('B') as called by Julia's exp! (stdlib LinearAlgebra, dense.jl: `ilo, ihi, scale =
 * LAPACK.gebal!('B', A)` in front of the Pade evaluation, the inverse transformation behind it).  Restated from the
 * published algorithm (Parlett & Reinsch 1969; LAPACK 3.5+ form with 2-norms of the rows / columns and the factors
 * SCLFAC = 2, FACTOR = 0.95): first the permutations that push rows isolating an eigenvalue to the bottom and such columns
 * to the left, then powers of two d_i with A <- D^-1 A D until no row / column pair improves by 5 %.  Exact in floating
 * point (powers of two), so exp(A) = P D exp(A_bal) D^-1 P^T holds to the rounding of the exponential alone.
 * scale[j]: for ilo <= j <= ihi the factor d_j, outside the index of the row / column j was swapped with (0-based here). */
static int balance_on = 1;
void grape_ref_set_balance(int on) { balance_on = on; }

static void swap_rc(int n, cplx *A, int a, int b, int lcols, int kstart) {
    /* columns a <-> b over rows 0..lcols, rows a <-> b over columns kstart..n-1 (zgebal's exchange) */
    if (a == b) return;
    for (int i = 0; i <= lcols; ++i) { cplx t = A[(size_t)a * n + i]; A[(size_t)a * n + i] = A[(size_t)b * n + i]; A[(size_t)b * n + i] = t; }
    for (int j = kstart; j < n; ++j) { cplx t = A[(size_t)j * n + a]; A[(size_t)j * n + a] = A[(size_t)j * n + b]; A[(size_t)j * n + b] = t; }
}

static void zgebal(int n, cplx *A, int *ilo_, int *ihi_, double *scale) {
    int k = 0, l = n - 1;   /* active block k..l */
    /* rows isolating an eigenvalue -> bottom */
    for (int again = 1; again && l >= 0;) {
        again = 0;
        for (int i = l; i >= 0; --i) {
            int canswap = 1;
            for (int j = 0; j <= l; ++j)
                if (i != j && (creal(A[(size_t)j * n + i]) != 0.0 || cimag(A[(size_t)j * n + i]) != 0.0)) { canswap = 0; break; }
            if (canswap) {
                scale[l] = (double)i;
                swap_rc(n, A, i, l, l, k);
                again = 1;
                /* (one row left: LAPACK leaves with ILO = IHI = 1 and SCALE(1) = 1, which is its own index AND the factor one) */
                if (l == 0) { scale[0] = 1.0; *ilo_ = 0; *ihi_ = 0; return; }
                --l;
                break;
            }
        }
    }
    /* columns isolating an eigenvalue -> left */
    for (int again = 1; again;) {
        again = 0;
        for (int j = k; j <= l; ++j) {
            int canswap = 1;
            for (int i = k; i <= l; ++i)
                if (i != j && (creal(A[(size_t)j * n + i]) != 0.0 || cimag(A[(size_t)j * n + i]) != 0.0)) { canswap = 0; break; }
            if (canswap) {
                scale[k] = (double)j;
                swap_rc(n, A, j, k, l, k);
                again = 1;
                ++k;
                break;
            }
        }
    }
    for (int i = k; i <= l; ++i) scale[i] = 1.0;
    const double radix = 2.0, sfmin1 = 2.2250738585072014e-308 / 2.220446049250313e-16, sfmax1 = 1.0 / sfmin1;
    const double sfmin2 = sfmin1 * radix, sfmax2 = 1.0 / sfmin2;
    for (int noconv = 1; noconv;) {
        noconv = 0;
        for (int i = k; i <= l; ++i) {
            double c = 0.0, r = 0.0, ca = 0.0, ra = 0.0;
            for (int j = k; j <= l; ++j) {
                const cplx cji = A[(size_t)i * n + j], rij = A[(size_t)j * n + i];   /* column i, row i inside the block */
                c += creal(cji) * creal(cji) + cimag(cji) * cimag(cji);
                r += creal(rij) * creal(rij) + cimag(rij) * cimag(rij);
            }
            c = sqrt(c); r = sqrt(r);
            for (int j = 0; j <= l; ++j) ca = fmax(ca, cabs(A[(size_t)i * n + j]));      /* izamax over A(1:l, i) */
            for (int j = k; j < n; ++j) ra = fmax(ra, cabs(A[(size_t)j * n + i]));       /* izamax over A(i, k:n) */
            if (c == 0.0 || r == 0.0) continue;
            double g = r / radix, f = 1.0;
            const double s = c + r;
            while (c < g && fmax(f, fmax(c, ca)) < sfmax2 && fmin(r, fmin(g, ra)) > sfmin2) {
                f *= radix; c *= radix; ca *= radix; r /= radix; g /= radix; ra /= radix;
            }
            g = c / radix;
            while (g >= r && fmax(r, ra) < sfmax2 && fmin(fmin(f, c), fmin(g, ca)) > sfmin2) {
                f /= radix; c /= radix; g /= radix; ca /= radix; r *= radix; ra *= radix;
            }
            if (c + r >= 0.95 * s) continue;
            if (f < 1.0 && scale[i] < 1.0 && f * scale[i] <= sfmin1) continue;
            if (f > 1.0 && scale[i] > 1.0 && scale[i] >= sfmax1 / f) continue;
            scale[i] *= f;
            noconv = 1;
            const double gi = 1.0 / f;
            for (int j = k; j < n; ++j) A[(size_t)j * n + i] *= gi;      /* row i */
            for (int j = 0; j <= l; ++j) A[(size_t)i * n + j] *= f;      /* column i */
        }
    }
    *ilo_ = k; *ihi_ = l;
}

/* X <- P D X D^-1 P^T (Julia's exp!: `X[j, :] *= scale[j]`, `X[:, j] /= scale[j]` for ilo <= j <= ihi, then the swaps undone) */
static void zgebak_exp(int n, cplx *X, int ilo, int ihi, const double *scale) {
    for (int j = ilo; j <= ihi; ++j) {
        const double sj = scale[j], isj = 1.0 / sj;
        for (int c = 0; c < n; ++c) X[(size_t)c * n + j] *= sj;
        for (int r = 0; r < n; ++r) X[(size_t)j * n + r] *= isj;
    }
    for (int j = ilo - 1; j >= 0; --j) {
        const int m = (int)scale[j];
        if (m == j) continue;
        for (int c = 0; c < n; ++c) { cplx t = X[(size_t)c * n + j]; X[(size_t)c * n + j] = X[(size_t)c * n + m]; X[(size_t)c * n + m] = t; }
        for (int r = 0; r < n; ++r) { cplx t = X[(size_t)j * n + r]; X[(size_t)j * n + r] = X[(size_t)m * n + r]; X[(size_t)m * n + r] = t; }
    }
    for (int j = ihi + 1; j < n; ++j) {
        const int m = (int)scale[j];
        if (m == j) continue;
        for (int c = 0; c < n; ++c) { cplx t = X[(size_t)c * n + j]; X[(size_t)c * n + j] = X[(size_t)c * n + m]; X[(size_t)c * n + m] = t; }
        for (int r = 0; r < n; ++r) { cplx t = X[(size_t)j * n + r]; X[(size_t)j * n + r] = X[(size_t)m * n + r]; X[(size_t)m * n + r] = t; }
    }
}

static int expm_core(int n, cplx *A, cplx *E, cplx *w, int *squarings);

/* E = exp(A), A is n x n column-major and is destroyed.  Higham (2005) behind gebal balancing, as in Julia's exp!.
 * work: 6*n*n cplx.  Returns the Pade order used (negative on singular solve);
 * *squarings receives s. */
int grape_ref_expm(int n, double *A_, double *E_, double *work_, int *squarings) {
    cplx *A = (cplx *)A_, *E = (cplx *)E_, *w = (cplx *)work_;
    if (!balance_on) return expm_core(n, A, E, w, squarings);
    double *scale = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int ilo = 0, ihi = n - 1;
    zgebal(n, A, &ilo, &ihi, scale);
    const int order = expm_core(n, A, E, w, squarings);
    if (order >= 0) zgebak_exp(n, E, ilo, ihi, scale);
    free(scale);
    return order;
}

static int expm_core(int n, cplx *A, cplx *E, cplx *w, int *squarings) {
    const size_t nn = (size_t)n * n;
    cplx *A2 = w, *U = w + nn, *V = w + 2 * nn, *T = w + 3 * nn, *A4 = w + 4 * nn, *A6 = w + 5 * nn;
    const double nA = norm1(n, A);
    int order, s = 0;
    if (nA <= 2.1) {
        const double *c; int nc;
        if (nA > 0.95) { c = PADE9; nc = 10; }
        else if (nA > 0.25) { c = PADE7; nc = 8; }
        else if (nA > 0.015) { c = PADE5; nc = 6; }
        else { c = PADE3; nc = 4; }
        order = nc - 1;
        zgemm_nn(n, A, A, A2);
        /* P = I; U = c1 P; V = c0 P; then P *= A2 ... */
        cplx *P = A4, *Pn = A6;
        memset(P, 0, sizeof(cplx) * nn); memset(U, 0, sizeof(cplx) * nn); memset(V, 0, sizeof(cplx) * nn);
        for (int i = 0; i < n; ++i) { P[(size_t)i * n + i] = 1; U[(size_t)i * n + i] = c[1]; V[(size_t)i * n + i] = c[0]; }
        for (int k = 1; k <= nc / 2 - 1; ++k) {
            zgemm_nn(n, P, A2, Pn);
            cplx *t = P; P = Pn; Pn = t;
            for (size_t i = 0; i < nn; ++i) { U[i] += c[2 * k + 1] * P[i]; V[i] += c[2 * k] * P[i]; }
        }
        zgemm_nn(n, A, U, T); /* U = A*U */
        for (size_t i = 0; i < nn; ++i) { E[i] = V[i] + T[i]; V[i] = V[i] - T[i]; }
        if (zgesv(n, V, E)) return -1;
    } else {
        const double sl = log2(nA / 5.4);
        if (sl > 0) {
            s = (int)ceil(sl);
            const double f = ldexp(1.0, -s);
            for (size_t i = 0; i < nn; ++i) A[i] *= f;
        }
        order = 13;
        const double *c = PADE13;
        zgemm_nn(n, A, A, A2);
        zgemm_nn(n, A2, A2, A4);
        zgemm_nn(n, A2, A4, A6);
        for (size_t i = 0; i < nn; ++i) T[i] = c[13] * A6[i] + c[11] * A4[i] + c[9] * A2[i];
        zgemm_nn(n, A6, T, U);
        for (size_t i = 0; i < nn; ++i) U[i] += c[7] * A6[i] + c[5] * A4[i] + c[3] * A2[i];
        for (int i = 0; i < n; ++i) U[(size_t)i * n + i] += c[1];
        zgemm_nn(n, A, U, T); /* T = U_final */
        for (size_t i = 0; i < nn; ++i) U[i] = c[12] * A6[i] + c[10] * A4[i] + c[8] * A2[i];
        zgemm_nn(n, A6, U, V);
        for (size_t i = 0; i < nn; ++i) V[i] += c[6] * A6[i] + c[4] * A4[i] + c[2] * A2[i];
        for (int i = 0; i < n; ++i) V[(size_t)i * n + i] += c[0];
        for (size_t i = 0; i < nn; ++i) { E[i] = V[i] + T[i]; V[i] = V[i] - T[i]; }
        if (zgesv(n, V, E)) return -1;
        for (int t = 0; t < s; ++t) {
            zgemm_nn(n, E, E, T);
            memcpy(E, T, sizeof(cplx) * nn);
        }
    }
    if (squarings) *squarings = s;
    return order;
}

/* H = H0_k + sum_l eps_l * shape_l * H_l  (ExpProp `evaluate!`) */
static void build_H(int N, int L, const cplx *H0k, const cplx *Hck, const double *eps, cplx *H) {
    const size_t nn = (size_t)N * N;
    memcpy(H, H0k, sizeof(cplx) * nn);
    for (int l = 0; l < L; ++l) {
        const double a = eps[l];
        const cplx *Hl = Hck + (size_t)l * nn;
        for (size_t i = 0; i < nn; ++i) H[i] += a * Hl[i];
    }
}

/* taylor_grad_max_order / taylor_grad_tolerance / taylor_grad_check_convergence of the reference
 * (src/optimize.jl:914-918: defaults 100, 1e-16, true), set by the tests that vary them */
static int taylor_max_order = 100, taylor_check = 1;
static double taylor_tol = 1e-16;
void grape_ref_set_taylor(int max_order, double tol, int check_convergence) {
    taylor_max_order = max_order > 0 ? max_order : 100;
    taylor_tol = tol > 0 ? tol : 1e-16;
    taylor_check = check_convergence != 0;
}

/* restatement of taylor_grad_step! (src/optimize.jl:604-653); mats column-major; tmp: 4*N.
 * check == 0 (check_convergence = false, :631, :644-651): all max_order terms, no residual test, no error */
static int taylor_grad_step(int N, cplx *out, const cplx *psi, const cplx *H, const cplx *mu,
                            double dt, cplx *tmp, int max_order, double tol, int check) {
    cplx *phi = tmp, *phi_prev = tmp + N, *Hn = tmp + 2 * N, *Hn1 = tmp + 3 * N;
    cplx *scratch = (cplx *)malloc(sizeof(cplx) * N);
    zgemv_n(N, mu, psi, phi_prev);
    zgemv_n(N, H, psi, Hn1);
    cplx alpha = -I * dt;
    for (int i = 0; i < N; ++i) out[i] = alpha * phi_prev[i];
    int converged = 0;
    for (int n = 2; n <= max_order; ++n) {
        zgemv_n(N, H, phi_prev, phi);
        zgemv_n(N, mu, Hn1, scratch);
        double nrm = 0;
        for (int i = 0; i < N; ++i) { phi[i] += scratch[i]; }
        alpha *= -I * dt / n;
        for (int i = 0; i < N; ++i) { out[i] += alpha * phi[i]; nrm += creal(phi[i]) * creal(phi[i]) + cimag(phi[i]) * cimag(phi[i]); }
        if (check && cabs(alpha) * sqrt(nrm) < tol) { converged = 1; break; }
        zgemv_n(N, H, Hn1, Hn);
        cplx *t = Hn; Hn = Hn1; Hn1 = t;
        t = phi; phi = phi_prev; phi_prev = t;
    }
    free(scratch);
    if (!check || max_order <= 1) return 0;   /* :644-651 */
    return converged ? 0 : -2;
}

/*
 * One gradient evaluation (evaluate_gradient!, src/optimize.jl:824-1014, no running costs).
 * G may be NULL (functional only == evaluate_functional).  psiT/tau_grads may be NULL.
 * gradient_method: 0 = :gradgen literal ((L+1)N block exponential), 1 = :taylor.
 * Returns 0, -1 (singular Pade solve), -2 (taylor not converged), -3 (chi norm < 1e-100).
 */
static int grape_ref_eval_gb(int N, int L, int K, int N_T, const double *tlist, const double *H0_,
                   const double *Hc_, int hc_per_traj, const double *psi0_, const double *target_,
                   const double *weights, int functional, int gradient_method,
                   const double *pulsevals, double *J, double *G, double *tau_, double *psiT_,
                   double *tau_grads_ /* K*N_T*L cplx, [k][l][n] */, int nthreads,
                   const double *D_ /* NULL or [Kd][N*N] column-major Hermitian penalty operator */,
                   int d_per_traj, double lambda_b,
                   const double *chi_in_ /* NULL or [K][N]: chi_k(T) of a user-supplied chi(), optimize.jl:845-855 */);

int grape_ref_eval(int N, int L, int K, int N_T, const double *tlist, const double *H0_,
                   const double *Hc_, int hc_per_traj, const double *psi0_, const double *target_,
                   const double *weights, int functional, int gradient_method,
                   const double *pulsevals, double *J, double *G, double *tau_, double *psiT_,
                   double *tau_grads_, int nthreads) {
    return grape_ref_eval_gb(N, L, K, N_T, tlist, H0_, Hc_, hc_per_traj, psi0_, target_, weights, functional,
                             gradient_method, pulsevals, J, G, tau_, psiT_, tau_grads_, nthreads, NULL, 0, 0.0, NULL);
}

/* same with the state-dependent running cost g_b(Psi) = <Psi|D|Psi>, xi = -D Psi
 * (src/optimize.jl:727-750, 764-766, 856-866, 897-908; test/test_state_running_cost.jl:32-40) */
int grape_ref_eval_b(int N, int L, int K, int N_T, const double *tlist, const double *H0_,
                     const double *Hc_, int hc_per_traj, const double *psi0_, const double *target_,
                     const double *weights, int functional, int gradient_method,
                     const double *pulsevals, double *J, double *G, double *tau_, double *psiT_,
                     double *tau_grads_, int nthreads, const double *D_, int d_per_traj, double lambda_b) {
    return grape_ref_eval_gb(N, L, K, N_T, tlist, H0_, Hc_, hc_per_traj, psi0_, target_, weights, functional,
                             gradient_method, pulsevals, J, G, tau_, psiT_, tau_grads_, nthreads, D_, d_per_traj, lambda_b, NULL);
}

/* same with the boundary states chi_k(T) handed in by the caller: the result of the user's
 * `chi(Psi, trajectories; tau)` (src/optimize.jl:845-855), any functional.  J is J_T_sm of the inputs (the caller
 * evaluates its own J_T from psiT / tau); everything downstream of chi (:856-1014) is unchanged. */
int grape_ref_eval_chi(int N, int L, int K, int N_T, const double *tlist, const double *H0_,
                       const double *Hc_, int hc_per_traj, const double *psi0_, const double *target_,
                       const double *weights, int gradient_method, const double *pulsevals, double *J, double *G,
                       double *tau_, double *psiT_, double *tau_grads_, int nthreads, const double *D_, int d_per_traj,
                       double lambda_b, const double *chi_in_) {
    return grape_ref_eval_gb(N, L, K, N_T, tlist, H0_, Hc_, hc_per_traj, psi0_, target_, weights, GRAPE_REF_SM,
                             gradient_method, pulsevals, J, G, tau_, psiT_, tau_grads_, nthreads, D_, d_per_traj, lambda_b,
                             chi_in_);
}

static int grape_ref_eval_gb(int N, int L, int K, int N_T, const double *tlist, const double *H0_,
                   const double *Hc_, int hc_per_traj, const double *psi0_, const double *target_,
                   const double *weights, int functional, int gradient_method,
                   const double *pulsevals, double *J, double *G, double *tau_, double *psiT_,
                   double *tau_grads_ /* K*N_T*L cplx, [k][l][n] */, int nthreads,
                   const double *D_, int d_per_traj, double lambda_b, const double *chi_in_) {
    const cplx *Dop = (const cplx *)D_;
    double *Jb_traj = (double *)calloc((size_t)K, sizeof(double));
    const size_t nn = (size_t)N * N;
    const cplx *H0 = (const cplx *)H0_, *Hc = (const cplx *)Hc_;
    const cplx *psi0 = (const cplx *)psi0_, *target = (const cplx *)target_;
    cplx *tau = (cplx *)tau_;
    cplx *storage = (cplx *)malloc(sizeof(cplx) * (size_t)K * (N_T + 1) * N); /* workspace.jl:215 */
    cplx *tg = (cplx *)calloc((size_t)K * N_T * L, sizeof(cplx));             /* workspace.jl:236 */
    int err = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif

    /* ---- forward sweep: optimize.jl:720-754 ---- */
#pragma omp parallel for schedule(dynamic) reduction(min : err)
    for (int k = 0; k < K; ++k) {
        cplx *H = (cplx *)malloc(sizeof(cplx) * nn * 9);
        cplx *A = H + nn, *U = H + 2 * nn, *work = H + 3 * nn;
        const cplx *Hck = hc_per_traj ? Hc + (size_t)k * L * nn : Hc;
        double eps[64];
        cplx *st = storage + (size_t)k * (N_T + 1) * N;
        memcpy(st, psi0 + (size_t)k * N, sizeof(cplx) * N);
        for (int n = 0; n < N_T; ++n) {
            const double dt = tlist[n + 1] - tlist[n];
            for (int l = 0; l < L; ++l) eps[l] = pulsevals[(size_t)l * N_T + n];
            build_H(N, L, H0 + (size_t)k * nn, Hck, eps, H);
            for (size_t i = 0; i < nn; ++i) A[i] = -I * dt * H[i];
            if (grape_ref_expm(N, (double *)A, (double *)U, (double *)work, NULL) < 0) err = -1;
            zgemv_n(N, U, st + (size_t)n * N, st + (size_t)(n + 1) * N); /* :732, :738 */
        }
        if (Dop) { /* J_b trapezoid, optimize.jl:727-750 */
            const cplx *Dk = Dop + (d_per_traj ? (size_t)k * nn : 0);
            cplx *dps = (cplx *)malloc(sizeof(cplx) * N);
            double jb = 0;
            for (int m = 0; m <= N_T; ++m) {
                double wq;
                if (m == 0) wq = (tlist[1] - tlist[0]) / 2.0;
                else if (m < N_T) wq = 0.5 * (tlist[m + 1] - tlist[m - 1]);
                else wq = (tlist[N_T] - tlist[N_T - 1]) / 2.0;
                zgemv_n(N, Dk, st + (size_t)m * N, dps);
                double g = 0;
                for (int i = 0; i < N; ++i) g += creal(conj(st[(size_t)m * N + i]) * dps[i]);
                jb += wq * g;
            }
            Jb_traj[k] = jb;
            free(dps);
        }
        cplx t = 0;
        for (int i = 0; i < N; ++i) t += conj(target[(size_t)k * N + i]) * st[(size_t)N_T * N + i]; /* :753 */
        tau[k] = t;
        if (psiT_) memcpy((cplx *)psiT_ + (size_t)k * N, st + (size_t)N_T * N, sizeof(cplx) * N);
        free(H);
    }

    /* ---- J_T and chi boundary (QuantumControl.Functionals; tutorial.md:349-356, 402) ---- */
    cplx f = 0;
    double ss = 0, re = 0;
    for (int k = 0; k < K; ++k) {
        const double w = weights ? weights[k] : 1.0;
        f += w * tau[k];
        ss += w * (creal(tau[k]) * creal(tau[k]) + cimag(tau[k]) * cimag(tau[k]));
        re += w * creal(tau[k]);
    }
    if (functional == GRAPE_REF_SM) *J = 1.0 - (creal(f) * creal(f) + cimag(f) * cimag(f)) / ((double)K * K);
    else if (functional == GRAPE_REF_SS) *J = 1.0 - ss / K;
    else *J = 1.0 - re / K;
    if (Dop) { /* :764-766 */
        double jb = 0;
        for (int k = 0; k < K; ++k) jb += Jb_traj[k];
        *J += lambda_b * jb;
    }

    if (G && !err) {
        const int D = (L + 1) * N;
        const size_t dd = (size_t)D * D;
#pragma omp parallel for schedule(dynamic) reduction(min : err)
        for (int k = 0; k < K; ++k) {
            const double w = weights ? weights[k] : 1.0;
            cplx coeff;
            if (functional == GRAPE_REF_SM) coeff = w * f / ((double)K * K);
            else if (functional == GRAPE_REF_SS) coeff = w * tau[k] / K;
            else coeff = w / (2.0 * K);
            cplx *chi = (cplx *)malloc(sizeof(cplx) * (size_t)(D + D + 6 * N));
            cplx *ext = chi + N, *ext2 = ext + D, *tmp = ext2 + D; /* tmp: 5*N */
            double rho = 0;
            const cplx *stT = storage + (size_t)k * (N_T + 1) * N + (size_t)N_T * N;
            const cplx *Dk = Dop ? Dop + (d_per_traj ? (size_t)k * nn : 0) : NULL;
            cplx *xi = (cplx *)malloc(sizeof(cplx) * N);
            for (int i = 0; i < N; ++i) /* optimize.jl:848-855 */
                chi[i] = chi_in_ ? ((const cplx *)chi_in_)[(size_t)k * N + i] : coeff * target[(size_t)k * N + i];
            if (Dk && lambda_b != 0.0) { /* :856-866 chi += lambda_b dt/2 xi(T), xi = -D Psi */
                const double dtl = tlist[N_T] - tlist[N_T - 1];
                zgemv_n(N, Dk, stT, xi);
                for (int i = 0; i < N; ++i) chi[i] -= (lambda_b * dtl / 2.0) * xi[i];
            }
            for (int i = 0; i < N; ++i) rho += creal(chi[i]) * creal(chi[i]) + cimag(chi[i]) * cimag(chi[i]);
            rho = sqrt(rho);                       /* :867 */
            if (rho < 1e-100) { err = -3; free(chi); free(xi); continue; } /* :1021-1025 */
            for (int i = 0; i < N; ++i) chi[i] /= rho; /* :868 */

            const cplx *Hck = hc_per_traj ? Hc + (size_t)k * L * nn : Hc;
            const cplx *st = storage + (size_t)k * (N_T + 1) * N;
            cplx *H = (cplx *)malloc(sizeof(cplx) * (nn * 2 + (size_t)L * nn));
            cplx *Hdag = H + nn, *mudag = H + 2 * nn;
            for (int l = 0; l < L; ++l) /* mu_l^dagger */
                for (int j = 0; j < N; ++j)
                    for (int i = 0; i < N; ++i)
                        mudag[(size_t)l * nn + (size_t)j * N + i] = conj(Hck[(size_t)l * nn + (size_t)i * N + j]);
            cplx *Gm = NULL, *EG = NULL, *workG = NULL;
            if (gradient_method == GRAPE_REF_GRADGEN) {
                Gm = (cplx *)malloc(sizeof(cplx) * dd * 8);
                EG = Gm + dd; workG = Gm + 2 * dd;
            } else {
                Gm = (cplx *)malloc(sizeof(cplx) * nn * 8);
                EG = Gm + nn; workG = Gm + 2 * nn;
            }
            double eps[64];
            for (int n = N_T - 1; n >= 0; --n) { /* reference n = N_T:-1:1 */
                const double dt = tlist[n + 1] - tlist[n];
                for (int l = 0; l < L; ++l) eps[l] = pulsevals[(size_t)l * N_T + n];
                build_H(N, L, H0 + (size_t)k * nn, Hck, eps, H);
                for (int j = 0; j < N; ++j)
                    for (int i = 0; i < N; ++i) Hdag[(size_t)j * N + i] = conj(H[(size_t)i * N + j]);
                const cplx *psi = st + (size_t)n * N; /* storage[k][:, n], :888-892 */
                if (gradient_method == GRAPE_REF_GRADGEN) {
                    /* A = -i * G[H^dagger] * (-dt), G block upper triangular (background.md:467-477) */
                    memset(Gm, 0, sizeof(cplx) * dd);
                    for (int b = 0; b <= L; ++b)
                        for (int j = 0; j < N; ++j)
                            for (int i = 0; i < N; ++i)
                                Gm[(size_t)(b * N + j) * D + b * N + i] = I * dt * Hdag[(size_t)j * N + i];
                    for (int l = 0; l < L; ++l)
                        for (int j = 0; j < N; ++j)
                            for (int i = 0; i < N; ++i)
                                Gm[(size_t)(L * N + j) * D + l * N + i] = I * dt * mudag[(size_t)l * nn + (size_t)j * N + i];
                    if (grape_ref_expm(D, (double *)Gm, (double *)EG, (double *)workG, NULL) < 0) err = -1;
                    for (int i = 0; i < D; ++i) ext[i] = 0;       /* GradVector / resetgradvec!, :878, :896 */
                    memcpy(ext + (size_t)L * N, chi, sizeof(cplx) * N);
                    zgemv_n(D, EG, ext, ext2);                     /* :881 */
                    for (int l = 0; l < L; ++l) {
                        cplx d = 0;
                        for (int i = 0; i < N; ++i) d += conj(ext2[(size_t)l * N + i]) * psi[i];
                        tg[((size_t)k * L + l) * N_T + n] = rho * d; /* :894 */
                    }
                    memcpy(chi, ext2 + (size_t)L * N, sizeof(cplx) * N);
                } else {
                    for (int l = 0; l < L; ++l) {
                        if (taylor_grad_step(N, ext, chi, Hdag, mudag + (size_t)l * nn, -dt, tmp, taylor_max_order, taylor_tol, taylor_check)) err = -2;
                        cplx d = 0;
                        for (int i = 0; i < N; ++i) d += conj(ext[i]) * psi[i];
                        tg[((size_t)k * L + l) * N_T + n] = rho * d; /* :970 */
                    }
                    for (size_t i = 0; i < nn; ++i) Gm[i] = I * dt * Hdag[i];
                    if (grape_ref_expm(N, (double *)Gm, (double *)EG, (double *)workG, NULL) < 0) err = -1;
                    zgemv_n(N, EG, chi, ext); /* :972 */
                    memcpy(chi, ext, sizeof(cplx) * N);
                }
                if (Dk && lambda_b != 0.0 && n > 0) { /* :897-908 inhomogeneity at interior grid points */
                    const double dtn = 0.5 * (tlist[n + 1] - tlist[n - 1]);
                    zgemv_n(N, Dk, psi, xi);
                    for (int i = 0; i < N; ++i) chi[i] -= (lambda_b * dtn / rho) * xi[i];
                }
            }
            free(Gm); free(H); free(chi); free(xi);
        }
        /* _grad_J_T_via_chi!: optimize.jl:574-584 */
        for (int l = 0; l < L; ++l)
            for (int n = 0; n < N_T; ++n) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += creal(tg[((size_t)k * L + l) * N_T + n]);
                G[(size_t)l * N_T + n] = -2.0 * s;
            }
        if (tau_grads_) memcpy(tau_grads_, tg, sizeof(cplx) * (size_t)K * N_T * L);
    }
    free(storage); free(tg); free(Jb_traj);
    return err;
}

int grape_ref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
