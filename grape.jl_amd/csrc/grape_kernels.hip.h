// grape_kernels.hip.h -- gfx950 (CDNA4) device kernels of the GRAPE gradient evaluator.
//
// Hot path restated from /root/reference/src/optimize.jl:696-768 (forward sweep),
// :824-911 (chi boundary + backward sweep + per-cell derivative overlaps) and :574-584
// (reduction over trajectories).  Nothing here is translated from the reference (which has
// no kernels): the per-cell exponential is hoisted out of the serial time loop into one
// cell-parallel MFMA kernel, the sweeps become HBM-streaming mat-vec chains and the
// derivative is a cell-parallel register-resident Krylov recursion.
//
// Conventions: a *cell* is one (trajectory k, time interval n) pair.  NP = padded Hilbert
// dimension (multiple of 16), NT = NP/16 MFMA tiles per side.  All arithmetic is fp64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4 __attribute__((ext_vector_type(4)));

#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------------------
// Pade coefficients of Higham (2005), as used by Julia's LinearAlgebra.exp (dependency D4
// of SURVEY.md section 2a).
// ---------------------------------------------------------------------------------------
__device__ __constant__ double c_pade3[4] = {120., 60., 12., 1.};
__device__ __constant__ double c_pade5[6] = {30240., 15120., 3360., 420., 30., 1.};
__device__ __constant__ double c_pade7[8] = {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1.};
__device__ __constant__ double c_pade9[10] = {17643225600., 8821612800., 2075673600., 302702400.,
                                              30270240., 2162160., 110880., 3960., 90., 1.};
#define B13_0 64764752532480000.
#define B13_1 32382376266240000.
#define B13_2 7771770303897600.
#define B13_3 1187353796428800.
#define B13_4 129060195264000.
#define B13_5 10559470521600.
#define B13_6 670442572800.
#define B13_7 33522128640.
#define B13_8 1323241920.
#define B13_9 40840800.
#define B13_10 960960.
#define B13_11 16380.
#define B13_12 182.
#define B13_13 1.

// Work statistics are counted per cell or per batch: tens of thousands of atomics that would all hit the same few
// addresses (0.4 ms of serialised L2 atomics at C2, where the whole evaluation takes 0.9 ms).  The counters are kept in
// GRAPE_STAT_SHARDS copies selected by the workgroup index; grape_get_work adds them up.
#define GRAPE_STAT_SHARDS 64
#define GRAPE_STAT_SLOTS 16
__device__ __forceinline__ void stat_add(unsigned long long *stats, int idx, unsigned long long v) {
    atomicAdd(&stats[(size_t)(blockIdx.x & (GRAPE_STAT_SHARDS - 1)) * GRAPE_STAT_SLOTS + idx], v);
}

struct ExpmArgs {
    const double *H0f;   // [K][2][NP*NP]  planar row-major drift (re plane, im plane)
    const double *Hcf;   // [Kc][L][2][NP*NP] planar row-major control operators
    const double *eps;   // [L][N_T] pulse values (control-major, workspace.jl:159-162)
    const double *shape; // nullptr or [L][N_T]
    const double *dts;   // [N_T] dt_n = tlist[n+1]-tlist[n]
    double2 *U;          // [K][N_T][NP*NP] row-major interleaved complex  U_kn = exp(-i H_kn dt_n)
    int *flags;          // [0] error flag (|=), [1] max squarings
    unsigned long long *stats; // [0] sum of squarings, [3..7] #cells with Pade order 3/5/7/9/13
    int K, L, N_T, hc_per_traj;   // K = number of generator classes (cells = K * N_T)
    const int *rep;      // nullptr or [K]: representative trajectory of every generator class (trajectories with
                         // identical H0 / control operators share their propagators)
    int *cellflag;       // [K*N_T] set by the fast kernel for cells that need the pivoted solve
    const double *n1;    // nullptr or 1-norms: [n1_k] of H0_k (per trajectory), then [Kc][L] of the control operators
    int n1_k;
    // polynomial kernel (grape_t18.hip.h): cells the four-product route must hand to the five-product one are appended
    // to cell_list (counter: flags[4]); `listed` != 0: this launch works through that list instead of all cells
    int *cell_list;
    int listed;
    // polynomial kernel, more than two controls shared by all trajectories: S_n = sum_l eps_ln shape_ln H_l, formed once per
    // evaluation by ctrl_sum_kernel ([N_T][2][NP*NP], planar like Hcf) -- the cell then fetches H0_k and S_n, whatever L
    const double *Sf;
#ifdef GRAPE_DIAG
    int ablate;  // diagnostic builds only (tools/ablate.sh): bit0 skip invert16, bit1 skip solve
    unsigned long long *stamps;  // [nblocks][16] s_memtime at phase boundaries (diagnostic builds only)
#endif
};

// ---- which cells may take the four-product route (grape_t18.hip.h, asm/gen_t16.py) ----
// The route certifies a cell by a bound on its spectral radius and hands the others to the five-product route, at the
// price of the four products already spent.  Whether it is TRIED is decided per evaluation, on the device, from the pulse
// values alone (no history of the handle enters: the same pulses take the same route and give the same bits):
//     ||H_kn dt||_F^2 = dt^2 e^T G_k e,   e = (1, eps_n1 s_n1, .., eps_nL s_nL),   G_k[a][b] = Re tr(O_a^dagger O_b)
// (Gram matrix of the operators of generator class k, from grape_create), and for a spectrum that fills [-R, R] like a
// semicircle sum lam^2 = N R^2 / 4, i.e. R_est = 2 ||H dt||_F / sqrt(N).  Spectra of another shape (round 5): the
// estimate is multiplied by kappa = sum_a |e_a| f_a kappa_a / sum_a |e_a| f_a, f_a = ||O_a||_F, kappa_a the shape factor
// of operator a from grape_create -- the ratio of its Schatten-8 norm (the quantity the kernel's bound tests) to the
// value a semicircle of the same Frobenius norm would have: 1 for the ensembles of the benchmarks, >> 1 for a low-rank
// control on a weak drift (the route is then NOT tried in vain every evaluation), 0.72 for a two-point spectrum.
// flags[6] counts the cells with kappa R_est > T16_PLAN_R (NaN included); when they are more than a quarter of the
// evaluation the four-product kernels leave at once and the five-product launch behind them walks all cells instead of
// a hand-over list (t16_skipped()).
// The estimate decides speed only: a cell it lets through is still certified (or handed over) by the kernel's own bound.
#define T16_PLAN_R 1.15   // the kernel's bound m8^(1/8) is 3.5^(1/8) = 1.17 R for a semicircle: 1.17 * 1.15 * 1.01 < 1.36
// Round 5 -- scaling and squaring around the four products (assembly kernel only, splan != nullptr): a cell whose estimate
// is beyond T16_PLAN_R is not given up: it exponentiates A / 2^s and squares the result s times (asm/gen_t16.py), with the
// smallest s <= T16_PLAN_SMAX that brings the estimate to T16_PLAN_RS (a little below T16_PLAN_R: a cell whose rigorous
// bound fails after all has wasted 4 + s products).  4 + s products in the assembly kernel, with the walk's state riding
// on the result, against five (+ squarings beyond a radius of 2) in the compiled kernel and a full-length sweep:
// C3 shape at dt = 1.5: 26.8 -> 22.9 ms, dt = 2: 30.7 -> 23.6 ms.  Only cells beyond 2^SMAX count as out of range.
#define T16_PLAN_RS 1.11
#define T16_PLAN_SMAX 3
struct T16PlanArgs {
    const double *gram;   // [KC][(L + 1)^2 + (L + 1)]: Gram matrix | shape factors
    const double *eps, *shape, *dts;
    int *flags;
    int KC, L, N_T, N;
    int *splan;           // nullptr or [KC * N_T]: squarings planned for every cell (0 where the cell is left to its own bound)
};
__global__ void __launch_bounds__(256) t16_plan_kernel(T16PlanArgs a) {
    const int cell = blockIdx.x * 256 + threadIdx.x;
    bool out = false;
    if (cell < a.KC * a.N_T) {
        const int kc = cell / a.N_T, n = cell - kc * a.N_T, M = a.L + 1;
        const double *G = a.gram + (size_t)kc * (M * M + M), *kap = G + M * M;
        double e[9];
        e[0] = 1.0;
        for (int l = 0; l < a.L; ++l) e[1 + l] = a.eps[(size_t)l * a.N_T + n] * (a.shape ? a.shape[(size_t)l * a.N_T + n] : 1.0);
        double m2 = 0., wk = 0., w = 0.;
        for (int i = 0; i < M; ++i) {
            for (int j = 0; j < M; ++j) m2 += e[i] * e[j] * G[i * M + j];
            const double wi = fabs(e[i]) * sqrt(fmax(G[i * M + i], 0.0));
            w += wi; wk += wi * kap[i];
        }
        const double dt = a.dts[n];
        const double r = 2.0 * fabs(dt) * sqrt(fmax(m2, 0.0) / (double)a.N) * (w > 0.0 ? wk / w : 1.0);
        out = !(r <= T16_PLAN_R);
        if (a.splan) {
            int sq = 0;
            if (out) {
                double rs = r;
                while (sq < T16_PLAN_SMAX && !(rs <= T16_PLAN_RS)) { rs *= 0.5; ++sq; }
                out = !(rs <= T16_PLAN_RS);      // (NaN included)
                if (out) sq = 0;
            }
            a.splan[cell] = sq;
        }
    }
    const unsigned long long m = __ballot(out);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&a.flags[6], (int)__popcll(m));
}
__device__ __forceinline__ bool t16_skipped(const int *flags, const int ncell) { return 4L * flags[6] > (long)ncell; }

// S_n = sum_l eps_ln shape_ln H_l for every time step (see ExpmArgs::Sf): one workgroup per step
struct CtrlSumArgs {
    const double *Hcf, *eps, *shape;
    double *Sf;
    int L, N_T, pp2;   // pp2 = 2 NP^2 doubles per operator
    // control operators per trajectory (round 5): one block per CELL of the generator classes, grid.x = KC N_T; block kc N_T + n
    // sums the operators of the representative trajectory of class kc
    int per_traj;
    const int *rep;
};
// (grid: N_T x parts -- a workgroup per step walked 256 dependent-latency iterations per thread at NP = 256: 0.86 ms per
// evaluation of a C5 shard; with one part per 2048 element pairs the launch is bound by what it writes)
__global__ void __launch_bounds__(256) ctrl_sum_kernel(CtrlSumArgs a) {
    const int blk = blockIdx.x, kc = a.per_traj ? blk / a.N_T : 0, n = blk - kc * a.N_T;
    if (a.per_traj) a.Hcf += (size_t)(a.rep ? a.rep[kc] : kc) * a.L * a.pp2;
    double e[8];
    for (int l = 0; l < a.L; ++l) e[l] = a.eps[(size_t)l * a.N_T + n] * (a.shape ? a.shape[(size_t)l * a.N_T + n] : 1.0);
    double2 *dst = (double2 *)(a.Sf + (size_t)blk * a.pp2);
#pragma unroll 4
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < a.pp2 / 2; i += gridDim.y * blockDim.x) {
        double2 acc = make_double2(0., 0.);
        for (int l = 0; l < a.L; ++l) {
            const double2 h = ((const double2 *)(a.Hcf + (size_t)l * a.pp2))[i];
            acc.x = fma(e[l], h.x, acc.x); acc.y = fma(e[l], h.y, acc.y);
        }
        dst[i] = acc;
    }
}

// table of expm_t16p_asm / expm_t16p4_asm (control operators per trajectory, asm/gen_t16p.py): row n = dt_n, e_1n .. e_(slots)n,
// zeros up to 2 * slots doubles, with e_l = eps_ln shape_ln (0 beyond the problem's controls)
__global__ void __launch_bounds__(256) dte_kernel(const double *eps, const double *shape, const double *dts, int L, int N_T, int slots, double *out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N_T) return;
    double *row = out + (size_t)2 * slots * n;
    row[0] = dts[n];
    for (int l = 0; l < 2 * slots - 1; ++l)
        row[1 + l] = l < L ? eps[(size_t)l * N_T + n] * (shape ? shape[(size_t)l * N_T + n] : 1.0) : 0.0;
}

// A column strip of an NP x NP complex matrix held by one wave in MFMA C/D layout:
// lane l holds rows 16*t + 4*r + (l>>4) (t = row tile, r = register) of column 16*w + (l&15).
// Register r of tile t is also exactly the B operand of k-step k0 = 16*t + 4*r.
template <int NT>
struct Strip {
    d4 re[NT];
    d4 im[NT];
};

template <int NT>
__device__ __forceinline__ void strip_zero(Strip<NT> &s) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        s.re[t] = (d4){0., 0., 0., 0.};
        s.im[t] = (d4){0., 0., 0., 0.};
    }
}

// acc += X * B, X (left operand) in LDS planar row-major with leading dimension LD,
// B (right operand) a register strip.  Complex product on real MFMA by the 3M scheme: P1 = Xr Br, P2 = Xi Bi,
// P3 = (Xr + Xi)(Br + Bi); re = P1 - P2, im = P3 - P1 - P2 -- three real products per complex one (normwise
// stable), 3 NT instead of 4 NT MFMAs per k-step; the operand sums are one VALU add each.
template <int NT>
struct Strip3 {   // the three partial products of the 3M scheme for a strip: re = p1 - p2, im = p3 - p1 - p2
    d4 p1[NT], p2[NT], p3[NT];
};
template <int NT>
__device__ __forceinline__ void strip3_zero(Strip3<NT> &q) {
#pragma unroll
    for (int t = 0; t < NT; ++t) { q.p1[t] = (d4){0., 0., 0., 0.}; q.p2[t] = (d4){0., 0., 0., 0.}; q.p3[t] = (d4){0., 0., 0., 0.}; }
}
template <int NT>
__device__ __forceinline__ void strip3_add_to(const Strip3<NT> &q, Strip<NT> &acc) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc.re[t] += q.p1[t] - q.p2[t];
        acc.im[t] += q.p3[t] - q.p1[t] - q.p2[t];
    }
}

// q += X * B in 3M form (callers that accumulate over several k blocks keep q and combine once at the end)
template <int NT, int LD>
__device__ __forceinline__ void gemm_xb3(Strip3<NT> &q, const double *__restrict__ Xre,
                                         const double *__restrict__ Xim, const Strip<NT> &B, int lane) {
    // one per-lane base address; every tile / k-step is a compile-time immediate offset from it.
    // Software pipeline without extra registers: the real-plane operands of k-step ks+1 are requested as soon as
    // the P1 MFMAs have issued (their registers are free), the imaginary-plane operands after the P2 MFMAs, so
    // every LDS read has at least NT MFMAs to land before the next operand sum needs it.  The scheduling
    // barriers keep the compiler from sinking the reads next to their first use (which exposes the LDS latency
    // once per k-step) or hoisting them further (which costs registers and ends in spills).
    const double *__restrict__ xr = Xre + (lane & 15) * LD + (lane >> 4);
    const double *__restrict__ xi = Xim + (lane & 15) * LD + (lane >> 4);
    double are[NT], aim[NT];
#pragma unroll
    for (int tr = 0; tr < NT; ++tr) {
        are[tr] = xr[16 * tr * LD];
        aim[tr] = xi[16 * tr * LD];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ks = 4 * t + r;
            const double bre = B.re[t][r], bim = B.im[t][r];
            const double bs = bre + bim;
            double as[NT];
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) as[tr] = are[tr] + aim[tr];
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) q.p1[tr] = MFMA64(are[tr], bre, q.p1[tr]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < 4 * NT) {
#pragma unroll
                for (int tr = 0; tr < NT; ++tr) are[tr] = xr[16 * tr * LD + 4 * (ks + 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) q.p2[tr] = MFMA64(aim[tr], bim, q.p2[tr]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < 4 * NT) {
#pragma unroll
                for (int tr = 0; tr < NT; ++tr) aim[tr] = xi[16 * tr * LD + 4 * (ks + 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) q.p3[tr] = MFMA64(as[tr], bs, q.p3[tr]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// acc += X * B
template <int NT, int LD>
__device__ __forceinline__ void gemm_xb(Strip<NT> &acc, const double *__restrict__ Xre,
                                        const double *__restrict__ Xim, const Strip<NT> &B, int lane) {
    Strip3<NT> q;
    strip3_zero(q);
    gemm_xb3<NT, LD>(q, Xre, Xim, B, lane);
    strip3_add_to(q, acc);
}

// Fused pair of products for the order-13 Pade polynomials:
//   T += X * (b13 A6 + b11 A4 + b9 A2),   V += X * (b12 A6 + b10 A4 + b8 A2)
// The two right operands are formed on the fly from the A2/A4/A6 strips (never materialised)
// and the LDS reads of the left operand X are shared by both accumulations.
template <int NT, int LD>
__device__ __forceinline__ void gemm_dual13(Strip<NT> &T, Strip<NT> &V, const double *__restrict__ Xre,
                                            const double *__restrict__ Xim, const Strip<NT> &A2,
                                            const Strip<NT> &A4, const Strip<NT> &A6, int wave, int lane) {
    // 3M scheme (see gemm_xb3) without a third accumulator set per product: T and V arrive holding T0 and V0, and
    //   T.re accumulates T0.re + sum P1,   T.im accumulates T0.im + sum P3,   t2 = sum P2   (likewise V, v2);
    // at the end re = T.re - t2 and im = T.im - (T.re - T0.re) - t2, with T0.re formed again from the A2/A4/A6 strips
    // that are live anyway.  6 NT instead of 8 NT MFMAs per k-step for 2 NT extra accumulator tiles.
    const double *__restrict__ xr = Xre + (lane & 15) * LD + (lane >> 4);
    const double *__restrict__ xi = Xim + (lane & 15) * LD + (lane >> 4);
    d4 t2[NT], v2[NT];
#pragma unroll
    for (int tr = 0; tr < NT; ++tr) { t2[tr] = (d4){0., 0., 0., 0.}; v2[tr] = (d4){0., 0., 0., 0.}; }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double wr = B13_13 * A6.re[t][r] + B13_11 * A4.re[t][r] + B13_9 * A2.re[t][r];
            const double wi = B13_13 * A6.im[t][r] + B13_11 * A4.im[t][r] + B13_9 * A2.im[t][r];
            const double zr = B13_12 * A6.re[t][r] + B13_10 * A4.re[t][r] + B13_8 * A2.re[t][r];
            const double zi = B13_12 * A6.im[t][r] + B13_10 * A4.im[t][r] + B13_8 * A2.im[t][r];
            const double ws = wr + wi, zs = zr + zi;
            double are[NT], aim[NT];
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) {
                are[tr] = xr[16 * tr * LD + 16 * t + 4 * r];
                aim[tr] = xi[16 * tr * LD + 16 * t + 4 * r];
            }
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) {
                T.re[tr] = MFMA64(are[tr], wr, T.re[tr]);
                V.re[tr] = MFMA64(are[tr], zr, V.re[tr]);
            }
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) {
                t2[tr] = MFMA64(aim[tr], wi, t2[tr]);
                v2[tr] = MFMA64(aim[tr], zi, v2[tr]);
            }
#pragma unroll
            for (int tr = 0; tr < NT; ++tr) {
                const double as = are[tr] + aim[tr];
                T.im[tr] = MFMA64(as, ws, T.im[tr]);
                V.im[tr] = MFMA64(as, zs, V.im[tr]);
            }
        }
    }
    const int cl = lane & 15, rg = lane >> 4;
#pragma unroll
    for (int tr = 0; tr < NT; ++tr) {
        d4 t0 = B13_7 * A6.re[tr] + B13_5 * A4.re[tr] + B13_3 * A2.re[tr];   // T0.re, V0.re once more
        d4 v0 = B13_6 * A6.re[tr] + B13_4 * A4.re[tr] + B13_2 * A2.re[tr];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (tr == wave && (4 * r + rg) == cl) { t0[r] += B13_1; v0[r] += B13_0; }
        T.im[tr] = T.im[tr] - T.re[tr] + t0 - t2[tr];
        T.re[tr] = T.re[tr] - t2[tr];
        V.im[tr] = V.im[tr] - V.re[tr] + v0 - v2[tr];
        V.re[tr] = V.re[tr] - v2[tr];
    }
}

template <int NT, int LD>
__device__ __forceinline__ void strip_store_lds(double *Xre, double *Xim, const Strip<NT> &s, int wave, int lane) {
    double *xr = Xre + (lane >> 4) * LD + 16 * wave + (lane & 15);
    double *xi = Xim + (lane >> 4) * LD + 16 * wave + (lane & 15);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xr[(16 * t + 4 * r) * LD] = s.re[t][r];
            xi[(16 * t + 4 * r) * LD] = s.im[t][r];
        }
}

template <int NT, int LD>
__device__ __forceinline__ void strip_load_lds(const double *Xre, const double *Xim, Strip<NT> &s, int wave, int lane) {
    const double *xr = Xre + (lane >> 4) * LD + 16 * wave + (lane & 15);
    const double *xi = Xim + (lane >> 4) * LD + 16 * wave + (lane & 15);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s.re[t][r] = xr[(16 * t + 4 * r) * LD];
            s.im[t][r] = xi[(16 * t + 4 * r) * LD];
        }
}

// s += c * I (identity restricted to this strip): diagonal sits in tile t == wave.
template <int NT>
__device__ __forceinline__ void strip_add_identity(Strip<NT> &s, double c, int wave, int lane) {
    const int cl = lane & 15, rg = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (t == wave && (4 * r + rg) == cl) s.re[t][r] += c;
}

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// Broadcast of lane K of every 16-lane row to the whole row (DPP row_newbcast, gfx90a+): a VALU-rate
// move, no trip through the LDS crossbar like ds_bpermute.
template <int K>
__device__ __forceinline__ double row_bcast(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + K, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + K, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_bcast_k(double v, int k) {
    switch (k) {   // k is a compile-time constant after unrolling
        case 0: return row_bcast<0>(v);   case 1: return row_bcast<1>(v);
        case 2: return row_bcast<2>(v);   case 3: return row_bcast<3>(v);
        case 4: return row_bcast<4>(v);   case 5: return row_bcast<5>(v);
        case 6: return row_bcast<6>(v);   case 7: return row_bcast<7>(v);
        case 8: return row_bcast<8>(v);   case 9: return row_bcast<9>(v);
        case 10: return row_bcast<10>(v); case 11: return row_bcast<11>(v);
        case 12: return row_bcast<12>(v); case 13: return row_bcast<13>(v);
        case 14: return row_bcast<14>(v); default: return row_bcast<15>(v);
    }
}

// In-place Gauss-Jordan inverse of one 16x16 complex tile held by ONE wave in the layout
// lane (i = lane&15, g = lane>>4), register c  <->  D[i][4*c + g]
// (so register c is directly the MFMA A operand of k-step c).  No pivoting: the Pade
// denominator q(A) = b0*exp(-A/2)(1+O(u)) has a positive definite Hermitian part for the
// propagators this path is used for; the smallest |pivot|^2 relative to b0^2 (inv_scale2 = 1/b0^2)
// is returned so that the caller can flag numerically unsafe eliminations instead of returning
// garbage.
// acc += m * (value of src in lane K of this lane's 16-lane row): one DP-ALU DPP instruction
// (v_fmac_f64 with row_newbcast, gfx90a+), i.e. the broadcast of the pivot row is folded into the FMA.
// The s_nop covers the VALU-write -> DPP-read hazard the compiler cannot see inside inline assembly.
template <int K>
__device__ __forceinline__ void fmac_rowbcast(double &acc, double src, double m) {
#ifdef GRAPE_DPP_FUSED
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc) : "v"(src), "v"(m), "n"(K));
#else
    // two full-rate 32-bit DPP moves + a plain FMA: the DP-ALU DPP form of v_fmac_f64 costs 16 cycles per
    // instruction (tools/latency_probe.hip), this sequence about 13
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(src), 0x150 + K, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(src), 0x150 + K, 0xf, 0xf, true);
    acc = fma(__hiloint2double(hi, lo), m, acc);
#endif
}

// one pivot step of invert16 (K is a template parameter: the DPP lane select is an immediate)
template <int K>
__device__ __forceinline__ void invert16_step(double (&ar)[4], double (&ai)[4], const int i, const int g,
                                              double &minrel, double &myqr, double &myqi, double inv_scale2) {
    constexpr int kg = K & 3, kc = K >> 2;
    const double pr = readlane_f64(ar[kc], 16 * kg + K);   // pivot p_k = D[k][k]
    const double pi = readlane_f64(ai[kc], 16 * kg + K);
#ifdef ABL_NO_BPERM
    const double mr = ar[kc], mi = ai[kc];
#else
    const double mr = __shfl(ar[kc], 16 * kg + i, 64);      // a_ik of this lane's row
    const double mi = __shfl(ai[kc], 16 * kg + i, 64);
#endif
    // 1/p_k: v_rcp_f64 seed + two Newton steps
    const double den = fma(pr, pr, pi * pi);
#ifdef ABL_NO_RCP
    double inv = den;
#else
    double inv = __builtin_amdgcn_rcp(den);
    inv = inv * fma(-den, inv, 2.0);
    inv = inv * fma(-den, inv, 2.0);
#endif
    const double qr = pr * inv, qi = -pi * inv;
    minrel = fmin(minrel, den * inv_scale2);
    const bool isk = (i == K);
    myqr = isk ? qr : myqr;
    myqi = isk ? qi : myqi;
    // multiplier m' = a_ik / p_k, zero for the pivot row itself (which therefore stays untouched and can
    // be read by the row broadcasts while the other rows are being updated)
    const double tr_ = fma(mr, qr, -mi * qi), ti_ = fma(mr, qi, mi * qr);
    const double nmr = isk ? 0.0 : -tr_, nmi = isk ? 0.0 : -ti_, pmi = -nmi;
    // column k of the in-place inverse: the e_k column of [D | I]
    const bool pc = (g == kg);
    ar[kc] = pc ? (isk ? 1.0 : 0.0) : ar[kc];
    ai[kc] = pc ? 0.0 : ai[kc];
#ifdef GRAPE_DPP_FUSED
#pragma unroll
    for (int c = 0; c < 4; ++c) {   // a_ij -= m' a_kj for this lane's columns j = 4c+g
        fmac_rowbcast<K>(ar[c], ar[c], nmr);
        fmac_rowbcast<K>(ar[c], ai[c], pmi);
        fmac_rowbcast<K>(ai[c], ai[c], nmr);
        fmac_rowbcast<K>(ai[c], ar[c], nmi);
    }
#else
    // the pivot row (untouched by its own step) is broadcast once per register -- 16 DPP moves -- and feeds the
    // 16 FMAs of the step (a broadcast per FMA costs twice the moves)
    double kr[4], ki[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        kr[c] = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(ar[c]), 0x150 + K, 0xf, 0xf, true),
                                 __builtin_amdgcn_mov_dpp(__double2loint(ar[c]), 0x150 + K, 0xf, 0xf, true));
        ki[c] = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(ai[c]), 0x150 + K, 0xf, 0xf, true),
                                 __builtin_amdgcn_mov_dpp(__double2loint(ai[c]), 0x150 + K, 0xf, 0xf, true));
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {   // a_ij -= m' a_kj for this lane's columns j = 4c+g
        ar[c] = fma(kr[c], nmr, ar[c]);
        ar[c] = fma(ki[c], pmi, ar[c]);
        ai[c] = fma(ki[c], nmr, ai[c]);
        ai[c] = fma(kr[c], nmi, ai[c]);
    }
#endif
}

template <int K>
__device__ __forceinline__ void invert16_steps(double (&ar)[4], double (&ai)[4], const int i, const int g,
                                               double &minrel, double &myqr, double &myqi, double inv_scale2) {
    if constexpr (K < 16) {
        invert16_step<K>(ar, ai, i, g, minrel, myqr, myqi, inv_scale2);
        invert16_steps<K + 1>(ar, ai, i, g, minrel, myqr, myqi, inv_scale2);
    }
}

__device__ __forceinline__ double invert16(double (&ar)[4], double (&ai)[4], int lane, double inv_scale2) {
    const int i = lane & 15, g = lane >> 4;
    double minrel = 1e300;
    double myqr = 1.0, myqi = 0.0;  // 1/pivot of this lane's row, applied once at the end
    // Gauss-Jordan with UNSCALED pivot rows: step k only touches rows i != k,
    //   a_ij -= (a_ik / p_k) a_kj  (j != k),   a_ik = -(a_ik / p_k),   a_kk = 1,
    // and every row is divided by its own pivot afterwards.  Per step: one complex reciprocal,
    // one complex multiply and the 4 complex FMAs of this lane's columns.
    invert16_steps<0>(ar, ai, i, g, minrel, myqr, myqi, inv_scale2);
#pragma unroll
    for (int c = 0; c < 4; ++c) {   // row i /= p_i
        const double r = fma(ar[c], myqr, -ai[c] * myqi);
        ai[c] = fma(ar[c], myqi, ai[c] * myqr);
        ar[c] = r;
    }
    return minrel;
}

// Block Gauss-Jordan solve  Q X = P  on register strips (wave w owns column strip w of Q and P),
// NT block steps.  Step jb needs the panel (block column jb of the current Q) and the inverse of
// its 16x16 diagonal tile; every wave then updates its strips with MFMA:
//     Y      = Dinv * S[jb]             (16x16x16)
//     S[tr] -= Panel[tr] * Y  (tr != jb),   S[jb] = Y
// After NT steps Q == I and P == X.
// The register inversion of a diagonal tile is serial (one wave, 16 dependent pivot steps, ~6.5 K cycles): the two
// schedules below (block_gj_solve: split steps, NT <= 3; block_gj_solve_lookahead: NT = 4) differ in how they keep the
// other waves busy meanwhile.  Panels and inverses live in 3 rotating LDS slots.
#ifdef GRAPE_DIAG
__device__ unsigned long long *g_diag_slot_base = nullptr;
#define g_diag_slot (g_diag_slot_base ? g_diag_slot_base + (size_t)blockIdx.x * 32 + 10 : nullptr)
// stamps of the persistent kernel are taken in the SECOND cell of each workgroup (steady state: its A was prefetched
// and it prefetches the next one); g_diag_off[block] != 0 switches the stamps off
__device__ volatile int g_diag_off[1024];
#define STAMP(i) do { if (threadIdx.x == 0 && g_diag_slot_base && !g_diag_off[blockIdx.x & 1023]) g_diag_slot_base[(size_t)blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
// per-wave stamp (diagnostic builds): when a wave has finished its work of a solve step
#ifdef GRAPE_DIAG
#define WSTAMP(i) do { if ((threadIdx.x & 63) == 0 && g_diag_slot_base && !g_diag_off[blockIdx.x & 1023]) g_diag_slot_base[(size_t)blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSTAMP(i) do {} while (0)
#endif
#define GJSTAMP(i) STAMP(i)
template <int NT>
struct GjLds {
    static constexpr int NP = 16 * NT, PLD = 18;
    static constexpr int PAN = 2 * NP * PLD;  // doubles per panel slot (re plane, im plane)
    static constexpr int DV = 512;            // doubles per inverse slot
};

template <int NT>
__device__ __forceinline__ void gj_publish_invert(const Strip<NT> &Q, int jb, double *pan, double *dv, int lane,
                                                  double &minrel, double inv_scale2, bool do_invert) {
    constexpr int NP = 16 * NT, PLD = 18;
    const int ai = lane & 15, ak = lane >> 4;
    double *pwr = pan + ak * PLD + ai, *pwi = pan + NP * PLD + ak * PLD + ai;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pwr[(16 * t + 4 * r) * PLD] = Q.re[t][r];
            pwi[(16 * t + 4 * r) * PLD] = Q.im[t][r];
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // gather the diagonal tile in inversion layout: D[i][4c+g], i = ai, g = ak
    double dr[4], di[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        dr[c] = pan[(16 * jb + ai) * PLD + 4 * c + ak];
        di[c] = pan[NP * PLD + (16 * jb + ai) * PLD + 4 * c + ak];
    }
    double mr = 1.0;
#ifdef GRAPE_DIAG
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
    if (do_invert) mr = invert16(dr, di, lane, inv_scale2);
#ifdef GRAPE_DIAG
    if (lane == 0 && g_diag_slot) atomicAdd(g_diag_slot, __builtin_amdgcn_s_memtime() - t0_);
#endif
    minrel = fmin(minrel, mr);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        dv[c * 64 + lane] = dr[c];
        dv[256 + c * 64 + lane] = di[c];
    }
}

// the two halves of gj_publish_invert for the split schedule: the owner of strip jb publishes its panel, another
// wave gathers the diagonal tile from the panel, inverts it and publishes the inverse
template <int NT>
__device__ __forceinline__ void gj_publish(const Strip<NT> &Q, double *pan, int lane) {
    constexpr int NP = 16 * NT, PLD = 18;
    const int ai = lane & 15, ak = lane >> 4;
    double *pwr = pan + ak * PLD + ai, *pwi = pan + NP * PLD + ak * PLD + ai;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pwr[(16 * t + 4 * r) * PLD] = Q.re[t][r];
            pwi[(16 * t + 4 * r) * PLD] = Q.im[t][r];
        }
}
template <int NT>
__device__ __forceinline__ void gj_invert_panel(int jb, const double *pan, double *dv, int lane, double &minrel,
                                                double inv_scale2) {
    constexpr int NP = 16 * NT, PLD = 18;
    const int ai = lane & 15, ak = lane >> 4;
    double dr[4], di[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        dr[c] = pan[(16 * jb + ai) * PLD + 4 * c + ak];
        di[c] = pan[NP * PLD + (16 * jb + ai) * PLD + 4 * c + ak];
    }
    const double mr = invert16(dr, di, lane, inv_scale2);
    minrel = fmin(minrel, mr);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        dv[c * 64 + lane] = dr[c];
        dv[256 + c * 64 + lane] = di[c];
    }
}

template <int NT>
__device__ __forceinline__ void gj_update(Strip<NT> &S, int jb, const double *pan, const double *dv, int lane) {
    constexpr int NP = 16 * NT, PLD = 18;
    const int ai = lane & 15, ak = lane >> 4;
    const double *par = pan + ai * PLD + ak, *pai = pan + NP * PLD + ai * PLD + ak;
    double dvr[4], dvi[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        dvr[c] = dv[c * 64 + lane];
        dvi[c] = dv[256 + c * 64 + lane];
    }
    // all panel operands of this update are requested up front: their LDS latency hides behind the MFMAs of
    // Y = Dinv * S[jb] (the solve phase has registers to spare, unlike the polynomial phase)
    double pre[4][NT], pim[4][NT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int tr = 0; tr < NT; ++tr) {
            if (tr == jb) continue;
            pre[c][tr] = par[16 * tr * PLD + 4 * c];
            pim[c][tr] = pai[16 * tr * PLD + 4 * c];
        }
    // both products by the 3M scheme (see gemm_xb): 12 + 12 (NT - 1) instead of 16 + 16 (NT - 1) MFMAs
    d4 yr = {0., 0., 0., 0.}, yi = {0., 0., 0., 0.};
    {
        d4 y1 = {0., 0., 0., 0.}, y2 = {0., 0., 0., 0.}, y3 = {0., 0., 0., 0.};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t != jb) continue;   // jb is a compile-time constant after unrolling
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double bre = S.re[t][c], bim = S.im[t][c];
                y1 = MFMA64(dvr[c], bre, y1);
                y2 = MFMA64(dvi[c], bim, y2);
                y3 = MFMA64(dvr[c] + dvi[c], bre + bim, y3);
            }
        }
        yr = y1 - y2;
        yi = y3 - y1 - y2;
    }
    d4 q1[NT], q2[NT], q3[NT];
#pragma unroll
    for (int tr = 0; tr < NT; ++tr) { q1[tr] = (d4){0., 0., 0., 0.}; q2[tr] = (d4){0., 0., 0., 0.}; q3[tr] = (d4){0., 0., 0., 0.}; }
    // row tile by row tile (every accumulation chain still runs over the k-steps c = 0..3 in order): the sums of the
    // three partial products of one tile are vector work that runs under the matrix instructions of the next tile --
    // with the k-step outermost all chains end together and the whole tail (a third of the update's time) is exposed
    double bs[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bs[c] = yr[c] + yi[c];
#pragma unroll
    for (int tr = 0; tr < NT; ++tr) {
        if (tr == jb) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // k-step c: k = 4c + ak within the panel's 16 columns
            const double are = pre[c][tr], aim = pim[c][tr];
            q1[tr] = MFMA64(are, yr[c], q1[tr]);
            q2[tr] = MFMA64(aim, yi[c], q2[tr]);
            q3[tr] = MFMA64(are + aim, bs[c], q3[tr]);
        }
        S.re[tr] -= q1[tr] - q2[tr];
        S.im[tr] -= q3[tr] - q1[tr] - q2[tr];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t == jb) { S.re[t] = yr; S.im[t] = yi; }
}

// the Q- and the P-strip update of one block step in one pass: both strips multiply the same panel and the same
// inverse, so the A operands (LDS loads, the 3M sums) are formed once, and the two sets of accumulation chains
// interleave.  Every chain has the operation order of gj_update: the results are bit-identical.
template <int NT>
__device__ __forceinline__ void gj_update2(Strip<NT> &S, Strip<NT> &T, int jb, const double *pan, const double *dv, int lane) {
    constexpr int NP = 16 * NT, PLD = 18;
    const int ai = lane & 15, ak = lane >> 4;
    const double *par = pan + ai * PLD + ak, *pai = pan + NP * PLD + ai * PLD + ak;
    double dvr[4], dvi[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        dvr[c] = dv[c * 64 + lane];
        dvi[c] = dv[256 + c * 64 + lane];
    }
    d4 sr, si, tr_, ti_;
    {
        d4 y1 = {0., 0., 0., 0.}, y2 = {0., 0., 0., 0.}, y3 = {0., 0., 0., 0.};
        d4 z1 = {0., 0., 0., 0.}, z2 = {0., 0., 0., 0.}, z3 = {0., 0., 0., 0.};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t != jb) continue;   // jb is a compile-time constant after unrolling
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double are = dvr[c], aim = dvi[c], as = dvr[c] + dvi[c];
                y1 = MFMA64(are, S.re[t][c], y1);
                z1 = MFMA64(are, T.re[t][c], z1);
                y2 = MFMA64(aim, S.im[t][c], y2);
                z2 = MFMA64(aim, T.im[t][c], z2);
                y3 = MFMA64(as, S.re[t][c] + S.im[t][c], y3);
                z3 = MFMA64(as, T.re[t][c] + T.im[t][c], z3);
            }
        }
        sr = y1 - y2;
        si = y3 - y1 - y2;
        tr_ = z1 - z2;
        ti_ = z3 - z1 - z2;
    }
    d4 q1[NT], q2[NT], q3[NT], r1[NT], r2[NT], r3[NT];
#pragma unroll
    for (int tr = 0; tr < NT; ++tr) {
        q1[tr] = (d4){0., 0., 0., 0.}; q2[tr] = (d4){0., 0., 0., 0.}; q3[tr] = (d4){0., 0., 0., 0.};
        r1[tr] = (d4){0., 0., 0., 0.}; r2[tr] = (d4){0., 0., 0., 0.}; r3[tr] = (d4){0., 0., 0., 0.};
    }
    double bs[4], bt[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { bs[c] = sr[c] + si[c]; bt[c] = tr_[c] + ti_[c]; }
#pragma unroll
    for (int tr = 0; tr < NT; ++tr) {   // row tile by row tile, see gj_update
        if (tr == jb) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double are = par[16 * tr * PLD + 4 * c], aim = pai[16 * tr * PLD + 4 * c], as = are + aim;
            q1[tr] = MFMA64(are, sr[c], q1[tr]);
            r1[tr] = MFMA64(are, tr_[c], r1[tr]);
            q2[tr] = MFMA64(aim, si[c], q2[tr]);
            r2[tr] = MFMA64(aim, ti_[c], r2[tr]);
            q3[tr] = MFMA64(as, bs[c], q3[tr]);
            r3[tr] = MFMA64(as, bt[c], r3[tr]);
        }
        S.re[tr] -= q1[tr] - q2[tr];
        S.im[tr] -= q3[tr] - q1[tr] - q2[tr];
        T.re[tr] -= r1[tr] - r2[tr];
        T.im[tr] -= r3[tr] - r1[tr] - r2[tr];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t == jb) { S.re[t] = sr; S.im[t] = si; T.re[t] = tr_; T.im[t] = ti_; }
}

struct GjNoHook {
    __device__ __forceinline__ void operator()(int) const {}
    __device__ __forceinline__ void first_inversion_idle() const {}
};

// `hook(jb)` runs in every wave after its own work of block step jb and before the barrier that ends the step:
// the place for work that is independent of the solve (the waves that do not invert are otherwise waiting).
// `hook.first_inversion_idle()` runs in the waves 1..NT-1 while wave 0 inverts the first diagonal tile: the one stretch
// of the solve in which they have nothing else to do (6-7 K cycles).
template <int NT, class Hook = GjNoHook>
__device__ __forceinline__ void block_gj_solve(Strip<NT> &Q, Strip<NT> &P, double *panbase, double *dvbase,
                                               int wave, int lane, double &minrel, double inv_scale2,
                                               bool do_invert, const Hook &hook = Hook()) {
    constexpr int PAN = GjLds<NT>::PAN, DV = GjLds<NT>::DV;
    __syncthreads();  // previous users of the staging region are done
    if (wave == 0) gj_publish_invert<NT>(Q, 0, panbase, dvbase, lane, minrel, inv_scale2, do_invert);
    else hook.first_inversion_idle();
    __syncthreads();
    STAMP(5);
    // SPLIT SCHEDULE.  The register inversion of the next diagonal tile is serial (one wave, 16 dependent pivot
    // steps, ~6 K cycles) and each strip update is ~5 K cycles of MFMAs.  Block step jb runs in two halves:
    //   first half : every wave does ONE update -- the owner of strip jb+1 and the waves behind it update their
    //                Q strip (the owner then publishes the next panel), the waves whose Q strip is finished
    //                (w <= jb) update their P strip;
    //   second half: wave jb (its Q strip is finished, its P strip is up to date) inverts the next diagonal
    //                tile out of the published panel while the waves w > jb update their P strips.
    // A step therefore costs max(update) + max(inversion, update) instead of update + inversion + update for
    // the owner, and nobody carries a deferred update into the last step.
#pragma unroll
    for (int jb = 0; jb < NT; ++jb) {
        const double *pan = panbase + (jb % 3) * PAN, *dv = dvbase + (jb % 3) * DV;
        double *pan_next = panbase + ((jb + 1) % 3) * PAN, *dv_next = dvbase + ((jb + 1) % 3) * DV;
        if (wave > jb) {
            gj_update<NT>(Q, jb, pan, dv, lane);
            if (jb + 1 < NT && wave == jb + 1) gj_publish<NT>(Q, pan_next, lane);
        } else {
            gj_update<NT>(P, jb, pan, dv, lane);
        }
        if (jb + 1 < NT) {
            __syncthreads();
            if (wave == jb) {
                if (do_invert) gj_invert_panel<NT>(jb + 1, pan_next, dv_next, lane, minrel, inv_scale2);
            } else if (wave > jb) {
                gj_update<NT>(P, jb, pan, dv, lane);
            }
        }
        hook(jb);
        __syncthreads();
        STAMP(6 + jb);
    }
}

// ---------------------------------------------------------------------------------------
// LOOK-AHEAD SCHEDULE of the block Gauss-Jordan solve (NT = 4, one barrier per block step, no flags).
// What bounds the split schedule above is the chain  owner's strip update -> barrier -> tile inversion -> barrier
// (12.7 K cycles per step, of which a wave issues MFMAs for 5-10 K).  The next diagonal tile,
//     D'_{jb+1} = Q[jb+1][jb+1] - Panel_jb[jb+1] (Dinv_jb Q[jb][jb+1]),
// only needs two tiles of strip jb+1 as they stand BEFORE step jb, the panel and the inverse of step jb: all of it is
// published before the barrier that starts the step (the owner of strip jb+1 stores its two tiles one step ahead, 16 LDS
// stores).  So the wave that inverts does the 24 MFMAs of that tile update itself -- in exactly the operation order of
// gj_update, the tile is bit-identical to the one the owner computes -- inverts it and publishes the inverse while the
// other waves issue the MFMAs of their two strip updates; nobody waits for anybody inside a step.  The inverting wave
// is wave jb, whose Q strip is finished; its own P-strip update of the step is deferred to the next step, where that
// wave has one update less than the others anyway (panels and inverses rotate through 3 slots, so the operands of step
// jb are intact during step jb+1).  A step costs two strip updates (measured 4 x 9.8 K cycles instead of
// 3 x 12.7 + 5.1 K; the inverting wave needs 9.1 K).  Every wave runs its own straight-line program (W is a template
// parameter): with run-time branches on the wave index the register allocator moves strips between vector and
// accumulation registers at every merge point.
// ---------------------------------------------------------------------------------------
// two row tiles (t0, t1) of this wave's strip -> look-ahead area [tile][plane][16][GJ_LAP]
constexpr int GJ_LAP = 18;                  // row stride of a look-ahead tile (doubles)
constexpr int GJ_LAPL = 16 * GJ_LAP;        // one plane of a tile
constexpr int GJ_LA_SIZE = 4 * GJ_LAPL;     // two tiles, two planes
template <int NT>
__device__ __forceinline__ void gj_publish_tiles(const Strip<NT> &Q, int t0, int t1, double *la, int lane) {
    const int ai = lane & 15, ak = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t != t0 && t != t1) continue;
        double *dr = la + (t == t0 ? 0 : 2 * GJ_LAPL) + ak * GJ_LAP + ai, *di = dr + GJ_LAPL;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dr[4 * r * GJ_LAP] = Q.re[t][r];   // element [4r + ak][ai] of the tile
            di[4 * r * GJ_LAP] = Q.im[t][r];
        }
    }
}

// the inverting wave: D' = tile1 - Panel[jn] (Dinv tile0) with the operation order of gj_update, then its inverse
template <int NT>
__device__ __forceinline__ void gj_lookahead_invert(int jn, const double *pan, const double *dv, double *la, double *dv_next,
                                                    int lane, double &minrel, double inv_scale2) {
    constexpr int NP = 16 * NT, PLD = 18;
    const int ai = lane & 15, ak = lane >> 4;
    const double *par = pan + ai * PLD + ak + 16 * jn * PLD, *pai = par + NP * PLD;
    double dvr[4], dvi[4], pre[4], pim[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        dvr[c] = dv[c * 64 + lane];
        dvi[c] = dv[256 + c * 64 + lane];
        pre[c] = par[4 * c];
        pim[c] = pai[4 * c];
    }
    // tile0 as the B operand (k-step c: element [4c + ak][ai]); tile1 in the accumulator layout (register r: [4r + ak][ai])
    const double *t0r = la + ak * GJ_LAP + ai, *t0i = t0r + GJ_LAPL, *t1r = t0r + 2 * GJ_LAPL, *t1i = t0r + 3 * GJ_LAPL;
    d4 y1 = {0., 0., 0., 0.}, y2 = {0., 0., 0., 0.}, y3 = {0., 0., 0., 0.};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double bre = t0r[4 * c * GJ_LAP], bim = t0i[4 * c * GJ_LAP];
        y1 = MFMA64(dvr[c], bre, y1);
        y2 = MFMA64(dvi[c], bim, y2);
        y3 = MFMA64(dvr[c] + dvi[c], bre + bim, y3);
    }
    const d4 yr = y1 - y2, yi = y3 - y1 - y2;
    d4 q1 = {0., 0., 0., 0.}, q2 = {0., 0., 0., 0.}, q3 = {0., 0., 0., 0.};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double bre = yr[c], bim = yi[c], bs = yr[c] + yi[c];
        q1 = MFMA64(pre[c], bre, q1);
        q2 = MFMA64(pim[c], bim, q2);
        q3 = MFMA64(pre[c] + pim[c], bs, q3);
    }
    d4 dre, dim;
#pragma unroll
    for (int r = 0; r < 4; ++r) { dre[r] = t1r[4 * r * GJ_LAP]; dim[r] = t1i[4 * r * GJ_LAP]; }
    dre -= q1 - q2;
    dim -= q3 - q1 - q2;
    // accumulator layout -> inversion layout through the (consumed) tile1 area: only this wave touches it
    double *s_r = la + 2 * GJ_LAPL, *s_i = la + 3 * GJ_LAPL;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s_r[(4 * r + ak) * GJ_LAP + ai] = dre[r];
        s_i[(4 * r + ak) * GJ_LAP + ai] = dim[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    double dr[4], di[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {   // D[i][4c + g], i = ai, g = ak
        dr[c] = s_r[ai * GJ_LAP + 4 * c + ak];
        di[c] = s_i[ai * GJ_LAP + 4 * c + ak];
    }
    const double mr = invert16(dr, di, lane, inv_scale2);
    minrel = fmin(minrel, mr);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        dv_next[c * 64 + lane] = dr[c];
        dv_next[256 + c * 64 + lane] = di[c];
    }
}

// a barrier the instruction scheduler may not move matrix instructions across: it otherwise sinks the tail of a
// step's updates below the barrier, in front of the next step's tile inversion (measured: 44 K -> 39 K cycles per solve)
#define GJ_SYNC() do { __builtin_amdgcn_sched_barrier(0); __syncthreads(); __builtin_amdgcn_sched_barrier(0); } while (0)
template <int NT, int W, bool ROT, class Hook>
__device__ __forceinline__ void gj_lookahead_program(Strip<NT> &Qio, Strip<NT> &Pio, double *panbase, double *dvbase,
                                                     double *la0, double *la1, int lane, double &minrel,
                                                     double inv_scale2, const Hook &hook) {
    static_assert(NT >= 2 && W < NT, "one program per strip owner");
    // ROT: the strips arrive rotated (slot s = row tile (W + s) % NT); W is a compile-time constant here, so taking them
    // into natural order is a renaming of registers.  P leaves in natural order.
    Strip<NT> Q, P;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int src = ROT ? (t - W + NT) % NT : t;
        Q.re[t] = Qio.re[src]; Q.im[t] = Qio.im[src];
        P.re[t] = Pio.re[src]; P.im[t] = Pio.im[src];
    }
    constexpr int PAN = GjLds<NT>::PAN, DV = GjLds<NT>::DV;
    GJ_SYNC();  // previous users of the staging region are done
    STAMP(22);
    if (W == 0) gj_publish_invert<NT>(Q, 0, panbase, dvbase, lane, minrel, inv_scale2, true);
    else {
        if (W == 1) gj_publish_tiles<NT>(Q, 0, 1, la0, lane);   // what wave 0 needs for D'_1 in step 0
        hook.first_inversion_idle();
    }
    if (NT == 4 && W < 3) { __builtin_amdgcn_sched_barrier(0); WSTAMP(29 + W); }
    GJ_SYNC();
    STAMP(5);
#pragma unroll
    for (int jb = 0; jb < NT; ++jb) {
        const double *pan = panbase + (jb % 3) * PAN, *dv = dvbase + (jb % 3) * DV;
        const double *pan_prev = panbase + ((jb + 2) % 3) * PAN, *dv_prev = dvbase + ((jb + 2) % 3) * DV;   // step jb-1
        double *pan_next = panbase + ((jb + 1) % 3) * PAN, *dv_next = dvbase + ((jb + 1) % 3) * DV;
        double *la_cur = (jb & 1) ? la1 : la0, *la_nxt = (jb & 1) ? la0 : la1;
        if (jb >= 1 && W == jb - 1) {   // the P update this wave deferred while it inverted in the previous step
            gj_update<NT>(P, jb - 1, pan_prev, dv_prev, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (jb + 1 < NT && W == jb) {
            gj_lookahead_invert<NT>(jb + 1, pan, dv, la_cur, dv_next, lane, minrel, inv_scale2);
        } else {
            if (W > jb) {
                gj_update2<NT>(Q, P, jb, pan, dv, lane);
                if (W == jb + 1) gj_publish<NT>(Q, pan_next, lane);
                if (W == jb + 2) gj_publish_tiles<NT>(Q, jb + 1, jb + 2, la_nxt, lane);
            } else {
                gj_update<NT>(P, jb, pan, dv, lane);
            }
        }
        hook(jb);
        if (NT == 4 && jb < 1) { __builtin_amdgcn_sched_barrier(0); WSTAMP(18 + 4 * jb + W); }
        GJ_SYNC();
        STAMP(6 + jb);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) { Pio.re[t] = P.re[t]; Pio.im[t] = P.im[t]; }
}

template <int NT, int W, bool ROT, class Hook>
__device__ __forceinline__ void gj_lookahead_dispatch(Strip<NT> &Q, Strip<NT> &P, double *panbase, double *dvbase,
                                                      double *la0, double *la1, int wave, int lane, double &minrel,
                                                      double inv_scale2, const Hook &hook) {
    // wave-uniform (scalar) branches: each wave's path through the solve is straight-line code
    if constexpr (W + 1 < NT) {
        if (wave != W) {
            gj_lookahead_dispatch<NT, W + 1, ROT>(Q, P, panbase, dvbase, la0, la1, wave, lane, minrel, inv_scale2, hook);
            return;
        }
    }
    gj_lookahead_program<NT, W, ROT>(Q, P, panbase, dvbase, la0, la1, lane, minrel, inv_scale2, hook);
}
template <int NT, bool ROT = false, class Hook = GjNoHook>
__device__ __forceinline__ void block_gj_solve_lookahead(Strip<NT> &Q, Strip<NT> &P, double *panbase, double *dvbase,
                                                         double *la0, double *la1, int wave, int lane, double &minrel,
                                                         double inv_scale2, const Hook &hook = Hook()) {
    gj_lookahead_dispatch<NT, 0, ROT>(Q, P, panbase, dvbase, la0, la1, wave, lane, minrel, inv_scale2, hook);
}

// Robust fallback: Q X = P by Gaussian elimination with partial pivoting (LAPACK gesv semantics),
// both matrices in LDS (planar, leading dimension LD), all NTH threads.  X overwrites P.
// `scr` needs 2 * NP + 8 doubles.  Returns false when a pivot is exactly zero.  Slow (hundreds of
// barriers) by design: it only runs for cells whose unpivoted elimination was flagged.
template <int NP, int LD, int NTH>
__device__ __forceinline__ bool pivoted_solve_lds(double *Qre, double *Qim, double *Pre, double *Pim, double *scr, int tid) {
    double *mr = scr, *mi = scr + NP;   // multipliers of the current step
    int *piv = (int *)(scr + 2 * NP);
    bool ok = true;
    for (int k = 0; k < NP; ++k) {
        if (tid == 0) {   // pivot search down column k
            int p = k;
            double best = Qre[k * LD + k] * Qre[k * LD + k] + Qim[k * LD + k] * Qim[k * LD + k];
            for (int i = k + 1; i < NP; ++i) {
                const double v = Qre[i * LD + k] * Qre[i * LD + k] + Qim[i * LD + k] * Qim[i * LD + k];
                if (v > best) { best = v; p = i; }
            }
            piv[0] = p;
            piv[1] = best > 0. ? 1 : 0;
        }
        __syncthreads();
        const int p = piv[0];
        if (!piv[1]) ok = false;
        if (p != k) {
            for (int j = tid; j < 2 * NP; j += NTH) {
                double *re = j < NP ? Qre : Pre, *im = j < NP ? Qim : Pim;
                const int c = j < NP ? j : j - NP;
                const double tr_ = re[k * LD + c], ti_ = im[k * LD + c];
                re[k * LD + c] = re[p * LD + c]; im[k * LD + c] = im[p * LD + c];
                re[p * LD + c] = tr_; im[p * LD + c] = ti_;
            }
            __syncthreads();
        }
        {   // multipliers m_i = q_ik / q_kk
            const double pr = Qre[k * LD + k], pi = Qim[k * LD + k];
            const double den = pr * pr + pi * pi, inv = den > 0. ? 1.0 / den : 0.;
            for (int i = k + 1 + tid; i < NP; i += NTH) {
                const double ar = Qre[i * LD + k], ai = Qim[i * LD + k];
                mr[i] = (ar * pr + ai * pi) * inv;
                mi[i] = (ai * pr - ar * pi) * inv;
            }
        }
        __syncthreads();
        // row_i -= m_i * row_k over the remaining columns of Q and all columns of P
        const int ncol = (NP - 1 - k) + NP;
        for (int idx = tid; idx < (NP - 1 - k) * ncol; idx += NTH) {
            const int i = k + 1 + idx / ncol, cc = idx % ncol;
            double *re = cc < NP - 1 - k ? Qre : Pre, *im = cc < NP - 1 - k ? Qim : Pim;
            const int c = cc < NP - 1 - k ? k + 1 + cc : cc - (NP - 1 - k);
            const double xr = re[k * LD + c], xi = im[k * LD + c];
            re[i * LD + c] -= mr[i] * xr - mi[i] * xi;
            im[i * LD + c] -= mr[i] * xi + mi[i] * xr;
        }
        __syncthreads();
    }
    // back substitution, one thread per right-hand-side column
    for (int j = tid; j < NP; j += NTH) {
        for (int k = NP - 1; k >= 0; --k) {
            double sr = Pre[k * LD + j], si = Pim[k * LD + j];
            for (int i = k + 1; i < NP; ++i) {
                const double qr = Qre[k * LD + i], qi = Qim[k * LD + i];
                const double xr = Pre[i * LD + j], xi = Pim[i * LD + j];
                sr -= qr * xr - qi * xi;
                si -= qr * xi + qi * xr;
            }
            const double pr = Qre[k * LD + k], pi = Qim[k * LD + k];
            const double den = pr * pr + pi * pi, inv = den > 0. ? 1.0 / den : 0.;
            Pre[k * LD + j] = (sr * pr + si * pi) * inv;
            Pim[k * LD + j] = (si * pr - sr * pi) * inv;
        }
    }
    __syncthreads();
    return ok;
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    // bijective XCD-aware remap: blocks that share an XCD (bid % 8) get a contiguous run of cells
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// ---------------------------------------------------------------------------------------
// Kernel 1: U_kn = exp(-i H_kn dt_n) for every cell, scaling-and-squaring Pade (orders
// 3/5/7/9/13 selected by ||A||_1 exactly as Julia's exp!), complex fp64 on
// v_mfma_f64_16x16x4_f64.  One workgroup of NT waves per cell; every NP x NP matrix lives
// in registers as column strips, the left GEMM operand is staged in LDS.
// Replaces the `exp` inside ExpProp's prop_step! (optimize.jl:732, 881, 972).
// ---------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------
// Hermitian generators (NT = 4): A = -i dt H is skew-Hermitian, so A^2, A^4, A^6 and the even polynomials
// T, V of the order-13 approximant are Hermitian and U = A T is skew-Hermitian.  A column strip then needs
// only three of its four row tiles from the MFMAs; the fourth is the (signed) conjugate transpose of a tile
// the neighbouring wave computes anyway.  To keep the code identical for all waves the strips are held
// ROTATED: slot s of wave w is row tile (w + s) & 3, so that every wave computes slots 0, 1, 2 (diagonal
// tile and the two below it, cyclically) and receives slot 3 = tile (w-1, w) from wave w-1's slot 1 =
// tile (w, w-1).  The k loop of a product runs in the same rotated order (register r of slot s is the B
// operand of k-step 16 ((w+s)&3) + 4r); only LDS addresses depend on the wave, through scalar offsets.
// 12 instead of 16 MFMAs per k-step: the six products of the approximant cost 4.5.
// ---------------------------------------------------------------------------------------
template <int LD, int NT = 4>
__device__ __forceinline__ void rot_load_strip(const double *Xre, const double *Xim, Strip<NT> &S, int wave, int lane) {
    const double *xr = Xre + (lane >> 4) * LD + 16 * wave + (lane & 15);
    const double *xi = Xim + (lane >> 4) * LD + 16 * wave + (lane & 15);
#pragma unroll
    for (int sl = 0; sl < NT; ++sl) {
        const int tb = (wave + sl) % NT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            S.re[sl][r] = xr[(16 * tb + 4 * r) * LD];
            S.im[sl][r] = xi[(16 * tb + 4 * r) * LD];
        }
    }
}
// slots 0..NS-1 to their natural plane positions
template <int LD, int NS, int NT = 4>
__device__ __forceinline__ void rot_store_slots(double *Xre, double *Xim, const Strip<NT> &S, int wave, int lane) {
    double *xr = Xre + (lane >> 4) * LD + 16 * wave + (lane & 15);
    double *xi = Xim + (lane >> 4) * LD + 16 * wave + (lane & 15);
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        const int tb = (wave + sl) % NT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xr[(16 * tb + 4 * r) * LD] = S.re[sl][r];
            xi[(16 * tb + 4 * r) * LD] = S.im[sl][r];
        }
    }
}
// (signed) conjugate transpose of the slot-1 tile (row block w+1, column block w) into plane position
// (row block w, column block w+1); sgn = +1 Hermitian, -1 skew-Hermitian
template <int LD, int NT = 4>
__device__ __forceinline__ void rot_store_adjoint(double *Xre, double *Xim, const d4 &tre, const d4 &tim, int wave,
                                                  int lane, double sgn) {
    const int tb = (wave + 1) % NT, c = lane & 15, rg = lane >> 4;
    double *xr = Xre + (16 * wave + c) * LD + 16 * tb + rg;
    double *xi = Xim + (16 * wave + c) * LD + 16 * tb + rg;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        xr[4 * r] = sgn * tre[r];
        xi[4 * r] = -sgn * tim[r];
    }
}
template <int LD, int NT = 4>
__device__ __forceinline__ void rot_load_slot3(const double *Xre, const double *Xim, Strip<NT> &S, int wave, int lane) {
    const int tb = (wave + NT - 1) % NT;
    const double *xr = Xre + ((lane >> 4) + 16 * tb) * LD + 16 * wave + (lane & 15);
    const double *xi = Xim + ((lane >> 4) + 16 * tb) * LD + 16 * wave + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        S.re[NT - 1][r] = xr[4 * r * LD];
        S.im[NT - 1][r] = xi[4 * r * LD];
    }
}
// exchange through a small area (4 waves x 2 planes x 256 doubles) when no plane is free: the writer stores
// its slot-1 tile in the reader's register layout, [lane'][r'] with lane' = 16 (c & 3) + 4r + rg, r' = c >> 2
template <int NT = 4, int DIST = 1>   // DIST: the reader is wave + DIST (1: slot-1 tile -> mirrored slot, 2: see gemm_rot HALF_LAST)
__device__ __forceinline__ void rot_exch_write(double *area, const d4 &tre, const d4 &tim, int wave, int lane, double sgn) {
    const int c = lane & 15, rg = lane >> 4;
    double *dst = area + ((wave + DIST) % NT) * 512 + (16 * (c & 3) + rg) * 4 + (c >> 2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        dst[16 * r] = sgn * tre[r];          // lane' advances by 4 per r: (4r) * 4 doubles
        dst[256 + 16 * r] = -sgn * tim[r];
    }
}
// the tile addressed to this wave, added to slot SLOT
template <int NT, int SLOT>
__device__ __forceinline__ void rot_exch_add(const double *area, Strip<NT> &S, int wave, int lane) {
    const double *src = area + wave * 512 + lane * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        S.re[SLOT][r] += src[r];
        S.im[SLOT][r] += src[256 + r];
    }
}
template <int NT = 4>
__device__ __forceinline__ void rot_exch_read(const double *area, Strip<NT> &S, int wave, int lane) {
    const double *src = area + wave * 512 + lane * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        S.re[NT - 1][r] = src[r];
        S.im[NT - 1][r] = src[256 + r];
    }
}

// acc[0..NS-1] += X * B for rotated strips (X in LDS planes, natural layout).
// Complex product by the 3M scheme: P1 = Xr Br, P2 = Xi Bi, P3 = (Xr + Xi)(Br + Bi); re = P1 - P2,
// im = P3 - P1 - P2: three real MFMA products per complex one (normwise stable, Higham 1992), 9 instead of 12 MFMAs
// per k-step; the operand sums are one VALU add each against 64-cycle MFMAs.  Software pipeline as in gemm_xb: the
// real-plane operands of k-step ks+1 are requested once the P1 MFMAs have issued, the imaginary-plane operands
// after the P2 MFMAs, so that every LDS read has at least NS MFMAs to land before the next operand sum needs it.
// `late_exch` != nullptr: the mirrored slot NT-1 of B is still on its way through the exchange area (its owner wrote it with
// rot_exch_write just before this call); it is the LAST k-block of the rotated k order, so the barrier and the read wait
// until the first NT-1 k-blocks have been issued -- the exchange costs no time of its own.
// HALF_LAST (NT = 4, squares of a Hermitian matrix, X == B): the last computed slot, tile (w+2, w), is also computed -- as its
// conjugate transpose (w, w+2) -- by wave w+2.  In a square X X the two are adjoint TERM BY TERM,
// (X(i,k) X(k,j))^dagger = X(j,k) X(k,i), so each of the two waves only sums the first two k-blocks of its rotated k order
// (w, w+1 here; w+2, w+3 there) and the caller adds the partner's partial sum, exchanged like the mirrored tiles:
// 120 instead of 144 matrix instructions.
template <int LD, int NS, int NT = 4, bool HALF_LAST = false>
__device__ __forceinline__ void gemm_rot(Strip<NT> &acc, const double *__restrict__ Xre, const double *__restrict__ Xim,
                                         Strip<NT> &B, int wave, int lane, const double *late_exch = nullptr) {
    const double *__restrict__ xr = Xre + (lane & 15) * LD + (lane >> 4);
    const double *__restrict__ xi = Xim + (lane & 15) * LD + (lane >> 4);
    int rowoff[NS];
#pragma unroll
    for (int so = 0; so < NS; ++so) rowoff[so] = 16 * ((wave + so) % NT) * LD;
    double are[NS], aim[NS];
    {
        const int k0 = 16 * wave;
#pragma unroll
        for (int so = 0; so < NS; ++so) { are[so] = xr[rowoff[so] + k0]; aim[so] = xi[rowoff[so] + k0]; }
    }
    d4 p1[NS], p2[NS], p3[NS];
#pragma unroll
    for (int so = 0; so < NS; ++so) { p1[so] = (d4){0., 0., 0., 0.}; p2[so] = (d4){0., 0., 0., 0.}; p3[so] = (d4){0., 0., 0., 0.}; }
#pragma unroll
    for (int sk = 0; sk < NT; ++sk) {
        if (sk == NT - 1 && late_exch) {
            __syncthreads();
            rot_exch_read<NT>(late_exch, B, wave, lane);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kn = (r < 3) ? 16 * ((wave + sk) % NT) + 4 * (r + 1) : 16 * ((wave + sk + 1) % NT);   // next k column
            const bool more = !(sk == NT - 1 && r == 3);
            const double bre = B.re[sk][r], bim = B.im[sk][r];
            const double bs = bre + bim;
            const int ns = (HALF_LAST && 2 * sk >= NT) ? NS - 1 : NS;                                    // slots of this k-step
            const int nsn = (HALF_LAST && (2 * sk >= NT || (2 * (sk + 1) >= NT && r == 3))) ? NS - 1 : NS;   // ... of the next one
            double as[NS];
#pragma unroll
            for (int so = 0; so < NS; ++so) as[so] = are[so] + aim[so];
#pragma unroll
            for (int so = 0; so < NS; ++so)
                if (so < ns) p1[so] = MFMA64(are[so], bre, p1[so]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int so = 0; so < NS; ++so)
                    if (so < nsn) are[so] = xr[rowoff[so] + kn];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int so = 0; so < NS; ++so)
                if (so < ns) p2[so] = MFMA64(aim[so], bim, p2[so]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int so = 0; so < NS; ++so)
                    if (so < nsn) aim[so] = xi[rowoff[so] + kn];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int so = 0; so < NS; ++so)
                if (so < ns) p3[so] = MFMA64(as[so], bs, p3[so]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int so = 0; so < NS; ++so) {
        acc.re[so] += p1[so] - p2[so];
        acc.im[so] += p3[so] - p1[so] - p2[so];
    }
}

// rotated version of gemm_dual13: T[0..2] += X (b13 A6 + b11 A4 + b9 A2),  V[0..2] += X (b12 A6 + b10 A4 + b8 A2),
// both by the 3M scheme (see gemm_rot): 18 instead of 24 MFMAs per k-step, the operand sums of X are shared.
template <int LD, int NT = 4>
__device__ __forceinline__ void gemm_dual13_rot(Strip<NT> &T, Strip<NT> &V, const double *__restrict__ Xre,
                                                const double *__restrict__ Xim, const Strip<NT> &A2, const Strip<NT> &A4,
                                                const Strip<NT> &A6, int wave, int lane) {
    constexpr int NS = NT - 1;
    const double *__restrict__ xr = Xre + (lane & 15) * LD + (lane >> 4);
    const double *__restrict__ xi = Xim + (lane & 15) * LD + (lane >> 4);
    int rowoff[NS];
#pragma unroll
    for (int so = 0; so < NS; ++so) rowoff[so] = 16 * ((wave + so) % NT) * LD;
    d4 t1[NS], t2[NS], t3[NS], v1[NS], v2[NS], v3[NS];
#pragma unroll
    for (int so = 0; so < NS; ++so) {
        t1[so] = (d4){0., 0., 0., 0.}; t2[so] = (d4){0., 0., 0., 0.}; t3[so] = (d4){0., 0., 0., 0.};
        v1[so] = (d4){0., 0., 0., 0.}; v2[so] = (d4){0., 0., 0., 0.}; v3[so] = (d4){0., 0., 0., 0.};
    }
#pragma unroll
    for (int sk = 0; sk < NT; ++sk) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kc = 16 * ((wave + sk) % NT) + 4 * r;
            const double wr = B13_13 * A6.re[sk][r] + B13_11 * A4.re[sk][r] + B13_9 * A2.re[sk][r];
            const double wi = B13_13 * A6.im[sk][r] + B13_11 * A4.im[sk][r] + B13_9 * A2.im[sk][r];
            const double zr = B13_12 * A6.re[sk][r] + B13_10 * A4.re[sk][r] + B13_8 * A2.re[sk][r];
            const double zi = B13_12 * A6.im[sk][r] + B13_10 * A4.im[sk][r] + B13_8 * A2.im[sk][r];
            const double ws = wr + wi, zs = zr + zi;
            double are[NS], aim[NS], as[NS];
#pragma unroll
            for (int so = 0; so < NS; ++so) { are[so] = xr[rowoff[so] + kc]; aim[so] = xi[rowoff[so] + kc]; }
#pragma unroll
            for (int so = 0; so < NS; ++so) {
                t1[so] = MFMA64(are[so], wr, t1[so]);
                v1[so] = MFMA64(are[so], zr, v1[so]);
            }
#pragma unroll
            for (int so = 0; so < NS; ++so) {
                t2[so] = MFMA64(aim[so], wi, t2[so]);
                v2[so] = MFMA64(aim[so], zi, v2[so]);
            }
#pragma unroll
            for (int so = 0; so < NS; ++so) as[so] = are[so] + aim[so];
#pragma unroll
            for (int so = 0; so < NS; ++so) {
                t3[so] = MFMA64(as[so], ws, t3[so]);
                v3[so] = MFMA64(as[so], zs, v3[so]);
            }
        }
    }
#pragma unroll
    for (int so = 0; so < NS; ++so) {
        T.re[so] += t1[so] - t2[so];
        T.im[so] += t3[so] - t1[so] - t2[so];
        V.re[so] += v1[so] - v2[so];
        V.im[so] += v3[so] - v1[so] - v2[so];
    }
}

// LDS carve shared by the device code and the host (expm_lds_bytes): two regions that hold a plane
// pair (A, and the staged left operand X) during the polynomial phase -- the X region doubles as the
// Gauss-Jordan panel slots during the solve -- then the inverse slots and the reduction scratch.
// sqrt for the 1-norm estimate: v_rsq_f64 seed (about 2^-24) and one coupled Newton step, relative error
// about 1e-14.  The norm only selects the Pade order and the number of squarings, so the few ulps saved
// by the full IEEE sequence (three times the instructions) buy nothing.
__device__ __forceinline__ double fast_sqrt(double v) {
    const double y = __builtin_amdgcn_rsq(v);
    double g = v * y;
    const double h = 0.5 * y;
    g = fma(fma(-h, g, 0.5), g, g);
    return v > 0. ? g : 0.;
}

template <int NT>
struct ExpmLds {
    static constexpr int NP = 16 * NT, LD = NP + 2, NTH = NT * 64;
    static constexpr int PLANES = 2 * NP * LD;
    static constexpr int NSLOT = NT < 3 ? NT : 3;
    static constexpr int SLOTS = NSLOT * 2 * NP * 18;
    static constexpr int REG = PLANES > SLOTS ? PLANES : SLOTS;   // doubles per region
    // inverse slots of the Gauss-Jordan solve (512 doubles each, min(NT, 3) in use); NT = 4 also keeps the exchange
    // area of the Hermitian path here (2048).  The small kernels are latency-bound: less LDS = more cells per CU.
    static constexpr int DV = NT < 4 ? NSLOT * 512 : 2048;
    static constexpr int RED = NTH + 8 + NP;
    // look-ahead areas of the solve (two buffers of two padded 16 x 16 tiles, 1152 doubles each).  The per-cell kernels
    // put them into the A region, which is dead during the solve; the persistent kernel (NT = 4) forms the next cell's
    // A there, so it has one buffer of its own and the other one in the tail of the X region behind the three panel
    // slots (REG - SLOTS = 1536 doubles)
    static constexpr int LA = NT == 4 ? 1152 : 0;
    static constexpr int LA0 = 2 * REG + DV + RED, LA1 = REG + SLOTS;   // offsets in the persistent kernel
    static_assert(NT < 2 || REG >= 2304, "the A region holds both look-ahead buffers");
    static constexpr int TOTAL = 2 * REG + DV + RED + LA;         // doubles
};

// A = -i dt (H0_k + sum_l a_l H_l) -> LDS (planar row-major), 16-byte coalesced loads: the element pairs
// idx in [i0, i1) (of NP*NP/2 per plane) are formed by threads t = 0..nt-1.
template <int NT>
__device__ __forceinline__ void expm_form_a(const ExpmArgs &a, const int cell, double *smem, const int t, const int nt,
                                            const int i0, const int i1) {
    using LY = ExpmLds<NT>;
    constexpr int NP = LY::NP, LD = LY::LD;
    double *Are = smem, *Aim = Are + NP * LD;
    const int kc = cell / a.N_T, n = cell - kc * a.N_T;
    const int k = a.rep ? a.rep[kc] : kc;
    const double dt = a.dts[n];
    const double2 *h0 = (const double2 *)(a.H0f + (size_t)k * 2 * NP * NP);
    const double2 *hc = (const double2 *)(a.Hcf + (size_t)(a.hc_per_traj ? k : 0) * a.L * 2 * NP * NP);
    constexpr int HALF = NP * NP / 2;  // double2 elements per plane
    double e[8];
    for (int l = 0; l < a.L; ++l) {
        e[l] = a.eps[(size_t)l * a.N_T + n];
        if (a.shape) e[l] *= a.shape[(size_t)l * a.N_T + n];
    }
    // chunks of 8 element pairs per thread, all 16 loads of one operator in flight at once: the time of this
    // routine is load latency times the number of dependent batches, whatever the number of threads
    for (int base = i0 + t; base < i1; base += 8 * nt) {
        double2 hr[8], hi[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * nt;
            if (idx < i1) { hr[u] = h0[idx]; hi[u] = h0[HALF + idx]; }
        }
        // the first two control operators are requested together with the drift (one latency, not three)
        double2 c0r[8], c0i[8], c1r[8], c1i[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * nt;
            if (idx < i1) { c0r[u] = hc[idx]; c0i[u] = hc[HALF + idx]; }
        }
        if (a.L > 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * nt;
                if (idx < i1) { c1r[u] = hc[(size_t)2 * HALF + idx]; c1i[u] = hc[(size_t)2 * HALF + HALF + idx]; }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            hr[u].x = fma(e[0], c0r[u].x, hr[u].x); hr[u].y = fma(e[0], c0r[u].y, hr[u].y);
            hi[u].x = fma(e[0], c0i[u].x, hi[u].x); hi[u].y = fma(e[0], c0i[u].y, hi[u].y);
        }
        if (a.L > 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                hr[u].x = fma(e[1], c1r[u].x, hr[u].x); hr[u].y = fma(e[1], c1r[u].y, hr[u].y);
                hi[u].x = fma(e[1], c1i[u].x, hi[u].x); hi[u].y = fma(e[1], c1i[u].y, hi[u].y);
            }
        }
        for (int l = 2; l < a.L; ++l) {
            double2 cr[8], ci[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * nt;
                if (idx < i1) { cr[u] = hc[(size_t)l * 2 * HALF + idx]; ci[u] = hc[(size_t)l * 2 * HALF + HALF + idx]; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                hr[u].x = fma(e[l], cr[u].x, hr[u].x); hr[u].y = fma(e[l], cr[u].y, hr[u].y);
                hi[u].x = fma(e[l], ci[u].x, hi[u].x); hi[u].y = fma(e[l], ci[u].y, hi[u].y);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * nt;
            if (idx < i1) {
                const int i = (2 * idx) / NP, j = 2 * idx - i * NP;
                Are[i * LD + j] = dt * hi[u].x;  Are[i * LD + j + 1] = dt * hi[u].y;
                Aim[i * LD + j] = -dt * hr[u].x; Aim[i * LD + j + 1] = -dt * hr[u].y;
            }
        }
    }
}

// Hermitian generators: A = -i dt H is skew-Hermitian, so only the NT (NT + 1) / 2 tiles (16 x 16) on and above the block
// diagonal are fetched (10 of 16 at NP = 64: the routine is bound by what a CU can pull from L2 per cell) and every
// off-diagonal tile is written twice, a_ji = -conj(a_ij).
template <int NTH = 256, int NT = 4>   // NTH threads take part (t = 0..NTH-1): the whole workgroup, or the waves that wait
                                       // for the first tile inversion of the previous cell (persistent kernel)
__device__ __forceinline__ void expm_form_a_herm64(const ExpmArgs &a, const int cell, double *smem, const int t) {
    using LY = ExpmLds<NT>;
    constexpr int NP = LY::NP, LD = LY::LD;
    double *Are = smem, *Aim = Are + NP * LD;
    const int kc = cell / a.N_T, n = cell - kc * a.N_T;
    const int k = a.rep ? a.rep[kc] : kc;
    const double dt = a.dts[n];
    const double2 *h0 = (const double2 *)(a.H0f + (size_t)k * 2 * NP * NP);
    const double2 *hc = (const double2 *)(a.Hcf + (size_t)(a.hc_per_traj ? k : 0) * a.L * 2 * NP * NP);
    constexpr int HALF = NP * NP / 2;  // double2 elements per plane
    double e[8];
    for (int l = 0; l < a.L; ++l) {
        e[l] = a.eps[(size_t)l * a.N_T + n];
        if (a.shape) e[l] *= a.shape[(size_t)l * a.N_T + n];
    }
    // NTILE tiles x 128 element pairs (1280 = 5 per thread at NP = 64 with 256 threads); tile q -> (ti, tj), ti <= tj,
    // row by row
    constexpr int NTILE = NT * (NT + 1) / 2, NPAIR = 128 * NTILE;
    constexpr int NU = (NPAIR + NTH - 1) / NTH;
    int off[NU], ii[NU], jj[NU];
    bool diag[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int ep = min(t + u * NTH, NPAIR - 1), q = ep >> 7, idx = ep & 127;   // (surplus threads repeat the last pair)
        int ti = 0, r = q;
#pragma unroll
        for (int it = 0; it < NT - 1; ++it)
            if (r >= NT - ti) { r -= NT - ti; ++ti; }
        const int tj = ti + r;
        ii[u] = 16 * ti + (idx >> 3);
        jj[u] = 16 * tj + 2 * (idx & 7);
        off[u] = (ii[u] * NP + jj[u]) >> 1;
        diag[u] = ti == tj;
    }
    double2 hr[NU], hi[NU], c0r[NU], c0i[NU], c1r[NU], c1i[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) { hr[u] = h0[off[u]]; hi[u] = h0[HALF + off[u]]; }
#pragma unroll
    for (int u = 0; u < NU; ++u) { c0r[u] = hc[off[u]]; c0i[u] = hc[HALF + off[u]]; }
    if (a.L > 1) {
#pragma unroll
        for (int u = 0; u < NU; ++u) { c1r[u] = hc[(size_t)2 * HALF + off[u]]; c1i[u] = hc[(size_t)2 * HALF + HALF + off[u]]; }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        hr[u].x = fma(e[0], c0r[u].x, hr[u].x); hr[u].y = fma(e[0], c0r[u].y, hr[u].y);
        hi[u].x = fma(e[0], c0i[u].x, hi[u].x); hi[u].y = fma(e[0], c0i[u].y, hi[u].y);
    }
    if (a.L > 1) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            hr[u].x = fma(e[1], c1r[u].x, hr[u].x); hr[u].y = fma(e[1], c1r[u].y, hr[u].y);
            hi[u].x = fma(e[1], c1i[u].x, hi[u].x); hi[u].y = fma(e[1], c1i[u].y, hi[u].y);
        }
    }
    for (int l = 2; l < a.L; ++l) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const double2 cr = hc[(size_t)l * 2 * HALF + off[u]], ci = hc[(size_t)l * 2 * HALF + HALF + off[u]];
            hr[u].x = fma(e[l], cr.x, hr[u].x); hr[u].y = fma(e[l], cr.y, hr[u].y);
            hi[u].x = fma(e[l], ci.x, hi[u].x); hi[u].y = fma(e[l], ci.y, hi[u].y);
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i = ii[u], j = jj[u];
        const double ar0 = dt * hi[u].x, ar1 = dt * hi[u].y, ai0 = -dt * hr[u].x, ai1 = -dt * hr[u].y;
        Are[i * LD + j] = ar0;  Are[i * LD + j + 1] = ar1;
        Aim[i * LD + j] = ai0;  Aim[i * LD + j + 1] = ai1;
        if (!diag[u]) {   // mirrored tile: a_ji = -conj(a_ij)
            Are[j * LD + i] = -ar0;  Are[(j + 1) * LD + i] = -ar1;
            Aim[j * LD + i] = ai0;   Aim[(j + 1) * LD + i] = ai1;
        }
    }
}

// partial column sums of |a_ij| for ||A||_1 = max_j sum_i |a_ij|: thread t of NP * parts threads covers column
// t % NP over the rows i = t / NP (mod parts); red[t] receives the partial sum
template <int NT>
__device__ __forceinline__ void expm_norm_partial(double *smem, const int t, const int parts) {
    using LY = ExpmLds<NT>;
    constexpr int NP = LY::NP, LD = LY::LD;
    const double *Are = smem, *Aim = Are + NP * LD;
    double *red = smem + 2 * LY::REG + LY::DV;
    const int j = t % NP, part = t / NP;
    double sum = 0.;
    for (int i = part; i < NP; i += parts) {
        const double xr = Are[i * LD + j], xi = Aim[i * LD + j];
        sum += fast_sqrt(xr * xr + xi * xi);
    }
    red[t] = sum;
}

// first wave: column sums of the partials, then the maximum over the columns (wavefront shuffles) -> red[NTH]
template <int NT>
__device__ __forceinline__ void expm_norm_combine(double *smem, const int tid, const int parts) {
    using LY = ExpmLds<NT>;
    constexpr int NP = LY::NP, NTH = LY::NTH;
    double *red = smem + 2 * LY::REG + LY::DV;
    if (tid < 64) {
        double c = 0.;
        if (tid < NP)
            for (int p = 0; p < parts; ++p) c += red[p * NP + tid];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c = fmax(c, __shfl_xor(c, off, 64));
        if (tid == 0) red[NTH] = c;
    }
}

// Order-13 numerator / denominator for a skew-Hermitian A (see the rotated-strip helpers): returns P = V+U and
// Q = V-U as NATURAL strips.  regA holds A, regX is the staging plane pair, exch the small exchange area.
// ROT_OUT (NT = 4, look-ahead solve): P and Q are handed over as ROTATED strips (slot s = row tile (w + s) % NT of column strip
// w).  A rotated strip IS the wave's column strip, in a wave-dependent register order which the per-wave programs of the solve
// undo by renaming; only the tile no wave computed for itself, (w-1, w), has to travel -- one exchange round of two tiles per
// wave instead of two full matrices through the planes (64 stores, two barriers, 64 loads per wave).
template <int LD, int NT = 4, bool ROT_OUT = false>
__device__ __forceinline__ void expm_poly13_herm(double *regA, double *regX, double *exch, const int wave,
                                                 const int lane, Strip<NT> &Pn, Strip<NT> &Qn) {
    constexpr int NP = 16 * NT, NS = NT - 1;   // NS slots are computed, slot NT - 1 is the mirrored tile
    double *Are = regA, *Aim = regA + NP * LD, *Xre = regX, *Xim = regX + NP * LD;
    Strip<NT> A2, A4, A6;
    {
        Strip<NT> As;
        rot_load_strip<LD, NT>(Are, Aim, As, wave, lane);
        strip_zero(A2);
        gemm_rot<LD, NS, NT, NT == 4>(A2, Are, Aim, As, wave, lane);        // A2 = A*A (slots 0..2; NT = 4: slot 2 half)
    }
    if constexpr (NT == 4) {   // slot 2 += (partial sum of wave w+2)^dagger
        rot_exch_write<NT, 2>(exch, A2.re[2], A2.im[2], wave, lane, 1.0);
        __syncthreads();
        rot_exch_add<NT, 2>(exch, A2, wave, lane);
    }
    STAMP(13);
    rot_store_slots<LD, NS, NT>(Xre, Xim, A2, wave, lane);                  // X = A2, with the mirrored tiles
    rot_store_adjoint<LD, NT>(Xre, Xim, A2.re[1], A2.im[1], wave, lane, 1.0);
    __syncthreads();
    rot_load_slot3<LD, NT>(Xre, Xim, A2, wave, lane);
    strip_zero(A4);
    gemm_rot<LD, NS, NT, NT == 4>(A4, Xre, Xim, A2, wave, lane);            // A4 = A2*A2 (NT = 4: slot 2 half)
    if constexpr (NT == 4) {   // slot 2 += (partial sum of wave w+2)^dagger; the second barrier frees the area again
        rot_exch_write<NT, 2>(exch, A4.re[2], A4.im[2], wave, lane, 1.0);
        __syncthreads();
        rot_exch_add<NT, 2>(exch, A4, wave, lane);
        __syncthreads();
    }
    rot_exch_write<NT>(exch, A4.re[1], A4.im[1], wave, lane, 1.0);
    strip_zero(A6);
    gemm_rot<LD, NS, NT>(A6, Xre, Xim, A4, wave, lane, exch);               // A6 = A2*A4 (mirrored tile of A4 arrives late)
    STAMP(14);
    __syncthreads();                                                   // everybody is done reading X = A2
    rot_store_slots<LD, NS, NT>(Xre, Xim, A6, wave, lane);                  // X = A6
    rot_store_adjoint<LD, NT>(Xre, Xim, A6.re[1], A6.im[1], wave, lane, 1.0);
    // T = A6*(b13 A6 + b11 A4 + b9 A2) + b7 A6 + b5 A4 + b3 A2 + b1 I      (U = A*T)
    // V = A6*(b12 A6 + b10 A4 + b8 A2) + b6 A6 + b4 A4 + b2 A2 + b0 I
    // (the start values only need the computed slots: formed while the stores drain, in front of the barrier)
    Strip<NT> T, V;
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        T.re[t] = B13_7 * A6.re[t] + B13_5 * A4.re[t] + B13_3 * A2.re[t];
        T.im[t] = B13_7 * A6.im[t] + B13_5 * A4.im[t] + B13_3 * A2.im[t];
        V.re[t] = B13_6 * A6.re[t] + B13_4 * A4.re[t] + B13_2 * A2.re[t];
        V.im[t] = B13_6 * A6.im[t] + B13_4 * A4.im[t] + B13_2 * A2.im[t];
    }
    {   // the diagonal tile is slot 0: row 4r + rg == column c
        const int c = lane & 15, rg = lane >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * r + rg == c) { T.re[0][r] += B13_1; V.re[0][r] += B13_0; }
    }
    __syncthreads();
    rot_load_slot3<LD, NT>(Xre, Xim, A6, wave, lane);
    STAMP(15);
    gemm_dual13_rot<LD, NT>(T, V, Xre, Xim, A2, A4, A6, wave, lane);
    STAMP(16);
    // T's mirrored tiles through the exchange area (its last readers, inside the A6 product, are two barriers back)
    rot_exch_write<NT>(exch, T.re[1], T.im[1], wave, lane, 1.0);
    Strip<NT> Uo;
    strip_zero(Uo);
    // (here the late arrival of the mirrored tile inside the product measured 1 % slower: 26.0 vs 25.8 ms)
    __syncthreads();
    rot_exch_read<NT>(exch, T, wave, lane);
    gemm_rot<LD, NS, NT>(Uo, Are, Aim, T, wave, lane);                      // U = A*T, skew-Hermitian (slots 0..2)
    STAMP(17);
    if constexpr (ROT_OUT) {
        // P = V + U, Q = V - U with P = Q^dagger: slot NT-1 of P, tile (w-1, w), is the conjugate transpose of Q's tile
        // (w, w-1) = slot 1 of wave w-1, and vice versa.  The X region has been free since the barrier in front of this
        // product (its last readers were in the T/V product); the A region is still being read by slower waves.
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            Pn.re[t] = V.re[t] + Uo.re[t]; Pn.im[t] = V.im[t] + Uo.im[t];
            Qn.re[t] = V.re[t] - Uo.re[t]; Qn.im[t] = V.im[t] - Uo.im[t];
        }
        rot_exch_write<NT>(regX, Qn.re[1], Qn.im[1], wave, lane, 1.0);               // -> P(w, w+1) of wave w+1
        rot_exch_write<NT>(regX + NT * 512, Pn.re[1], Pn.im[1], wave, lane, 1.0);    // -> Q(w, w+1) of wave w+1
        STAMP(27);
        __syncthreads();
        STAMP(28);
        rot_exch_read<NT>(regX, Pn, wave, lane);
        rot_exch_read<NT>(regX + NT * 512, Qn, wave, lane);
        return;
    }
    __syncthreads();                                                   // A is dead, the X planes are free
    STAMP(26);
    // P = V + U -> X planes, Q = V - U -> A planes (natural positions).  V is Hermitian and U skew-Hermitian, so
    // P = Q^dagger: the tile each wave did not compute, (w-1, w), is the conjugate transpose of the OTHER matrix's
    // tile (w, w-1), which is slot 1 of wave w-1 -- neither V nor U needs its own mirror exchange.
    {
        Strip<NT> Pr, Qr;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            Pr.re[t] = V.re[t] + Uo.re[t]; Pr.im[t] = V.im[t] + Uo.im[t];
            Qr.re[t] = V.re[t] - Uo.re[t]; Qr.im[t] = V.im[t] - Uo.im[t];
        }
        rot_store_slots<LD, NS, NT>(Xre, Xim, Pr, wave, lane);
        rot_store_slots<LD, NS, NT>(Are, Aim, Qr, wave, lane);
        rot_store_adjoint<LD, NT>(Xre, Xim, Qr.re[1], Qr.im[1], wave, lane, 1.0);   // P(w, w+1) = Q(w+1, w)^dagger
        rot_store_adjoint<LD, NT>(Are, Aim, Pr.re[1], Pr.im[1], wave, lane, 1.0);   // Q(w, w+1) = P(w+1, w)^dagger
    }
    STAMP(27);
    __syncthreads();
    STAMP(28);
    strip_load_lds<NT, LD>(Xre, Xim, Pn, wave, lane);
    strip_load_lds<NT, LD>(Are, Aim, Qn, wave, lane);
}

// Hermitian 64 x 64 cells hand P and Q to the look-ahead solve as rotated strips (expm_poly13_herm)
template <int NT, bool HERM>
struct ExpmRotOut { static constexpr bool value = HERM && NT == 4; };

// Polynomial phase of one cell: Pade order / squaring count from ||A||_1 (in red[NTH]; A = -i dt H in LDS)
// and the numerator P = V+U and denominator Q = V-U as register strips (strip index `wave` = column strip
// this wave owns).  On return other waves may still be reading the LDS regions.
template <int NT, bool HERM = false>
__device__ __forceinline__ void expm_poly(const ExpmArgs &a, const int wave, const int lane, const int tid,
                                          double *smem, Strip<NT> &Pn, Strip<NT> &Qn, int &s, int &order,
                                          double &inv_b0sq, const int stamp0 = 11) {
    using LY = ExpmLds<NT>;
    constexpr int NP = LY::NP, LD = LY::LD, NTH = LY::NTH;
    double *Are = smem;          // A = -i dt H stays resident (left operand of A*A and A*T)
    double *Aim = Are + NP * LD;
    double *Xre = smem + LY::REG;  // staging of the current left operand (A2, A6)
    double *Xim = Xre + NP * LD;
    double *red = smem + 2 * LY::REG + LY::DV;
    const double nA = red[NTH];
    STAMP(stamp0 + 1);
    s = 0;  // ceil(log2(nA / 5.4)) for nA > 5.4 (Julia's exp!), from the binary exponent
    if (nA > 5.4) {
        const double r = nA / 5.4;
        const int e = ilogb(r);
        s = (r == ldexp(1.0, e)) ? e : e + 1;
    }
    if (s > 0) {
        const double f = ldexp(1.0, -s);
        for (int idx = tid; idx < NP * NP; idx += NTH) {
            const int i = idx / NP, j = idx - i * NP;
            Are[i * LD + j] *= f;
            Aim[i * LD + j] *= f;
        }
        __syncthreads();
    }

    // inv_b0sq = 1 / b0^2 of the Pade order in use: scale of the pivots of q(A) ~ b0 exp(-A/2)
    // The fast instantiation (PIVOTED = false) solves the Pade system with the unpivoted block
    // Gauss-Jordan; if a pivot turns out numerically unsafe (e.g. a pi-pulse in one step: q(A) has a zero
    // diagonal) it flags the cell, and the PIVOTED instantiation, launched afterwards over the flagged
    // cells only, re-evaluates P and Q and solves with full partial pivoting in LDS (what LAPACK gesv
    // does in the reference).
#ifdef EXP_ONLY13
    if (true) {
#else
    if (nA > 2.1) {
#endif
        order = 13;
        inv_b0sq = 1.0 / (B13_0 * B13_0);
        if constexpr (HERM && NT >= 3) {
            expm_poly13_herm<LD, NT, ExpmRotOut<NT, HERM>::value>(smem, smem + LY::REG, smem + 2 * LY::REG, wave, lane, Pn, Qn);
        } else {
        Strip<NT> A2, A4, A6;
        {
            Strip<NT> As;
            strip_load_lds<NT, LD>(Are, Aim, As, wave, lane);
            strip_zero(A2);
            gemm_xb<NT, LD>(A2, Are, Aim, As, lane);  // A2 = A*A
        }
        STAMP(stamp0 + 2);
        strip_store_lds<NT, LD>(Xre, Xim, A2, wave, lane);  // X = A2
        __syncthreads();
        strip_zero(A4);
        gemm_xb<NT, LD>(A4, Xre, Xim, A2, lane);  // A4 = A2*A2
        strip_zero(A6);
        gemm_xb<NT, LD>(A6, Xre, Xim, A4, lane);  // A6 = A2*A4
        STAMP(stamp0 + 3);
        __syncthreads();
        strip_store_lds<NT, LD>(Xre, Xim, A6, wave, lane);  // X = A6
        __syncthreads();
        // T = A6*(b13 A6 + b11 A4 + b9 A2) + b7 A6 + b5 A4 + b3 A2 + b1 I      (U = A*T)
        // V = A6*(b12 A6 + b10 A4 + b8 A2) + b6 A6 + b4 A4 + b2 A2 + b0 I
        Strip<NT> T, V;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            T.re[t] = B13_7 * A6.re[t] + B13_5 * A4.re[t] + B13_3 * A2.re[t];
            T.im[t] = B13_7 * A6.im[t] + B13_5 * A4.im[t] + B13_3 * A2.im[t];
            V.re[t] = B13_6 * A6.re[t] + B13_4 * A4.re[t] + B13_2 * A2.re[t];
            V.im[t] = B13_6 * A6.im[t] + B13_4 * A4.im[t] + B13_2 * A2.im[t];
        }
        strip_add_identity<NT>(T, B13_1, wave, lane);
        strip_add_identity<NT>(V, B13_0, wave, lane);
        STAMP(stamp0 + 4);
        gemm_dual13<NT, LD>(T, V, Xre, Xim, A2, A4, A6, wave, lane);
        STAMP(stamp0 + 5);
        Strip<NT> Uo;
        strip_zero(Uo);
        gemm_xb<NT, LD>(Uo, Are, Aim, T, lane);  // U = A*T
        STAMP(stamp0 + 6);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            Pn.re[t] = V.re[t] + Uo.re[t];
            Pn.im[t] = V.im[t] + Uo.im[t];
            Qn.re[t] = V.re[t] - Uo.re[t];
            Qn.im[t] = V.im[t] - Uo.im[t];
        }
        }
    } else {
        const double *c;
        if (nA > 0.95) { c = c_pade9; order = 9; }
        else if (nA > 0.25) { c = c_pade7; order = 7; }
        else if (nA > 0.015) { c = c_pade5; order = 5; }
        else { c = c_pade3; order = 3; }
        inv_b0sq = 1.0 / (c[0] * c[0]);
        Strip<NT> Pk, Ui, V;
        {
            Strip<NT> As;
            strip_load_lds<NT, LD>(Are, Aim, As, wave, lane);
            strip_zero(Pk);
            gemm_xb<NT, LD>(Pk, Are, Aim, As, lane);  // A2
        }
        strip_store_lds<NT, LD>(Xre, Xim, Pk, wave, lane);  // X = A2
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            Ui.re[t] = c[3] * Pk.re[t];
            Ui.im[t] = c[3] * Pk.im[t];
            V.re[t] = c[2] * Pk.re[t];
            V.im[t] = c[2] * Pk.im[t];
        }
        strip_add_identity<NT>(Ui, c[1], wave, lane);
        strip_add_identity<NT>(V, c[0], wave, lane);
        const int half = (order + 1) / 2;  // number of coefficient pairs
        for (int kk = 2; kk < half; ++kk) {
            Strip<NT> Pnew;
            strip_zero(Pnew);
            gemm_xb<NT, LD>(Pnew, Xre, Xim, Pk, lane);  // A2 * A2^(kk-1)
            Pk = Pnew;
            const double cu = c[2 * kk + 1], cv = c[2 * kk];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                Ui.re[t] += cu * Pk.re[t];
                Ui.im[t] += cu * Pk.im[t];
                V.re[t] += cv * Pk.re[t];
                V.im[t] += cv * Pk.im[t];
            }
        }
        Strip<NT> Uo;
        strip_zero(Uo);
        gemm_xb<NT, LD>(Uo, Are, Aim, Ui, lane);  // U = A*Ui
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            Pn.re[t] = V.re[t] + Uo.re[t];
            Pn.im[t] = V.im[t] + Uo.im[t];
            Qn.re[t] = V.re[t] - Uo.re[t];
            Qn.im[t] = V.im[t] - Uo.im[t];
        }
        if constexpr (ExpmRotOut<NT, HERM>::value) {
            // (rare: ||A||_1 <= 2.1) the look-ahead solve expects rotated strips: the same column strip in another
            // register order -- through this wave's own columns of the X planes
            __syncthreads();   // everybody is done reading the X planes
            strip_store_lds<NT, LD>(Xre, Xim, Pn, wave, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            rot_load_strip<LD, NT>(Xre, Xim, Pn, wave, lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            strip_store_lds<NT, LD>(Xre, Xim, Qn, wave, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            rot_load_strip<LD, NT>(Xre, Xim, Qn, wave, lane);
        }
    }

}

// form A, norm and polynomial phase of one cell by the whole workgroup (the caller guarantees that the LDS
// regions are free on entry)
template <int NT>
__device__ __forceinline__ void expm_numden(const ExpmArgs &a, const int cell, const int wave, const int lane,
                                            const int tid, double *smem, Strip<NT> &Pn, Strip<NT> &Qn, int &s,
                                            int &order, double &inv_b0sq, const int stamp0 = 11) {
    using LY = ExpmLds<NT>;
    expm_form_a<NT>(a, cell, smem, tid, LY::NTH, 0, LY::NP * LY::NP / 2);
    __syncthreads();
    STAMP(stamp0 + 0);
    expm_norm_partial<NT>(smem, tid, LY::NTH / LY::NP);
    __syncthreads();
    expm_norm_combine<NT>(smem, tid, LY::NTH / LY::NP);
    __syncthreads();
    expm_poly<NT>(a, wave, lane, tid, smem, Pn, Qn, s, order, inv_b0sq, stamp0);
}

// Squarings and the store of U_kn (row-major interleaved complex) for one cell.
template <int NT>
__device__ __forceinline__ void expm_finish(const ExpmArgs &a, const int cell, const int wave, const int lane,
                                            double *smem, Strip<NT> &Pn, const int s) {
    using LY = ExpmLds<NT>;
    constexpr int NP = LY::NP, LD = LY::LD;
    double *Xre = smem + LY::REG, *Xim = Xre + NP * LD;
    // ---- squarings ----
    for (int it = 0; it < s; ++it) {
        __syncthreads();
        strip_store_lds<NT, LD>(Xre, Xim, Pn, wave, lane);
        __syncthreads();
        Strip<NT> Sq;
        strip_zero(Sq);
        gemm_xb<NT, LD>(Sq, Xre, Xim, Pn, lane);
        Pn = Sq;
    }

    // ---- store U (row-major interleaved complex) ----
    {
        double2 *Uc = a.U + (size_t)cell * NP * NP;
        const int col = 16 * wave + (lane & 15), rg = lane >> 4;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + 4 * r + rg;
                Uc[row * NP + col] = make_double2(Pn.re[t][r], Pn.im[t][r]);
            }
    }
}

template <int NT>
__device__ __forceinline__ void expm_stats(const ExpmArgs &a, const int s, const int order) {
    stat_add(a.stats, 0, (unsigned long long)s);
    stat_add(a.stats, 3 + (order == 13 ? 4 : (order - 3) / 2), 1ull);
    if (s > 0) atomicMax(&a.flags[1], s);   // (flags[1] starts at 0)
}

// Robust single-cell path (second pass over flagged cells): P and Q are re-evaluated and the system is
// solved with full partial pivoting in LDS (what LAPACK gesv does in the reference).
template <int NT>
__device__ __forceinline__ void expm_cell_pivoted(const ExpmArgs &a, const int cell) {
    using LY = ExpmLds<NT>;
    constexpr int NP = LY::NP, LD = LY::LD, NTH = LY::NTH;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *Are = smem, *Aim = Are + NP * LD, *Xre = smem + LY::REG, *Xim = Xre + NP * LD;
    double *red = smem + 2 * LY::REG + LY::DV;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Strip<NT> Pn, Qn;
    int s, order;
    double inv_b0sq;
    expm_numden<NT>(a, cell, wave, lane, tid, smem, Pn, Qn, s, order, inv_b0sq);
    __syncthreads();
    strip_store_lds<NT, LD>(Xre, Xim, Qn, wave, lane);   // Q -> X planes
    strip_store_lds<NT, LD>(Are, Aim, Pn, wave, lane);   // P -> A planes (A is dead by now)
    __syncthreads();
    const bool ok = pivoted_solve_lds<NP, LD, NTH>(Xre, Xim, Are, Aim, red, tid);
    strip_load_lds<NT, LD>(Are, Aim, Pn, wave, lane);    // X = Q^-1 P
    if (!ok && tid == 0) atomicOr(&a.flags[0], 1);       // exactly singular denominator
    if (tid == 0) stat_add(a.stats, 9, 1ull);          // cells that needed the pivoted solve
    __syncthreads();
    expm_finish<NT>(a, cell, wave, lane, smem, Pn, s);
}

// ||A||_1 <= dt (||H0_k||_1 + sum_l |eps_l| ||H_l||_1), -1 without the operator norms
__device__ __forceinline__ double expm_norm_bound(const ExpmArgs &a, const int cell) {
    if (!a.n1) return -1.0;
    const int kc = cell / a.N_T, n = cell - kc * a.N_T, k = a.rep ? a.rep[kc] : kc;
    const double *n1c = a.n1 + a.n1_k + (size_t)(a.hc_per_traj ? k : 0) * a.L;
    double bound = a.n1[k];
    for (int l = 0; l < a.L; ++l)
        bound += fabs(a.eps[(size_t)l * a.N_T + n] * (a.shape ? a.shape[(size_t)l * a.N_T + n] : 1.0)) * n1c[l];
    return bound * a.dts[n] * (1.0 + 1e-12);
}

// Fast single-cell path: one workgroup per cell, unpivoted block Gauss-Jordan with look-ahead.
template <int NT, bool HERM>
__device__ __forceinline__ void expm_single(const ExpmArgs &a, const int cell, const int tid) {
    using LY = ExpmLds<NT>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Strip<NT> Pn, Qn;
    int s, order;
    double inv_b0sq;
    STAMP(0);
    {
        using LYY = ExpmLds<NT>;
        // ||A||_1 <= dt (||H0_k||_1 + sum_l |eps_l| ||H_l||_1): when this bound already lies in (2.1, 5.4] it certifies
        // order 13 without squaring and the norm of the cell is not needed (a cell whose true norm is below 2.1 then
        // gets order 13 instead of Julia's 9: same result to rounding); every other case measures the norm
        const double bound = expm_norm_bound(a, cell);
        if constexpr (HERM && NT >= 3) expm_form_a_herm64<64 * NT, NT>(a, cell, smem, tid);
        else expm_form_a<NT>(a, cell, smem, tid, LYY::NTH, 0, LYY::NP * LYY::NP / 2);
        if (bound > 2.1 && bound <= 5.4) {
            if (tid == 0) (smem + 2 * LYY::REG + LYY::DV)[LYY::NTH] = bound;
            __syncthreads();
            STAMP(11);
        } else {
            __syncthreads();
            STAMP(11);
            expm_norm_partial<NT>(smem, tid, LYY::NTH / LYY::NP);
            __syncthreads();
            expm_norm_combine<NT>(smem, tid, LYY::NTH / LYY::NP);
            __syncthreads();
        }
        expm_poly<NT, HERM>(a, wave, lane, tid, smem, Pn, Qn, s, order, inv_b0sq);
    }
    STAMP(2);
    double minrel = 1e300;
#ifdef GRAPE_DIAG
    if constexpr (ExpmRotOut<NT, HERM>::value) {   // the ablations run the split schedule, which takes natural strips
        constexpr int LD = LY::LD, NP = LY::NP;
        double *Xre = smem + LY::REG, *Xim = Xre + NP * LD;
        __syncthreads();
        rot_store_slots<LD, NT, NT>(Xre, Xim, Pn, wave, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        strip_load_lds<NT, LD>(Xre, Xim, Pn, wave, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        rot_store_slots<LD, NT, NT>(Xre, Xim, Qn, wave, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        strip_load_lds<NT, LD>(Xre, Xim, Qn, wave, lane);
    }
    if (!(a.ablate & 2)) block_gj_solve<NT>(Qn, Pn, smem + LY::REG, smem + 2 * LY::REG, wave, lane, minrel, inv_b0sq, !(a.ablate & 1));
    if (a.ablate & 3) minrel = 1.0;
#else
    // (NT < 4 keeps the split schedule: with several cells per CU the latencies are hidden by the other workgroups,
    // measured 16.5 ms either way for N = 48 and 5.3 -> 5.5 ms with the look-ahead for N = 32)
    if constexpr (NT == 4)
        block_gj_solve_lookahead<NT, ExpmRotOut<NT, HERM>::value>(Qn, Pn, smem + LY::REG, smem + 2 * LY::REG, smem,
                                                                  smem + GJ_LA_SIZE, wave, lane, minrel, inv_b0sq);
    else
        block_gj_solve<NT>(Qn, Pn, smem + LY::REG, smem + 2 * LY::REG, wave, lane, minrel, inv_b0sq, true);
#endif
    // |pivot| < 1e-3 b0 (or NaN) in any of the diagonal tiles -> flag the cell for the pivoted pass
    if (lane == 0 && !(minrel > 1e-6)) { a.cellflag[cell] = 1; atomicAdd(&a.flags[2], 1); }   // flags[2]: flagged cells
    STAMP(3);
    expm_finish<NT>(a, cell, wave, lane, smem, Pn, s);
    STAMP(4);
    if (tid == 0) expm_stats<NT>(a, s, order);
}

// PERSISTENT variant of the fast path (NT = 4): one workgroup per CU walks its cells, and while wave 0 inverts the
// first diagonal tile of cell c -- the one stretch of the solve in which the other three waves have nothing to do --
// those waves form A of the workgroup's NEXT cell in the (by then dead) A region of the LDS, so that the 7-8 K cycles
// of operand fetch + norm at the front of a cell run underneath the 6.6 K cycles of the previous cell's inversion.
// The cells of an XCD are dealt to its workgroups round-robin: concurrently running workgroups touch neighbouring
// cells of the same trajectories (H0_k stays in that XCD's L2).
template <int NT, bool HERM>
struct ExpmPrefetchHook {
    const ExpmArgs &a;
    double *smem;
    int next_cell, tid;   // next_cell < 0: nothing to prefetch
    __device__ __forceinline__ void operator()(int) const {}
    __device__ __forceinline__ void first_inversion_idle() const {
        using LY = ExpmLds<NT>;
        if (next_cell < 0) return;
        const int t = tid - 64;   // waves 1..NT-1
        if constexpr (HERM && NT >= 3) expm_form_a_herm64<64 * (NT - 1), NT>(a, next_cell, smem, t);
        else expm_form_a<NT>(a, next_cell, smem, t, LY::NTH - 64, 0, LY::NP * LY::NP / 2);
        if (t == 0) {
            double *red = smem + 2 * LY::REG + LY::DV;
            const double bound = expm_norm_bound(a, next_cell);
            const bool cert = bound > 2.1 && bound <= 5.4;
            red[LY::NTH] = cert ? bound : 0.0;
            red[LY::NTH + 1] = cert ? 0.0 : 1.0;   // 1: the norm of the prefetched A has to be measured
        }
    }
};

template <int NT, bool HERM>
__device__ __forceinline__ void expm_persistent(const ExpmArgs &a, const int tid_in) {
    using LY = ExpmLds<NT>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *red = smem + 2 * LY::REG + LY::DV;
    const int tid = tid_in, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncell = a.K * a.N_T;
    const int x = blockIdx.x & 7, per_x = gridDim.x >> 3;
    const int lo = (int)((long)x * ncell / 8), hi = (int)((long)(x + 1) * ncell / 8);
    bool have_a = false;
    const int lane0 = lane, tid0 = tid;
    int st_s = 0, st_max = 0, st_ord[5] = {0, 0, 0, 0, 0};
    for (int cell = lo + ((int)blockIdx.x >> 3); cell < hi; cell += per_x) {
        // the per-lane LDS addresses are recomputed in every iteration: hoisted out of the cell loop they would stay
        // live through the whole body (several dozen registers, i.e. spills at the 512-register limit)
        int lane = lane0, tid = tid0;
        asm volatile("" : "+v"(lane), "+v"(tid));
        Strip<NT> Pn, Qn;
        int s, order;
        double inv_b0sq;
        bool measure;
#ifdef GRAPE_DIAG
        if (tid == 0) g_diag_off[blockIdx.x & 1023] = (cell != lo + ((int)blockIdx.x >> 3) + per_x);
        __syncthreads();
#endif
        STAMP(0);
        if (!have_a) {
            const double bound = expm_norm_bound(a, cell);
            if constexpr (HERM && NT >= 3) expm_form_a_herm64<64 * NT, NT>(a, cell, smem, tid);
            else expm_form_a<NT>(a, cell, smem, tid, LY::NTH, 0, LY::NP * LY::NP / 2);
            measure = !(bound > 2.1 && bound <= 5.4);
            if (!measure && tid == 0) red[LY::NTH] = bound;
            __syncthreads();
        } else {
            measure = red[LY::NTH + 1] != 0.0;   // written before the barriers of the previous cell's solve
        }
        if (measure) {
            __syncthreads();   // (everybody has read the flag)
            expm_norm_partial<NT>(smem, tid, LY::NTH / LY::NP);
            __syncthreads();
            expm_norm_combine<NT>(smem, tid, LY::NTH / LY::NP);
            __syncthreads();
        }
        STAMP(11);
        expm_poly<NT, HERM>(a, wave, lane, tid, smem, Pn, Qn, s, order, inv_b0sq);
        STAMP(2);
        const int next = cell + per_x;
        // only cells without squarings leave the A region alone until the end of the solve: with s > 0 nothing
        // changes either (the squarings use the X region), so the prefetch is unconditional
        const ExpmPrefetchHook<NT, HERM> hook{a, smem, next < hi ? next : -1, tid};
        double minrel = 1e300;
        if constexpr (NT == 4)   // (the only size the persistent variant is launched for)
            block_gj_solve_lookahead<NT, ExpmRotOut<NT, HERM>::value>(Qn, Pn, smem + LY::REG, smem + 2 * LY::REG, smem + LY::LA0,
                                                                      smem + LY::LA1, wave, lane, minrel, inv_b0sq, hook);
        else
            block_gj_solve<NT>(Qn, Pn, smem + LY::REG, smem + 2 * LY::REG, wave, lane, minrel, inv_b0sq, true, hook);
        have_a = next < hi;
        if (lane == 0 && !(minrel > 1e-6)) { a.cellflag[cell] = 1; atomicAdd(&a.flags[2], 1); }
        STAMP(3);
        expm_finish<NT>(a, cell, wave, lane, smem, Pn, s);
        STAMP(4);
        // statistics are accumulated per workgroup and published once (three atomics per cell otherwise)
        st_s += s; st_max = max(st_max, s);
        st_ord[order == 13 ? 4 : (order - 3) / 2] += 1;
        if (s > 0) __syncthreads();   // the last squaring read the X region, which the next cell writes after its first product
    }
#ifdef GRAPE_DIAG
    if (tid0 == 0) g_diag_off[blockIdx.x & 1023] = 0;
#endif
    if (tid0 == 0) {
        stat_add(a.stats, 0, (unsigned long long)st_s);
#pragma unroll
        for (int o = 0; o < 5; ++o)
            if (st_ord[o]) stat_add(a.stats, 3 + o, (unsigned long long)st_ord[o]);
        if (st_max > 0) atomicMax(&a.flags[1], st_max);
    }
}

template <int NT, bool HERM>
__global__ void __launch_bounds__(NT * 64) expm_persistent_kernel(ExpmArgs a) {
    expm_persistent<NT, HERM>(a, threadIdx.x);
}

template <int NT, bool PIVOTED, bool HERM = false>
__global__ void __launch_bounds__(NT * 64) expm_pade_kernel(ExpmArgs a) {
    const int ncell = a.K * a.N_T;
    if constexpr (!PIVOTED) {
        expm_single<NT, HERM>(a, xcd_remap(blockIdx.x, ncell), threadIdx.x);   // one workgroup per cell
    } else {
        // second pass, small grid: every workgroup scans a slice of the flags and re-solves flagged cells
        // (flags[2] counts them: the usual evaluation has none and the pass ends here)
        if (a.flags[2] == 0) return;
        for (int cell = blockIdx.x; cell < ncell; cell += gridDim.x) {
            if (!a.cellflag[cell]) continue;
            __syncthreads();
            expm_cell_pivoted<NT>(a, cell);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Kernel 2/4: forward and backward sweeps.  One workgroup (256 threads) per trajectory walks
// the serial recurrence; U_kn streams from HBM (row-major interleaved), the state lives in LDS.
//   forward : Psi_n     = U_n Psi_{n-1},  storage[k][n] (optimize.jl:731-738), tau_k (:753)
//   backward: chi_{n-1} = U_n^dagger chi_n (optimize.jl:881 bottom block), chi boundary :848-868
// Thread (lane = column j, wave q) covers rows [q*RW, (q+1)*RW) : loads are 16 B/lane coalesced.
// ---------------------------------------------------------------------------------------
struct SweepArgs {
    const double2 *U;     // [K][N_T][NP*NP]
    const double2 *psi0;  // [K][N]  (forward) initial states
    const double2 *target;// [K][N]
    const double *weights;// nullptr or [K]
    double2 *store;       // [K][N_T+1][NP] forward: Psi(t_n); backward: chi(t_n)
    double2 *tau;         // [K]  (forward: out; backward: in)
    const double *f;      // backward: all-reduced f = sum_k w_k tau_k  (2 doubles)
    double *rho;          // [K] backward: out
    int *flags;
    double chi_min_norm;
    int K, K_total, N, N_T, functional;
    const int *cls;       // nullptr or [K]: generator class of trajectory k (row of U it reads)
    // state running cost g_b = <Psi|D|Psi> (optimize.jl:856-866, 897-908): xi_k(t_n) = -D Psi_k(t_n),
    // trapezoid weights wq[n]; nullptr / 0 when off
    const double2 *xi;    // [K][N_T+1][NP]
    const double *wq;     // [N_T+1]
    double lambda_b;
    // unit_chi: the backward sweep starts from target_k / ||target_k|| instead of chi_k(T) = c_k target_k, so that it
    // does not depend on the forward sweep (both run concurrently); the recursion is linear in chi, and
    // chi_coeff_kernel supplies z_k = rho_k conj(c_k / |c_k|) afterwards (tau_grads, gradient) -- only without the
    // xi inhomogeneity of the state running cost
    const double *inv_tnorm;   // [K] 1 / ||target_k||
    int unit_chi;
    // host-supplied boundary states chi_k(T) of grape_backward_chi ([K][N], not normalised): replace c_k target_k
    const double2 *chi_in;
    // fault injection for the cooperative kernels (tests only, env GRAPE_TEST_DROP_SIBLING): sibling `drop_sibling - 1`
    // of every trajectory exits at once, so that the others run into their spin limit -- the evaluation must fail
    // with GRAPE_ERR_HIP and the grid must drain; 0 in production
    int drop_sibling;
    // round 5: the exponential kernel (asm/gen_t16.py) has already carried the state of trajectory k over its first
    // resume[k] steps -- forward from t = 0 / backward from t = T -- and stored those states; the sweep picks up behind them
    // (nullptr: from the boundary)
    const int *resume;
    // round 6, cooperative sweeps: 1 = the published slices ARE the signal -- the storage rows a sweep will write are armed
    // with an all-ones bit pattern before the launch, a reader polls the elements it needs until none shows the pattern;
    // the step counter, the wait for the store acknowledgements and one barrier per step disappear (0: step counter)
    int xmode;
    // ... and where the siblings of a trajectory check that they share an XCD ([K][32] XCC ids, -1 at launch; nullptr: not
    // checked): then the stores of the exchange only have to reach that XCD's L2 (sc0) instead of the memory fabric
    int *xcc;
};

// c_k of chi_k(T) = c_k target_k for the three functionals (docs/src/tutorial.md:349-356, 402)
__device__ __forceinline__ void chi_coefficient(const SweepArgs &a, const int k, double &cr, double &ci) {
    if (a.unit_chi) { cr = a.inv_tnorm[k]; ci = 0.; return; }
    const double w = a.weights ? a.weights[k] : 1.0;
    const double Kt = (double)a.K_total;
    if (a.functional == 0) { cr = w * a.f[0] / (Kt * Kt); ci = w * a.f[1] / (Kt * Kt); }
    else if (a.functional == 1) { const double2 t = a.tau[k]; cr = w * t.x / Kt; ci = w * t.y / Kt; }
    else { cr = w / (2.0 * Kt); ci = 0.; }
}

// element i of the boundary state chi_k(T) before the running-cost term: the user's chi (grape_backward_chi,
// optimize.jl:845-855) or c_k target_k of the built-in functionals
__device__ __forceinline__ double2 chi_boundary(const SweepArgs &a, const int k, const int i) {
    if (a.chi_in) return a.chi_in[(size_t)k * a.N + i];
    double cr, ci;
    chi_coefficient(a, k, cr, ci);
    const double2 t = a.target[(size_t)k * a.N + i];
    return make_double2(cr * t.x - ci * t.y, cr * t.y + ci * t.x);
}

// after concurrent sweeps: true rho_k = |c_k| ||target_k|| (optimize.jl:867-868, guard :1021-1025) and
// z_k = conj(c_k) ||target_k||, the factor that turns <chi~'_l|Psi> of the unit backward states into tau_grads
struct ChiCoeffArgs {
    SweepArgs s;      // weights, tau, f, functional, K_total, flags, chi_min_norm, inv_tnorm
    double *rho;      // [K]
    double2 *z;       // [K]
};
__global__ void chi_coeff_kernel(ChiCoeffArgs a) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.s.K) return;
    SweepArgs s = a.s;
    s.unit_chi = 0;
    double cr, ci;
    chi_coefficient(s, k, cr, ci);
    const double tn = 1.0 / a.s.inv_tnorm[k];
    const double rho = sqrt(cr * cr + ci * ci) * tn;
    a.rho[k] = rho;
    a.z[k] = make_double2(cr * tn, -ci * tn);
    if (rho < a.s.chi_min_norm) atomicOr(&a.s.flags[0], 2);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// one reduce-scatter step of the forward sweep: lanes l and l^OFF split 2H row sums between them
template <int H, int OFF, int RWT>
__device__ __forceinline__ void rs_step(double (&pr)[RWT], double (&pi)[RWT], int lane, int &row) {
    const bool up = (lane & OFF) != 0;  // upper lanes keep the upper half of the rows
#pragma unroll
    for (int r = 0; r < H; ++r) {
        const double sr = up ? pr[r] : pr[r + H], si = up ? pi[r] : pi[r + H];
        const double kr = up ? pr[r + H] : pr[r], ki = up ? pi[r + H] : pi[r];
        pr[r] = kr + __shfl_xor(sr, OFF, 64);
        pi[r] = ki + __shfl_xor(si, OFF, 64);
    }
    row += up ? H : 0;
}

template <int NP, bool BACKWARD>
__device__ __forceinline__ void sweep_body(const SweepArgs &a, const int k) {
    constexpr int NW = NP == 48 ? 3 : 4, RW = NP / NW;  // waves and rows per wave (NP = 16 -> 4; NP = 48: 3 x 16)
    __shared__ double2 x[2][NP];      // state ping-pong
    __shared__ double2 part[NW][NP];  // cross-wave partials (backward)
    __shared__ double sc[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double2 *Uk = a.U + (size_t)(a.cls ? a.cls[k] : k) * a.N_T * NP * NP;
    double2 *st = a.store + (size_t)k * (a.N_T + 1) * NP;

    // ---- initial state ----
    if (!BACKWARD) {
        if (tid < NP) {
            double2 v = tid < a.N ? a.psi0[(size_t)k * a.N + tid] : make_double2(0., 0.);
            x[0][tid] = v;
            st[tid] = v;
        }
    } else {
        // chi_k(T) = coeff_k * target_k, rho_k = ||chi_k||, chi_k /= rho_k (optimize.jl:848-868)
        double2 v = make_double2(0., 0.);
        if (tid < a.N) {
            v = chi_boundary(a, k, tid);
            if (a.xi) {   // chi_k(T) += lambda_b dt/2 xi_k(T)   (optimize.jl:856-866)
                const double2 x_ = a.xi[((size_t)k * (a.N_T + 1) + a.N_T) * NP + tid];
                const double c = a.lambda_b * a.wq[a.N_T];
                v.x += c * x_.x; v.y += c * x_.y;
            }
        }
        if (wave == 0) {
            double n2 = wave_sum(lane < NP ? v.x * v.x + v.y * v.y : 0.);
            if (lane == 0) sc[0] = sqrt(n2);
        }
        __syncthreads();
        const double rho = sc[0];
        if (tid == 0 && !a.unit_chi) {
            a.rho[k] = rho;
            if (rho < a.chi_min_norm) atomicOr(&a.flags[0], 2);
        }
        if (tid < NP) {
            const double ir = rho > 0. ? 1.0 / rho : 0.;
            v.x *= ir; v.y *= ir;
            x[0][tid] = v;
            st[(size_t)a.N_T * NP + tid] = v;
        }
    }
    __syncthreads();
    const int step_start = a.resume ? min(max(a.resume[k], 0), a.N_T) : 0;
    if (step_start > 0) {   // the state behind the steps the exponential kernel has done (it stored every one of them)
        if (tid < NP) x[0][tid] = st[(size_t)(BACKWARD ? a.N_T - step_start : step_start) * NP + tid];
        __syncthreads();
    }

    int cur = 0;
    // software prefetch: the U tiles of the next steps are in flight while step n is reduced (the recurrence only
    // couples the steps through the state, never through U)
    auto load_tile = [&](double2 (&dst)[RW], int step) __attribute__((always_inline)) {
        const int nn = BACKWARD ? a.N_T - 1 - step : step;
        const double2 *Un = Uk + (size_t)nn * NP * NP;
#pragma unroll
        for (int r = 0; r < RW; ++r)
            {   // streamed once per sweep: non-temporal (measured 2.94 -> 2.81 / 2.83 -> 2.79 ms at C3, both orders of an A/B)
                typedef double v2d __attribute__((ext_vector_type(2)));
                const v2d t = lane < NP ? __builtin_nontemporal_load((const v2d *)&Un[(size_t)(wave * RW + r) * NP + lane]) : (v2d){0., 0.};
                dst[r] = make_double2(t.x, t.y);
            }
    };
    // The tiles of the next D steps are in flight in a ring of registers with STATIC indices (the time loop is unrolled D
    // times).  A ring that is shifted with register moves makes every step wait for the newest load (s_waitcnt vmcnt(0)):
    // the sweeps then run at one memory latency per step whatever the depth.
    // (measured at the C3 shape, forward + backward in one launch: N = 64 2.74 / 2.82 / 2.82 ms with 2 / 3 / 4; N = 48 2.04 / 1.91 /
    // 3.86 ms with 3 / 4 / 6 -- six tiles spill --, N = 32 1.22 / 1.15 / 1.12 ms with 3 / 4 / 6)
    constexpr int D = NP == 64 ? 2 : NP == 48 ? 4 : 6;
    double2 un[D][RW];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (step_start + d < a.N_T) load_tile(un[d], step_start + d);
    for (int step0 = step_start; step0 < a.N_T; step0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int step = step0 + d;
        if (step >= a.N_T) break;
        const int n = BACKWARD ? a.N_T - 1 - step : step;
        double2 (&ucur)[RW] = un[d];
        if (!BACKWARD) {
            // y_i = sum_j U[i][j] x_j : lane j holds the products of this wave's RW rows; the sums over
            // lanes are a wavefront reduce-scatter (RW values -> 1 value per lane, then a 64/RW-lane sum)
            const double2 xv = lane < NP ? x[cur][lane] : make_double2(0., 0.);
            double pr[RW], pi[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const double2 u = ucur[r];
                pr[r] = u.x * xv.x - u.y * xv.y;
                pi[r] = u.x * xv.y + u.y * xv.x;
            }
            if (step + D < a.N_T) load_tile(un[d], step + D);   // (the tile of this step is consumed)
            int row = 0;
            rs_step<RW / 2, 32>(pr, pi, lane, row);
            if constexpr (RW >= 4) rs_step<RW / 4, 16>(pr, pi, lane, row);
            if constexpr (RW >= 8) rs_step<RW / 8, 8>(pr, pi, lane, row);
            if constexpr (RW >= 16) rs_step<RW / 16, 4>(pr, pi, lane, row);
            // lanes that share `row` differ in the low log2(64/RW) bits
#pragma unroll
            for (int off = 32 / RW; off >= 1; off >>= 1) {
                pr[0] += __shfl_xor(pr[0], off, 64);
                pi[0] += __shfl_xor(pi[0], off, 64);
            }
            if ((lane & (64 / RW - 1)) == 0) x[cur ^ 1][wave * RW + row] = make_double2(pr[0], pi[0]);
            __syncthreads();
            if (tid < NP) st[(size_t)(n + 1) * NP + tid] = x[cur ^ 1][tid];
        } else {
            // y_j = sum_i conj(U[i][j]) x_i : lane j accumulates over this wave's rows
            double ar = 0., ai = 0.;
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int i = wave * RW + r;
                const double2 u = ucur[r];
                const double2 xi = x[cur][i];
                ar += u.x * xi.x + u.y * xi.y;
                ai += u.x * xi.y - u.y * xi.x;
            }
            if (step + D < a.N_T) load_tile(un[d], step + D);   // (the tile of this step is consumed)
            if (lane < NP) part[wave][lane] = make_double2(ar, ai);
            __syncthreads();
            if (tid < NP) {
                double2 s = part[0][tid];
#pragma unroll
                for (int q = 1; q < NW; ++q) { s.x += part[q][tid].x; s.y += part[q][tid].y; }
                if (a.xi && n > 0) {   // chi(t_n) += lambda_b Dt_n / rho_k xi_k(t_n)   (optimize.jl:897-908)
                    const double2 x_ = a.xi[((size_t)k * (a.N_T + 1) + n) * NP + tid];
                    const double c = a.lambda_b * a.wq[n] / sc[0];
                    s.x += c * x_.x; s.y += c * x_.y;
                }
                x[cur ^ 1][tid] = s;
                st[(size_t)n * NP + tid] = s;
            }
        }
        __syncthreads();
        cur ^= 1;
      }
    }
    if (!BACKWARD) {
        // tau_k = <target_k | Psi_k(T)>  (optimize.jl:753)
        if (wave == 0) {
            double pr = 0., pi = 0.;
            if (lane < a.N) {
                const double2 t = a.target[(size_t)k * a.N + lane];
                const double2 p = x[cur][lane];
                pr = t.x * p.x + t.y * p.y;
                pi = t.x * p.y - t.y * p.x;
            }
            pr = wave_sum(pr);
            pi = wave_sum(pi);
            if (lane == 0) a.tau[k] = make_double2(pr, pi);
        }
    }
}

template <int NP, bool BACKWARD>
__global__ void __launch_bounds__(NP == 48 ? 192 : 256) sweep_kernel(SweepArgs a) {
    sweep_body<NP, BACKWARD>(a, blockIdx.x);
}

// Both sweeps in one launch (blocks [0, K): forward, [K, 2K): backward with unit_chi, see SweepArgs): with one
// workgroup per trajectory a sweep occupies K of the 256 CUs, and the backward recursion is linear in chi.
template <int NP>
__global__ void __launch_bounds__(NP == 48 ? 192 : 256) sweep_pair_kernel(SweepArgs af, SweepArgs ab) {
    if ((int)blockIdx.x < af.K) sweep_body<NP, false>(af, blockIdx.x);
    else sweep_body<NP, true>(ab, blockIdx.x - af.K);
}

// tau partial sums of this shard: out[0..1] = sum w tau, out[2] = sum w |tau|^2, out[3] = Re sum w tau
__global__ void tau_reduce_kernel(const double2 *tau, const double *weights, int K, double *out) {
    double fr = 0., fi = 0., ss = 0.;
    for (int k = threadIdx.x; k < K; k += 64) {
        const double w = weights ? weights[k] : 1.0;
        const double2 t = tau[k];
        fr += w * t.x; fi += w * t.y; ss += w * (t.x * t.x + t.y * t.y);
    }
    fr = wave_sum(fr); fi = wave_sum(fi); ss = wave_sum(ss);
    if (threadIdx.x == 0) { out[0] = fr; out[1] = fi; out[2] = ss; out[3] = fr; out[4] = 0.; out[5] = 0.; out[6] = 0.; out[7] = 0.; }
}

// ---------------------------------------------------------------------------------------
// Kernel 5: per-cell derivative overlaps
//   tau_grads[k][n][l] = rho_k <chi'_l | Psi_k(t_{n-1})>,   chi'_l = d/d eps_nl exp(-i H^dagger (-dt)) chi_k(t_n)
// (optimize.jl:893-895 / :957-970).  chi'_l is the top block of exp(-i G[H^dagger] (-dt)) applied to
// the extended state (docs/src/background.md:443-497); it is evaluated here by the series of that
// exponential on the extended *vector* (identical to the recursion of taylor_grad_step!,
// optimize.jl:604-653) instead of densifying the (L+1)N block matrix.
// One workgroup walks CPB consecutive cells of one trajectory; H0_k^dagger and mu_l^dagger tiles
// stay in registers (thread (row i, column chunk q)), vectors live in LDS, norms and overlaps use
// wavefront shuffles.
// ---------------------------------------------------------------------------------------
struct DerivArgs {
    const double *H0t;   // [K][2][NP*NP] planar row-major of H0_k^T   (H^dagger = conj of this)
    const double *Hct;   // [Kc][L][2][NP*NP] planar row-major of H_l^T
    const double *eps, *shape, *dts;
    const double2 *fw;   // [K][N_T+1][NP]
    const double2 *bw;   // [K][N_T+1][NP]
    const double *rho;   // [K]
    double2 *tg;         // [K][L][N_T]
    int *flags;
    unsigned long long *stats;  // [8] += series order summed over cells
    int K, L, N_T, hc_per_traj, cells_per_block, max_order;
    double tol;
    // sub-stepping of the derivative series (gradient_method = :gradgen only): 2-norm estimates of H0_k ([rb_k]) and of
    // the control operators ([Kc][L]), threshold on ||H_kn||_2 dt per sub-step; sub_theta <= 0 switches it off
    const double *rb;
    int rb_k;
    double sub_theta;
    // nullptr or a counter (flags[3]: batches that need sub-steps): the launch ends at once while it is zero -- deriv3_kernel
    // (grape_deriv3.hip.h) has done every cell; non-zero: this kernel does all of them
    const int *only_if;
};

// Number of sub-steps of the derivative series of one cell.  The series of exp(-i G dt) on the extended vector is
// summed unscaled; its rounding error grows like eps e^(||H|| dt), so for ||H_kn||_2 dt above theta the interval is cut
// into m equal parts and the series is applied m times to the extended vector (exp(X) = exp(X/m)^m: the gradient
// slots are simply not reset in between) -- the vector analogue of scaling and squaring, which is what keeps the
// reference's gradient-generator route accurate for any norm.  The bound uses the operator norm estimates of
// grape_create: ||H_kn|| <= ||H0_k|| + sum_l |eps_nl S_ln| ||H_l||.
__device__ __forceinline__ int deriv_substeps(const double *rb, int rb_k, double theta, int k, int hc_per_traj, int L,
                                              const double *eps, const double *shape, int N_T, int n, double dt) {
    if (!rb || !(theta > 0.0)) return 1;
    const double *rl = rb + rb_k + (size_t)(hc_per_traj ? k : 0) * L;
    double b = rb[k];
    for (int l = 0; l < L; ++l)
        b += fabs(eps[(size_t)l * N_T + n] * (shape ? shape[(size_t)l * N_T + n] : 1.0)) * rl[l];
    b *= fabs(dt);
    const int m = (int)ceil(b / theta);
    return m < 1 ? 1 : (m > 4096 ? 4096 : m);
}

// ---- wavefront-level reductions on DPP (no LDS traffic) ----
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
#define DPP_QUAD_XOR1 0xB1       // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E       // quad_perm [2,3,0,1]
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140

// sum over aligned groups of G adjacent lanes (G = 1, 2, 4, 8, 16); every lane of the group gets the sum
template <int G>
__device__ __forceinline__ double group_sum(double v) {
    if (G >= 2) v += dpp_f64<DPP_QUAD_XOR1>(v);
    if (G >= 4) v += dpp_f64<DPP_QUAD_XOR2>(v);
    if (G >= 8) v += dpp_f64<DPP_ROW_HALF_MIRROR>(v);
    if (G >= 16) v += dpp_f64<DPP_ROW_MIRROR>(v);
    return v;
}
// sum over all 64 lanes, result uniform (4 row sums combined through SGPRs)
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v = group_sum<16>(v);
    return readlane_f64(v, 0) + readlane_f64(v, 16) + readlane_f64(v, 32) + readlane_f64(v, 48);
}

// ---------------------------------------------------------------------------------------
// Sweeps for N <= 16 (NP = 16): ONE wave per trajectory and direction, no workgroup barrier in the time loop.
// A step of the general sweep_kernel costs two barriers and an LDS round trip for 256 threads of which 16 lanes
// carry data -- 1.1 us per step whatever the size; here lane (row r = lane >> 2, chunk c = lane & 3) holds four
// entries of U_n (forward: U[r][4c..4c+3]; backward: U[4c..4c+3][r], so that both directions are "4 products per
// lane, quad sum"), the four chunks of a row meet through two DPP quad permutes, and the state goes through 256 bytes
// of LDS that only this wave touches (a wave's LDS operations are ordered: no barrier).  The tile of step n+1 and
// n+2 is requested while step n is reduced.
// ---------------------------------------------------------------------------------------
#ifndef SWEEP1W32_DEPTH
#define SWEEP1W32_DEPTH 5   /* (X32 shape, forward + backward in one launch: 3 -> 1.012 ms, 4 -> 0.992, 5 -> 0.973; the 256-thread kernel 1.148) */
#endif
// Round 6: the same design at two tiles per side (NP = 32: lane (row r = lane >> 1, half c = lane & 1) holds SIXTEEN entries,
// a pair sum instead of a quad sum, three tiles in flight): the 256-thread kernel costs 1.1 us per step whatever the size.
template <int NP, bool BACKWARD>
__device__ __forceinline__ void sweep1w_body(const SweepArgs &a, const int k, double2 *xs /* [NP] LDS, this wave's */) {
    static_assert(NP == 16 || NP == 32, "one wave per trajectory: one or two tiles per side");
    constexpr int LPR = 64 / NP, EPL = NP / LPR;      // lanes per row, entries per lane
    const int lane = threadIdx.x & 63;
    const int r = lane / LPR, c = lane % LPR;
    const double2 *Uk = a.U + (size_t)(a.cls ? a.cls[k] : k) * a.N_T * NP * NP;
    double2 *st = a.store + (size_t)k * (a.N_T + 1) * NP;
    // ---- boundary state: lanes 0..15 hold element `lane` ----
    double rho = 1.0;
    {
        double2 v = make_double2(0., 0.);
        if (!BACKWARD) {
            if (lane < a.N) v = a.psi0[(size_t)k * a.N + lane];
        } else {
            if (lane < a.N) {
                v = chi_boundary(a, k, lane);
                if (a.xi) {   // chi_k(T) += lambda_b dt/2 xi_k(T)   (optimize.jl:856-866)
                    const double2 x_ = a.xi[((size_t)k * (a.N_T + 1) + a.N_T) * NP + lane];
                    const double cc = a.lambda_b * a.wq[a.N_T];
                    v.x += cc * x_.x; v.y += cc * x_.y;
                }
            }
            rho = sqrt(wave_sum(lane < NP ? v.x * v.x + v.y * v.y : 0.));
            if (lane == 0 && !a.unit_chi) {
                a.rho[k] = rho;
                if (rho < a.chi_min_norm) atomicOr(&a.flags[0], 2);
            }
            const double ir = rho > 0. ? 1.0 / rho : 0.;
            v.x *= ir; v.y *= ir;
        }
        if (lane < NP) {
            xs[lane] = v;
            st[(size_t)(BACKWARD ? a.N_T : 0) * NP + lane] = v;
        }
    }
    // tile loads: forward row r, columns 4c..4c+3 (64 contiguous bytes); backward rows 4c..4c+3 of column r
    auto load_tile = [&](double2 (&dst)[EPL], int step) __attribute__((always_inline)) {
        const int nn = BACKWARD ? a.N_T - 1 - step : step;
        const double2 *Un = Uk + (size_t)nn * NP * NP;
#pragma unroll
        for (int m = 0; m < EPL; ++m) dst[m] = BACKWARD ? Un[(EPL * c + m) * NP + r] : Un[r * NP + EPL * c + m];
    };
    // The tiles of the next DEPTH steps are in flight, in a ring of registers with STATIC indices (the time loop is
    // unrolled DEPTH times): shifting the ring with register moves makes every step wait for the newest load
    // (s_waitcnt vmcnt(0)), and the loop then runs at one memory latency per step whatever the depth.
    constexpr int DEPTH = NP == 16 ? 12 : SWEEP1W32_DEPTH;   // (C2 forward + backward: 8 -> 0.114 ms, 12 -> 0.108, 16 -> 0.111, 24 -> 0.120)
    double2 un[DEPTH][EPL];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < a.N_T) load_tile(un[d], d);
    double2 y = make_double2(0., 0.);
    for (int step0 = 0; step0 < a.N_T; step0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int step = step0 + d;
            if (step < a.N_T) {
                const int n = BACKWARD ? a.N_T - 1 - step : step;
                double pr = 0., pi = 0.;
#pragma unroll
                for (int m = 0; m < EPL; ++m) {
                    const double2 x = xs[EPL * c + m];
                    const double2 u = un[d][m];
                    if (!BACKWARD) {   // U x
                        pr = fma(u.x, x.x, pr); pr = fma(-u.y, x.y, pr);
                        pi = fma(u.x, x.y, pi); pi = fma(u.y, x.x, pi);
                    } else {           // conj(U) x
                        pr = fma(u.x, x.x, pr); pr = fma(u.y, x.y, pr);
                        pi = fma(u.x, x.y, pi); pi = fma(-u.y, x.x, pi);
                    }
                }
                if (step + DEPTH < a.N_T) load_tile(un[d], step + DEPTH);
                pr = group_sum<LPR>(pr);
                pi = group_sum<LPR>(pi);
                if (BACKWARD && a.xi && n > 0) {   // chi(t_n) += lambda_b Dt_n / rho_k xi_k(t_n)   (optimize.jl:897-908)
                    const double2 x_ = a.xi[((size_t)k * (a.N_T + 1) + n) * NP + r];
                    const double cc = a.lambda_b * a.wq[n] / rho;
                    pr += cc * x_.x; pi += cc * x_.y;
                }
                y = make_double2(pr, pi);
                if (c == 0) {
                    xs[r] = y;     // in order behind this wave's reads of the previous state
                    st[(size_t)(BACKWARD ? n : n + 1) * NP + r] = y;
                }
            }
        }
    }
    if (!BACKWARD) {
        // tau_k = <target_k | Psi_k(T)>  (optimize.jl:753)
        double pr = 0., pi = 0.;
        if (lane < a.N) {
            const double2 t = a.target[(size_t)k * a.N + lane];
            const double2 p = xs[lane];
            pr = t.x * p.x + t.y * p.y;
            pi = t.x * p.y - t.y * p.x;
        }
        pr = wave_sum(pr);
        pi = wave_sum(pi);
        if (lane == 0) a.tau[k] = make_double2(pr, pi);
    }
}

template <int NP, bool BACKWARD>
__global__ void __launch_bounds__(64) sweep1w_kernel(SweepArgs a) {
    __shared__ double2 xs[NP];
    sweep1w_body<NP, BACKWARD>(a, blockIdx.x, xs);
}
template <int NP>
__global__ void __launch_bounds__(64) sweep1w_pair_kernel(SweepArgs af, SweepArgs ab) {
    __shared__ double2 xs[NP];
    if ((int)blockIdx.x < af.K) sweep1w_body<NP, false>(af, blockIdx.x, xs);
    else sweep1w_body<NP, true>(ab, blockIdx.x - af.K, xs);
}


// ---------------------------------------------------------------------------------------
// Round 6 -- sweeps for N <= 16 as a PARALLEL SCAN over the time axis.  The recurrences Psi_n = U_n Psi_(n-1)
// (/root/reference/src/optimize.jl:731-738) and chi_(n-1) = U_n^dagger chi_n (:880-881) are latency chains: at C2 (32
// trajectories, 500 steps) sweep16_pair_kernel keeps 64 waves of a 256-CU chip busy for 500 x 0.22 us = 0.11 ms, 40 % of
// the evaluation.  Matrix products are associative, so the chain is cut into NB blocks of Bk steps:
//   1. scan16_block_kernel   F_b = U_(e-1) ... U_s of every block, all blocks at once (one wave each: a 16 x 16 x 16
//                            complex product per step on the matrix pipe, the result tile IS the next right operand);
//   2. the ordinary one-wave sweeps over the NB block propagators: Psi(t_e) = F_b Psi(t_s), chi(t_s) = F_b^dagger chi(t_e)
//      -- the backward block operator is the adjoint of the forward one, nothing else is formed; boundary states, rho_k
//      and tau_k come out of this pass exactly as from the fine sweep;
//   3. scan16_fill_kernel    the states inside every block from its boundary state, all blocks and both directions at
//                            once, into the storage arrays the derivative kernels read (workspace.jl:215).
// Chain length Bk + NB + Bk instead of N_T (16 + 32 + 16 at C2).  Results agree with the sequential order to rounding
// (a product of Bk unitary factors is formed before it is applied).  Not used with the running-cost inhomogeneity xi
// (it enters every fine step) and not for ensembles that fill the chip by themselves (phase 1 does 16 x the flops).
// ---------------------------------------------------------------------------------------
struct Scan16Args {
    const double2 *U;   // [KC][N_T][256]
    double2 *F;         // [KC][NB][256] block propagators, same layout
    int KC, N_T, Bk, NB;
};
__global__ void __launch_bounds__(64) scan16_block_kernel(Scan16Args a) {
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int kc = blockIdx.x / a.NB, b = blockIdx.x - kc * a.NB;
    const int s = b * a.Bk, e = min(a.N_T, s + a.Bk);
    const double2 *Un = a.U + ((size_t)kc * a.N_T + s) * 256 + j * 16 + g;   // lane (i = j, g): U[i][g], U[i][4 + g], U[i][8 + g], U[i][12 + g]
    // P = 1 in the result layout of the matrix instruction: register r of lane (j, g) is P[4r + g][j] -- which is exactly
    // the right operand of k-step r (k = 4r + g), so the result of a step feeds the next one without a move (Strip, above)
    d4 pr, pi;
#pragma unroll
    for (int r = 0; r < 4; ++r) { pr[r] = (4 * r + g == j) ? 1.0 : 0.0; pi[r] = 0.0; }
    constexpr int DEPTH = 4;
    double2 un[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (s + d < e) {
#pragma unroll
            for (int m = 0; m < 4; ++m) un[d][m] = Un[(size_t)d * 256 + 4 * m];
        }
    for (int n0 = s; n0 < e; n0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int n = n0 + d;
            if (n < e) {
                // k-step m covers k = 4m + g: left operand U[i][4m + g] (this lane's m-th loaded element), right operand
                // P[4m + g][j] = register m of the previous result
                d4 cr = {0., 0., 0., 0.}, ci = {0., 0., 0., 0.};
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const double ur = un[d][m].x, ui = un[d][m].y;
                    cr = MFMA64(ur, pr[m], cr);
                    ci = MFMA64(ur, pi[m], ci);
                    cr = MFMA64(-ui, pi[m], cr);
                    ci = MFMA64(ui, pr[m], ci);
                }
                pr = cr; pi = ci;
                if (n + DEPTH < e) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) un[d][m] = Un[(size_t)(n - s + DEPTH) * 256 + 4 * m];
                }
            }
        }
    }
    double2 *Fb = a.F + ((size_t)kc * a.NB + b) * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) Fb[(4 * r + g) * 16 + j] = make_double2(pr[r], pi[r]);
}

// phase 3: the fine steps of block b of trajectory k from the boundary state the coarse sweep left in cs[k][b] (forward:
// Psi(t_s)) / cs[k][b + 1] (backward: chi(t_e)); same lane layout and arithmetic as sweep16_body
struct Scan16FillArgs {
    const double2 *cfw, *cbw;   // [K][NB + 1][16] boundary states of the coarse sweeps (forward / backward)
    int Bk, NB, both;           // both: blocks [K NB, 2 K NB) of the grid are the backward blocks
};
template <int NP, bool BACKWARD>
__device__ __forceinline__ void scan1w_fill_body(const SweepArgs &a, const Scan16FillArgs &f, const int k, const int b, double2 *xs) {
    constexpr int LPR = 64 / NP, EPL = NP / LPR;
    const int lane = threadIdx.x & 63;
    const int r = lane / LPR, c = lane % LPR;
    const int s = b * f.Bk, e = min(a.N_T, s + f.Bk);
    const double2 *Uk = a.U + (size_t)(a.cls ? a.cls[k] : k) * a.N_T * NP * NP;
    double2 *st = a.store + (size_t)k * (a.N_T + 1) * NP;
    if (lane < NP) {
        const double2 v = BACKWARD ? f.cbw[((size_t)k * (f.NB + 1) + b + 1) * NP + lane] : f.cfw[((size_t)k * (f.NB + 1) + b) * NP + lane];
        xs[lane] = v;
        if (BACKWARD ? e == a.N_T : s == 0) st[(size_t)(BACKWARD ? a.N_T : 0) * NP + lane] = v;   // the boundary state of the trajectory itself
    }
    const int nsteps = e - s;
    auto load_tile = [&](double2 (&dst)[EPL], int t) __attribute__((always_inline)) {
        const int nn = BACKWARD ? e - 1 - t : s + t;
        const double2 *Un = Uk + (size_t)nn * NP * NP;
#pragma unroll
        for (int m = 0; m < EPL; ++m) dst[m] = BACKWARD ? Un[(EPL * c + m) * NP + r] : Un[r * NP + EPL * c + m];
    };
    constexpr int DEPTH = NP == 16 ? 8 : 3;
    double2 un[DEPTH][EPL];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < nsteps) load_tile(un[d], d);
    for (int t0 = 0; t0 < nsteps; t0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int t = t0 + d;
            if (t < nsteps) {
                const int n = BACKWARD ? e - 1 - t : s + t;
                double pr = 0., pi = 0.;
#pragma unroll
                for (int m = 0; m < EPL; ++m) {
                    const double2 x = xs[EPL * c + m];
                    const double2 u = un[d][m];
                    if (!BACKWARD) {
                        pr = fma(u.x, x.x, pr); pr = fma(-u.y, x.y, pr);
                        pi = fma(u.x, x.y, pi); pi = fma(u.y, x.x, pi);
                    } else {
                        pr = fma(u.x, x.x, pr); pr = fma(u.y, x.y, pr);
                        pi = fma(u.x, x.y, pi); pi = fma(-u.y, x.x, pi);
                    }
                }
                if (t + DEPTH < nsteps) load_tile(un[d], t + DEPTH);
                pr = group_sum<LPR>(pr);
                pi = group_sum<LPR>(pi);
                if (c == 0) {
                    const double2 y = make_double2(pr, pi);
                    xs[r] = y;
                    st[(size_t)(BACKWARD ? n : n + 1) * NP + r] = y;
                }
            }
        }
    }
}
template <int NP>
__global__ void __launch_bounds__(64) scan1w_fill_kernel(SweepArgs af, SweepArgs ab, Scan16FillArgs f, int backward_only) {
    __shared__ double2 xs[NP];
    const int nfw = backward_only ? 0 : af.K * f.NB;
    const int id = blockIdx.x;
    if (id < nfw) scan1w_fill_body<NP, false>(af, f, id / f.NB, id % f.NB, xs);
    else { const int q = id - nfw; scan1w_fill_body<NP, true>(ab, f, q / f.NB, q % f.NB, xs); }
}


// ---------------------------------------------------------------------------------------
// Round 6 -- the same scan for 17 <= N <= 64 (NT = 2 .. 4 tiles per side), for the problems the reference itself is used
// on: FEW trajectories.  There the sweeps are one latency chain per trajectory and direction (1.1 us per step at N = 32,
// 1.8 us at N = 48 / 64, whatever the size: two barriers and an LDS round trip) on a chip that is otherwise idle -- 60-76 %
// of an evaluation at K = 4, N_T = 1000.  Phase 1: one workgroup of NT waves per block multiplies the block propagator
// from the left by U_n, step by step (wave w owns column strip w of F: in the C/D layout that strip is exactly the right
// operand of the product, U_n is staged through the LDS as the left operand of the tile engine gemm_xb3: 3M arithmetic,
// 12 NT^2 matrix instructions per wave and step); phase 2: the ordinary sweeps over the block propagators; phase 3:
// scan_fill_kernel, all blocks and both directions at once.
// ---------------------------------------------------------------------------------------
struct ScanArgs {
    const double2 *U;   // [KC][N_T][NP*NP] row-major interleaved
    double2 *F;         // [KC][NB][NP*NP] block propagators, same layout
    int KC, N_T, Bk, NB;
};
template <int NT>
__global__ void __launch_bounds__(64 * NT) scan_block_kernel(ScanArgs a) {
    constexpr int NP = 16 * NT, LD = NP + 2, NTH = 64 * NT, EPT = NP * NP / NTH;   // elements (double2) per thread of a matrix
    extern __shared__ double scan_lds[];   // two stages of [re | im][NP][LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int kc = blockIdx.x / a.NB, b = blockIdx.x - kc * a.NB;
    const int s = b * a.Bk, e = min(a.N_T, s + a.Bk);
    const double2 *Un = a.U + ((size_t)kc * a.N_T + s) * NP * NP;
    // F = 1: register r of tile t of lane (j, g) is F[16 t + 4 r + g][16 wave + j]
    Strip<NT> F;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            F.re[t][r] = (16 * t + 4 * r + g == 16 * wave + j) ? 1.0 : 0.0;
            F.im[t][r] = 0.0;
        }
    auto stage = [&](int st) { return scan_lds + (size_t)st * 2 * NP * LD; };
    double2 pre[EPT];
    auto fetch = [&](int n) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) pre[q] = Un[(size_t)(n - s) * NP * NP + q * NTH + tid];
    };
    auto commit = [&](int st) __attribute__((always_inline)) {
        double *xr = stage(st), *xi = xr + NP * LD;
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int idx = q * NTH + tid, row = idx / NP, col = idx - row * NP;
            xr[row * LD + col] = pre[q].x;
            xi[row * LD + col] = pre[q].y;
        }
    };
    fetch(s);
    commit(0);
    __syncthreads();
    for (int n = s; n < e; ++n) {
        const int cur = (n - s) & 1;
        if (n + 1 < e) fetch(n + 1);                  // (in flight under the product)
        Strip3<NT> q;
        strip3_zero(q);
        const double *xr = stage(cur);
        gemm_xb3<NT, LD>(q, xr, xr + NP * LD, F, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            F.re[t] = q.p1[t] - q.p2[t];
            F.im[t] = q.p3[t] - q.p1[t] - q.p2[t];
        }
        if (n + 1 < e) commit(cur ^ 1);               // (the other stage: nobody reads it during this step)
        __syncthreads();
    }
    double2 *Fb = a.F + ((size_t)kc * a.NB + b) * NP * NP;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Fb[(size_t)(16 * t + 4 * r + g) * NP + 16 * wave + j] = make_double2(F.re[t][r], F.im[t][r]);
}

// phase 3 for 17 <= N <= 64: block b of trajectory k from the boundary state of the coarse sweep; thread map and arithmetic of
// sweep_body (one workgroup per block and direction)
struct ScanFillArgs {
    const double2 *cfw, *cbw;   // [K][NB + 1][NP]
    int Bk, NB;
};
template <int NP, bool BACKWARD>
__device__ __forceinline__ void scan_fill_body(const SweepArgs &a, const ScanFillArgs &f, const int k, const int b) {
    constexpr int NW = NP == 48 ? 3 : 4, RW = NP / NW;
    __shared__ double2 x[2][NP];
    __shared__ double2 part[NW][NP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = b * f.Bk, e = min(a.N_T, s + f.Bk), nsteps = e - s;
    const double2 *Uk = a.U + (size_t)(a.cls ? a.cls[k] : k) * a.N_T * NP * NP;
    double2 *st = a.store + (size_t)k * (a.N_T + 1) * NP;
    if (tid < NP) {
        const double2 v = BACKWARD ? f.cbw[((size_t)k * (f.NB + 1) + b + 1) * NP + tid] : f.cfw[((size_t)k * (f.NB + 1) + b) * NP + tid];
        x[0][tid] = v;
        if (BACKWARD ? e == a.N_T : s == 0) st[(size_t)(BACKWARD ? a.N_T : 0) * NP + tid] = v;
    }
    __syncthreads();
    int cur = 0;
    auto load_tile = [&](double2 (&dst)[RW], int t) __attribute__((always_inline)) {
        const int nn = BACKWARD ? e - 1 - t : s + t;
        const double2 *Un = Uk + (size_t)nn * NP * NP;
#pragma unroll
        for (int r = 0; r < RW; ++r) dst[r] = lane < NP ? Un[(size_t)(wave * RW + r) * NP + lane] : make_double2(0., 0.);
    };
    constexpr int D = NP == 64 ? 2 : NP == 48 ? 4 : 6;
    double2 un[D][RW];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < nsteps) load_tile(un[d], d);
    for (int t0 = 0; t0 < nsteps; t0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int t = t0 + d;
        if (t >= nsteps) break;
        const int n = BACKWARD ? e - 1 - t : s + t;
        double2 (&ucur)[RW] = un[d];
        if (!BACKWARD) {
            const double2 xv = lane < NP ? x[cur][lane] : make_double2(0., 0.);
            double pr[RW], pi[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const double2 u = ucur[r];
                pr[r] = u.x * xv.x - u.y * xv.y;
                pi[r] = u.x * xv.y + u.y * xv.x;
            }
            if (t + D < nsteps) load_tile(un[d], t + D);
            int row = 0;
            rs_step<RW / 2, 32>(pr, pi, lane, row);
            if constexpr (RW >= 4) rs_step<RW / 4, 16>(pr, pi, lane, row);
            if constexpr (RW >= 8) rs_step<RW / 8, 8>(pr, pi, lane, row);
            if constexpr (RW >= 16) rs_step<RW / 16, 4>(pr, pi, lane, row);
#pragma unroll
            for (int off = 32 / RW; off >= 1; off >>= 1) {
                pr[0] += __shfl_xor(pr[0], off, 64);
                pi[0] += __shfl_xor(pi[0], off, 64);
            }
            if ((lane & (64 / RW - 1)) == 0) x[cur ^ 1][wave * RW + row] = make_double2(pr[0], pi[0]);
            __syncthreads();
            if (tid < NP) st[(size_t)(n + 1) * NP + tid] = x[cur ^ 1][tid];
        } else {
            double ar = 0., ai = 0.;
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int i = wave * RW + r;
                const double2 u = ucur[r];
                const double2 xi = x[cur][i];
                ar += u.x * xi.x + u.y * xi.y;
                ai += u.x * xi.y - u.y * xi.x;
            }
            if (t + D < nsteps) load_tile(un[d], t + D);
            if (lane < NP) part[wave][lane] = make_double2(ar, ai);
            __syncthreads();
            if (tid < NP) {
                double2 sum = part[0][tid];
#pragma unroll
                for (int q = 1; q < NW; ++q) { sum.x += part[q][tid].x; sum.y += part[q][tid].y; }
                x[cur ^ 1][tid] = sum;
                st[(size_t)n * NP + tid] = sum;
            }
        }
        __syncthreads();
        cur ^= 1;
      }
    }
}
template <int NP>
__global__ void __launch_bounds__(NP == 48 ? 192 : 256) scan_fill_kernel(SweepArgs af, SweepArgs ab, ScanFillArgs f, int backward_only) {
    const int nfw = backward_only ? 0 : af.K * f.NB;
    const int id = blockIdx.x;
    if (id < nfw) scan_fill_body<NP, false>(af, f, id / f.NB, id % f.NB);
    else { const int q = id - nfw; scan_fill_body<NP, true>(ab, f, q / f.NB, q % f.NB); }
}


template <int NP, int LMAX, int NTH>
__global__ void __launch_bounds__(NTH) deriv_kernel(DerivArgs a) {
    constexpr int NCH = NTH / NP;      // column chunks per row = adjacent lanes (<= 16)
    constexpr int CW = NP / NCH;       // columns per thread
    constexpr int NV = 1 + LMAX;       // vectors: pw = Hd^(m-1) chi, phi_1..phi_L
    constexpr int NW = NTH / 64;       // waves per block
    constexpr int PAD = CW > 1 ? 1 : 0;  // one double2 of padding per chunk: the NCH broadcast reads of a
    constexpr int VLEN = NP + NCH * PAD; // wave-instruction then fall on different LDS banks
    static_assert(NCH <= 16 && NCH * CW == NP, "row chunks must be adjacent lanes of one DPP row");
    __shared__ double2 vec[2][NV][VLEN];   // ping-pong over series orders
    __shared__ double red[2][NW][LMAX];    // per wave: ||phi_l||^2 of its rows
    __shared__ double redc[2][NW];         // per wave: ||pw||^2 of its rows (sub-stepped cells only)
    __shared__ double2 gsum[NW][LMAX];     // per wave: final <chi'_l | psi> of its rows
    if (a.only_if && *a.only_if == 0) return;   // (uniform)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = tid % NCH, i = tid / NCH;  // column chunk, row
    const int L = a.L;
    const int nblk_per_k = (a.N_T + a.cells_per_block - 1) / a.cells_per_block;
    const int k = blockIdx.x / nblk_per_k;
    const int n0 = (blockIdx.x - k * nblk_per_k) * a.cells_per_block;
    const int n1 = min(a.N_T, n0 + a.cells_per_block);
    const int vi = i + (i / CW) * PAD;       // LDS index of row i
    const int vq = q * (CW + PAD);           // LDS index of this thread's first column

    // register tiles (coalesced: the chunks of a row are adjacent lanes):
    //   hr/hi  = Hd[i][q*CW + c] = (H0_k + sum_l e_l mu_l)^dagger, updated incrementally from cell to cell
    //   mur/mui = mu_l^dagger[i][q*CW + c]
    double hr[CW], hi[CW], mur[LMAX][CW], mui[LMAX][CW];
    {
        const double *h0 = a.H0t + (size_t)k * 2 * NP * NP;
        const double *hc = a.Hct + (size_t)(a.hc_per_traj ? k : 0) * L * 2 * NP * NP;
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const int idx = i * NP + q * CW + c;
            hr[c] = h0[idx];
            hi[c] = -h0[NP * NP + idx];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {
                mur[l][c] = l < L ? hc[(size_t)l * 2 * NP * NP + idx] : 0.;
                mui[l][c] = l < L ? -hc[(size_t)l * 2 * NP * NP + NP * NP + idx] : 0.;
            }
        }
    }
    const double rho = a.rho[k];
    const int all_done = (1 << L) - 1;
    double eprev[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) eprev[l] = 0.;

    for (int n = n0; n < n1; ++n) {
        const double dt = a.dts[n];
        double sh[LMAX];
        // Hd_n = Hd_{n-1} + sum_l (e_nl - e_{n-1,l}) mu_l^dagger   (first cell of the block: e_{n-1} = 0)
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            sh[l] = (l < L && a.shape) ? a.shape[(size_t)l * a.N_T + n] : 1.0;
            const double e = l < L ? a.eps[(size_t)l * a.N_T + n] * sh[l] : 0.;
            const double de = e - eprev[l];
            eprev[l] = e;
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                hr[c] = fma(de, mur[l][c], hr[c]);
                hi[c] = fma(de, mui[l][c], hi[c]);
            }
        }
        // row owner (q == 0) keeps Psi_k(t_{n-1}) and its element of chi_k(t_n) and of the accumulated chi'_l
        double2 psi = make_double2(0., 0.), chi_cur = make_double2(0., 0.);
        if (q == 0) {
            psi = a.fw[((size_t)k * (a.N_T + 1) + n) * NP + i];
            chi_cur = a.bw[((size_t)k * (a.N_T + 1) + n + 1) * NP + i];
        }
        // series of exp(-i G dtb) on the extended vector, dtb = -dt: alpha_m = (i dt)^m / m!
        // row owners accumulate chi'_l[i] = sum_m alpha_m phi_m^l[i]  (the accumulation of taylor_grad_step!)
        double outr[LMAX], outi[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { outr[l] = 0.; outi[l] = 0.; }
        // sub-steps (see deriv_substeps): the extended vector (chi'_1..chi'_L, chi) of sub-step s starts from the result
        // of sub-step s-1; with one sub-step this is exactly the single series of the reference
        const int nsub = deriv_substeps(a.rb, a.rb_k, a.sub_theta, k, a.hc_per_traj, L, a.eps, a.shape, a.N_T, n, dt);
        const double dts_ = dt / (double)nsub;
        const int all_bits = nsub > 1 ? (all_done | (1 << LMAX)) : all_done;   // bit LMAX: the chi chain itself
        int converged = 1, m_used = 0;
        for (int sub = 0; sub < nsub; ++sub) {
            __syncthreads();  // previous cell / sub-step is done with vec[*] / gsum
            if (q == 0) {
                vec[0][0][vi] = chi_cur;
#pragma unroll
                for (int l = 0; l < LMAX; ++l) vec[0][1 + l][vi] = make_double2(outr[l], outi[l]);
            }
            __syncthreads();
            double ocr = 0., oci = 0.;    // sum_m alpha_m pw_m (row owner): chi after the sub-step
            double alr = 0., ali = dts_;  // alpha_1 = i dt
            int done_mask = 0, conv_sub = 0, m_sub = a.max_order, cur = 0;
            for (int m = 1; m <= a.max_order; ++m) {
                // ---- this thread's partial products over its CW columns ----
                double pr = 0., pi = 0., gr[LMAX], gi[LMAX], ur[LMAX], ui[LMAX];
#pragma unroll
                for (int l = 0; l < LMAX; ++l) { gr[l] = 0.; gi[l] = 0.; ur[l] = 0.; ui[l] = 0.; }
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    const double2 x0 = vec[cur][0][vq + c];
                    pr = fma(hr[c], x0.x, pr);   // Hd * pw
                    pr = fma(-hi[c], x0.y, pr);
                    pi = fma(hr[c], x0.y, pi);
                    pi = fma(hi[c], x0.x, pi);
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) {
                        const double2 xl = vec[cur][1 + l][vq + c];
                        ur[l] = fma(mur[l][c], x0.x, ur[l]);   // mu_l^d * pw
                        ur[l] = fma(-mui[l][c], x0.y, ur[l]);
                        ui[l] = fma(mur[l][c], x0.y, ui[l]);
                        ui[l] = fma(mui[l][c], x0.x, ui[l]);
                        gr[l] = fma(hr[c], xl.x, gr[l]);       // Hd * phi_l
                        gr[l] = fma(-hi[c], xl.y, gr[l]);
                        gi[l] = fma(hr[c], xl.y, gi[l]);
                        gi[l] = fma(hi[c], xl.x, gi[l]);
                    }
                }
                // ---- sum the NCH chunks of each row (adjacent lanes, DPP) ----
                pr = group_sum<NCH>(pr);
                pi = group_sum<NCH>(pi);
                const int nxt = cur ^ 1;
                if (q == 0) vec[nxt][0][vi] = make_double2(pr, pi);
                if (nsub > 1) {   // the chi chain is part of the result of a sub-step
                    double nn = 0.;
                    if (q == 0) {
                        ocr += alr * pr - ali * pi;
                        oci += alr * pi + ali * pr;
                        nn = pr * pr + pi * pi;
                    }
                    nn = wave_sum_dpp(nn);
                    if (lane == 0) redc[nxt][wave] = nn;
                }
#pragma unroll
                for (int l = 0; l < LMAX; ++l) {
                    // phi_l(new) = S_l mu_l^d pw + Hd phi_l
                    const double fr = group_sum<NCH>(fma(sh[l], ur[l], gr[l]));
                    const double fi = group_sum<NCH>(fma(sh[l], ui[l], gi[l]));
                    double nn = 0.;
                    if (q == 0) {
                        vec[nxt][1 + l][vi] = make_double2(fr, fi);
                        if (!((done_mask >> l) & 1)) {
                            outr[l] += alr * fr - ali * fi;   // += alpha_m * phi_m
                            outi[l] += alr * fi + ali * fr;
                        }
                        nn = fr * fr + fi * fi;
                    }
                    // ||phi_l||^2 over this wave's rows (wavefront reduction); combined after the barrier
                    nn = wave_sum_dpp(nn);
                    if (lane == 0) red[nxt][wave][l] = nn;
                }
                __syncthreads();
                const double al2 = alr * alr + ali * ali;
#pragma unroll
                for (int l = 0; l < LMAX; ++l) {
                    if (l < L) {
                        double nn = 0.;
#pragma unroll
                        for (int w = 0; w < NW; ++w) nn += red[nxt][w][l];
                        // reference stopping rule: |alpha_m| * ||phi_m|| < tolerance (optimize.jl:631-636)
                        if (m >= 2 && al2 * nn < a.tol * a.tol) done_mask |= 1 << l;
                    }
                }
                if (nsub > 1) {
                    double nn = 0.;
#pragma unroll
                    for (int w = 0; w < NW; ++w) nn += redc[nxt][w];
                    if (m >= 2 && al2 * nn < a.tol * a.tol) done_mask |= 1 << LMAX;
                }
                cur = nxt;
                if (done_mask == all_bits) { conv_sub = 1; m_sub = m; break; }
                {   // alpha_{m+1} = alpha_m * (i dt) / (m+1)
                    const double f = dts_ / (double)(m + 1);
                    const double nr = -ali * f, ni = alr * f;
                    alr = nr; ali = ni;
                }
            }
            converged &= conv_sub;
            m_used += m_sub;
            chi_cur.x += ocr; chi_cur.y += oci;
        }   // sub-steps
        // tau_grads[k][l][n] = rho_k * <chi'_l | psi> = rho_k sum_i conj(chi'_l[i]) psi[i]
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            double orr = outr[l] * psi.x + outi[l] * psi.y;
            double oi = outr[l] * psi.y - outi[l] * psi.x;
            orr = wave_sum_dpp(orr);
            oi = wave_sum_dpp(oi);
            if (lane == 0) gsum[wave][l] = make_double2(orr, oi);
        }
        __syncthreads();
        if (tid < L) {
            double gr_ = 0., gi_ = 0.;
            for (int w = 0; w < NW; ++w) { gr_ += gsum[w][tid].x; gi_ += gsum[w][tid].y; }
            a.tg[((size_t)k * L + tid) * a.N_T + n] = make_double2(rho * gr_, rho * gi_);
        }
        if (tid == 0) {
            if (!converged) atomicOr(&a.flags[0], 4);
            stat_add(a.stats, 8, (unsigned long long)m_used);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Kernel 5b: the same per-cell derivative overlaps on fp64 MFMA, batched over 16 cells.
//
// The series recursion of deriv_kernel multiplies H_kn^dagger = H0_k^dagger + sum_l e_nl mu_l^dagger with
// vectors; with the 16 consecutive cells n0..n0+15 of one trajectory as the 16 MFMA columns, the
// cell-dependent part moves into per-column scalars (e_cl multiplies the B operand), so that every
// product is a *fixed* matrix (H0^dagger or mu_l^dagger) times an NP x 16 block of vectors:
//     pw'    = H0d pw + sum_l e_l (mud_l pw)
//     phi_l' = H0d phi_l + sum_l' mud_l' (e_l' phi_l) + S_l (mud_l pw)          ((1+L)^2 products)
// One wave per 16-row tile (block = NP/16 waves).  A operands come from fragment-packed copies of the
// operators (one coalesced 512-B load per tile, plane and k-step; cached in registers when they fit),
// B operands are the current vectors, kept in a block-private global scratch [2][1+L][2][NP][16]
// (L1/L2 resident) and ping-ponged over the series orders; accumulators are directly the new vectors.
// Column norms / overlaps: in-lane over the 4 accumulator registers, 2 wavefront shuffles across the
// row groups, then fp64 LDS atomics across the waves.
// ---------------------------------------------------------------------------------------
struct DerivMfmaArgs {
    const double *H0p;   // [K][RT][KS][2][64] fragment-packed conj-transposed drift (RT = NP/16, KS = NP/4)
    const double *Hcp;   // [Kc][L][RT][KS][2][64]
    const double *eps, *shape, *dts;
    const double2 *fw, *bw;
    const double *rho;
    double2 *tg;
    double *vecs;        // [gridDim.x][2][1+L][2][NP][16] scratch
    int *flags;
    unsigned long long *stats;
    int K, L, N_T, hc_per_traj, max_order, nbatch_total, batches_per_k;
    double tol;
    // batches that deriv_sub_kernel redoes (see Deriv2Args::batch_flag)
    const int *batch_flag;
};

template <int NP, int LMAX, bool CACHE_A, bool VEC_LDS>
__global__ void __launch_bounds__((NP / 16 <= 8 ? NP / 16 : 8) * 64) deriv_mfma_kernel(DerivMfmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) double dsm[];   // VEC_LDS: the ping-pong vectors
    constexpr int RT = NP / 16, KS = NP / 4;
    constexpr int NW = RT <= 8 ? RT : 8;      // waves per block
    constexpr int TPW = RT / NW;              // row tiles per wave (processed one after the other)
    constexpr int NTH = NW * 64;
    constexpr int NV = 1 + LMAX;
    static_assert(RT % NW == 0, "row tiles must divide evenly over the waves");
    static_assert(!CACHE_A || TPW == 1, "operand caching only for one tile per wave");
    // per wave and column: Re/Im <phi_l|psi>, ||phi_l||^2 of the wave's rows; summed in a fixed order
    // after the barrier (bitwise reproducible, unlike LDS atomics); double-buffered over the orders
    __shared__ double red[2][NW][LMAX][16][3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, rg = lane >> 4;
    const int L = a.L;
    const size_t vplane = (size_t)NP * 16;                      // doubles per vector plane
    double *vbase;
    if constexpr (VEC_LDS) vbase = dsm; else vbase = a.vecs + (size_t)blockIdx.x * 2 * NV * 2 * vplane;

    int cached_k = -1;
    double car[CACHE_A ? NV : 1][CACHE_A ? KS : 1], cai[CACHE_A ? NV : 1][CACHE_A ? KS : 1];

    for (int batch = blockIdx.x; batch < a.nbatch_total; batch += gridDim.x) {
        const int k = batch / a.batches_per_k;
        const int n0 = (batch - k * a.batches_per_k) * 16;
        const int n = n0 + c;
        const bool valid = n < a.N_T;
        const int nc = valid ? n : a.N_T - 1;
        const double *h0k = a.H0p + (size_t)k * RT * KS * 128;
        const double *hck = a.Hcp + (size_t)(a.hc_per_traj ? k : 0) * L * RT * KS * 128;
        if (CACHE_A && cached_k != k) {
#pragma unroll
            for (int ks = 0; ks < (CACHE_A ? KS : 1); ++ks) {
                car[0][ks] = h0k[((size_t)wave * KS + ks) * 128 + lane];
                cai[0][ks] = h0k[((size_t)wave * KS + ks) * 128 + 64 + lane];
#pragma unroll
                for (int l = 0; l < LMAX; ++l) {
                    car[1 + l][ks] = l < L ? hck[(((size_t)l * RT + wave) * KS + ks) * 128 + lane] : 0.;
                    cai[1 + l][ks] = l < L ? hck[(((size_t)l * RT + wave) * KS + ks) * 128 + 64 + lane] : 0.;
                }
            }
            cached_k = k;
        }
        // per-column scalars
        const double dt = a.dts[nc];
        double e[LMAX], sh[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            sh[l] = (l < L && a.shape) ? a.shape[(size_t)l * a.N_T + nc] : 1.0;
            e[l] = l < L ? a.eps[(size_t)l * a.N_T + nc] * sh[l] : 0.;
        }
        const double rho = a.rho[k];
        // psi tiles (rows 16rt + 4r + rg of column c) and initial vectors: pw_0 = chi_k(t_{n+1})
        double psr[TPW][4], psi_[TPW][4];
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (wave + tt * NW) + 4 * r + rg;
                const double2 p = a.fw[((size_t)k * (a.N_T + 1) + nc) * NP + row];
                const double2 x = a.bw[((size_t)k * (a.N_T + 1) + nc + 1) * NP + row];
                psr[tt][r] = p.x; psi_[tt][r] = p.y;
                vbase[0 * vplane + (size_t)row * 16 + c] = valid ? x.x : 0.;
                vbase[1 * vplane + (size_t)row * 16 + c] = valid ? x.y : 0.;
            }
        __syncthreads();
        double accr[LMAX], acci[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { accr[l] = 0.; acci[l] = 0.; }
        double alr = 0., ali = dt;              // alpha_1 = i dt  (dtb = -dt)
        const int all_done = (1 << L) - 1;
        int done_mask = valid ? 0 : all_done;
        int cur = 0, m_used = a.max_order, converged = 0;
        for (int m = 1; m <= a.max_order; ++m) {
            const double *vc = vbase + (size_t)cur * NV * 2 * vplane;
            double *vn = vbase + (size_t)(cur ^ 1) * NV * 2 * vplane;
            const int slot = m & 1;
            double sor[LMAX], soi[LMAX], snn[LMAX];   // this wave's column sums over its row tiles
#pragma unroll
            for (int l = 0; l < LMAX; ++l) { sor[l] = 0.; soi[l] = 0.; snn[l] = 0.; }
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) {
                const int rt = wave + tt * NW;
                const double *h0p = h0k + (size_t)rt * KS * 128;
                d4 pwr = {0., 0., 0., 0.}, pwi = {0., 0., 0., 0.};
                d4 mur[LMAX], mui[LMAX], phr[LMAX], phi[LMAX];
#pragma unroll
                for (int l = 0; l < LMAX; ++l) {
                    mur[l] = (d4){0., 0., 0., 0.}; mui[l] = (d4){0., 0., 0., 0.};
                    phr[l] = (d4){0., 0., 0., 0.}; phi[l] = (d4){0., 0., 0., 0.};
                }
                auto ks_step = [&](const int ks) __attribute__((always_inline)) {
                    // A operands (row tile rt): H0d and mud_l
                    double ar[NV], ai[NV];
                    if (CACHE_A) {
#pragma unroll
                        for (int v = 0; v < NV; ++v) { ar[v] = car[v][CACHE_A ? ks : 0]; ai[v] = cai[v][CACHE_A ? ks : 0]; }
                    } else {
                        ar[0] = h0p[(size_t)ks * 128 + lane];
                        ai[0] = h0p[(size_t)ks * 128 + 64 + lane];
#pragma unroll
                        for (int l = 0; l < LMAX; ++l) {
                            ar[1 + l] = l < L ? hck[(((size_t)l * RT + rt) * KS + ks) * 128 + lane] : 0.;
                            ai[1 + l] = l < L ? hck[(((size_t)l * RT + rt) * KS + ks) * 128 + 64 + lane] : 0.;
                        }
                    }
                    // B operands: rows 4ks + rg of the current vectors, column c
                    const size_t bo = (size_t)(4 * ks + rg) * 16 + c;
                    const double bwr = vc[0 * vplane + bo], bwi = vc[1 * vplane + bo];
                    pwr = MFMA64(ar[0], bwr, pwr);  pwi = MFMA64(ar[0], bwi, pwi);
                    pwr = MFMA64(ai[0], -bwi, pwr); pwi = MFMA64(ai[0], bwr, pwi);
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) {
                        mur[l] = MFMA64(ar[1 + l], bwr, mur[l]);  mui[l] = MFMA64(ar[1 + l], bwi, mui[l]);
                        mur[l] = MFMA64(ai[1 + l], -bwi, mur[l]); mui[l] = MFMA64(ai[1 + l], bwr, mui[l]);
                    }
                    if (m > 1) {
#pragma unroll
                        for (int l = 0; l < LMAX; ++l) {
                            const double br = vc[((1 + l) * 2 + 0) * vplane + bo], bi = vc[((1 + l) * 2 + 1) * vplane + bo];
                            phr[l] = MFMA64(ar[0], br, phr[l]);  phi[l] = MFMA64(ar[0], bi, phi[l]);
                            phr[l] = MFMA64(ai[0], -bi, phr[l]); phi[l] = MFMA64(ai[0], br, phi[l]);
#pragma unroll
                            for (int l2 = 0; l2 < LMAX; ++l2) {
                                const double sr = e[l2] * br, si = e[l2] * bi;   // column-scaled B operand
                                phr[l] = MFMA64(ar[1 + l2], sr, phr[l]);  phi[l] = MFMA64(ar[1 + l2], si, phi[l]);
                                phr[l] = MFMA64(ai[1 + l2], -si, phr[l]); phi[l] = MFMA64(ai[1 + l2], sr, phi[l]);
                            }
                        }
                    }
                };
                if constexpr (CACHE_A) {   // fully unrolled: the cached operand arrays need static indices
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) ks_step(ks);
                } else {
#pragma unroll 4
                    for (int ks = 0; ks < KS; ++ks) ks_step(ks);
                }
                // ---- new vectors (accumulator layout: row 16rt + 4r + rg, column c) ----
#pragma unroll
                for (int l = 0; l < LMAX; ++l) {
                    pwr += e[l] * mur[l];
                    pwi += e[l] * mui[l];
                    phr[l] += sh[l] * mur[l];
                    phi[l] += sh[l] * mui[l];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t o = (size_t)(16 * rt + 4 * r + rg) * 16 + c;
                    vn[0 * vplane + o] = pwr[r];
                    vn[1 * vplane + o] = pwi[r];
                }
#pragma unroll
                for (int l = 0; l < LMAX; ++l) {
                    double orr = 0., oi = 0., nn = 0.;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const size_t o = (size_t)(16 * rt + 4 * r + rg) * 16 + c;
                        vn[((1 + l) * 2 + 0) * vplane + o] = phr[l][r];
                        vn[((1 + l) * 2 + 1) * vplane + o] = phi[l][r];
                        orr += phr[l][r] * psr[tt][r] + phi[l][r] * psi_[tt][r];   // conj(phi) * psi
                        oi += phr[l][r] * psi_[tt][r] - phi[l][r] * psr[tt][r];
                        nn += phr[l][r] * phr[l][r] + phi[l][r] * phi[l][r];
                    }
                    // the 4 row groups of a column sit 16 lanes apart
                    orr += __shfl_xor(orr, 16, 64); oi += __shfl_xor(oi, 16, 64); nn += __shfl_xor(nn, 16, 64);
                    orr += __shfl_xor(orr, 32, 64); oi += __shfl_xor(oi, 32, 64); nn += __shfl_xor(nn, 32, 64);
                    sor[l] += orr; soi[l] += oi; snn[l] += nn;
                }
            }
            if (lane < 16) {
#pragma unroll
                for (int l = 0; l < LMAX; ++l) {
                    red[slot][wave][l][c][0] = sor[l];
                    red[slot][wave][l][c][1] = soi[l];
                    red[slot][wave][l][c][2] = snn[l];
                }
            }
            __syncthreads();
            const double al2 = alr * alr + ali * ali;
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {
                if (l < L && !((done_mask >> l) & 1)) {
                    double orr = 0., oi = 0., nn = 0.;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        orr += red[slot][w][l][c][0]; oi += red[slot][w][l][c][1]; nn += red[slot][w][l][c][2];
                    }
                    accr[l] += alr * orr + ali * oi;   // conj(alpha) <phi|psi>
                    acci[l] += alr * oi - ali * orr;
                    if (m >= 2 && al2 * nn < a.tol * a.tol) done_mask |= 1 << l;
                }
            }
            cur ^= 1;
            // all 16 columns converged?  (identical decision in every wave: same LDS values)
            if (__all(done_mask == all_done)) { converged = 1; m_used = m; break; }
            {
                const double f = dt / (double)(m + 1);
                const double nr = -ali * f, ni = alr * f;
                alr = nr; ali = ni;
            }
        }
        if (tid < 16 && valid) {
            for (int l = 0; l < L; ++l) {
                double gr_ = 0., gi_ = 0.;
#pragma unroll
                for (int ll = 0; ll < LMAX; ++ll) if (ll == l) { gr_ = accr[ll]; gi_ = acci[ll]; }
                a.tg[((size_t)k * L + l) * a.N_T + n] = make_double2(rho * gr_, rho * gi_);
            }
        }
        if (tid == 0) {
            const bool redone = a.batch_flag && a.batch_flag[batch];
            if (!converged && !redone) atomicOr(&a.flags[0], 4);
            if (!redone) stat_add(a.stats, 8, (unsigned long long)m_used * (unsigned long long)min(16, a.N_T - n0));
        }
        __syncthreads();   // red slots / scratch are reused by the next batch
    }
}

// ---------------------------------------------------------------------------------------
// Kernel 5d: derivative overlaps of the batches whose cells need SUB-STEPS (deriv_substeps > 1; flagged by
// deriv2_kernel / selected by `batch_flag`): the coupled recursion of deriv_mfma_kernel on the extended vectors,
// applied m times with the step dt / m without resetting the gradient slots, so that
//     (chi'_1..chi'_L, chi)  <-  exp(-i G dtb / m)^m (0, .., 0, chi)
// stays accurate for any ||H|| dt (the role the scaling and squaring of the dense block exponential plays in the
// reference's :gradgen route).  The results of a sub-step are needed as VECTORS (they start the next one), so the
// accumulators are vector tiles in registers and the overlaps <chi'_l|Psi> are taken once at the end.  Operators stream
// from L2 (fragment-packed), vectors ping-pong through the block-private global scratch: this is the slow, robust path.
// ---------------------------------------------------------------------------------------
// one thread per batch of 16 consecutive cells: does any of them need sub-steps?  (flags[3] counts the batches)
struct DerivFlagArgs {
    const double *rb, *eps, *shape, *dts;
    int rb_k, K, L, N_T, hc_per_traj, batches_per_k, nbatch_total;
    double sub_theta;
    int *batch_flag, *flags;
};
__global__ void deriv_flag_kernel(DerivFlagArgs a) {
    const int batch = blockIdx.x * blockDim.x + threadIdx.x;
    if (batch >= a.nbatch_total) return;
    const int k = batch / a.batches_per_k, n0 = (batch - k * a.batches_per_k) * 16;
    int ns = 1;
    for (int n = n0; n < min(n0 + 16, a.N_T); ++n)
        ns = max(ns, deriv_substeps(a.rb, a.rb_k, a.sub_theta, k, a.hc_per_traj, a.L, a.eps, a.shape, a.N_T, n, a.dts[n]));
    a.batch_flag[batch] = ns > 1;
    if (ns > 1) atomicAdd(&a.flags[3], 1);
}

// Round 6 -- one thread per batch: the degree of the economized derivative series (asm/gen_d3.py, tools/econ_coeffs.py)
// EVERY cell of the batch is certified for, 0: none.  Certificates of this evaluation's exponential kernels:
//   * four-product assembly kernels (asm/gen_t16.py): verdict 0 = spectral radius of the exponentiated matrix <= T16_THETA
//     = 1.36; a cell exponentiated as A / 2 (one planned squaring) is certified for 2.72; more squarings: none.  A route
//     that was not tried in this evaluation has written no verdicts.
//   * blocked path, Hermitian generators (lg_t18_decide_kernel): cell_deg, from the norms of A^2 and A^6.
// Of the degrees of a batch's cells the largest (its segment holds the others).
struct DerivEconArgs {
    const int *verdict, *splan, *cls, *flags;   // [KC * N_T] each; cls: generator class of trajectory k (nullptr: k)
    const int *cell_deg;                        // nullptr or [KC * N_T]: degrees named by the blocked path
    int K, KC, N_T, batches_per_k, nbatch_total;
    int deg0, deg1;                             // degrees of the segments 1.36 and 2.72 (grape_econ_coeffs.h)
    int *batch_econ;                            // [nbatch_total]
};
__global__ void deriv_econ_kernel(DerivEconArgs a) {
    const int batch = blockIdx.x * blockDim.x + threadIdx.x;
    if (batch >= a.nbatch_total) return;
    const int k = batch / a.batches_per_k, n0 = (batch - k * a.batches_per_k) * 16;
    const int kc = a.cls ? a.cls[k] : k;
    bool ok = a.cell_deg || !t16_skipped(a.flags, a.KC * a.N_T);
    int deg = 0;
    for (int n = n0; ok && n < min(n0 + 16, a.N_T); ++n) {
        const int cell = kc * a.N_T + n;
        int d;
        if (a.cell_deg) d = a.cell_deg[cell];
        else {
            const int sq = a.splan ? a.splan[cell] : 0;    // (the compiled four-product kernel plans no squarings)
            d = a.verdict[cell] != 0 ? 0 : sq == 0 ? a.deg0 : sq == 1 ? a.deg1 : 0;
        }
        ok = d > 0;
        deg = max(deg, d);
    }
    a.batch_econ[batch] = ok ? deg : 0;
}

struct DerivSubArgs {
    DerivMfmaArgs m;          // operators, states, scratch, tolerances (deriv_mfma_kernel's)
    const double *rb;         // 2-norm estimates, see deriv_substeps
    int rb_k;
    double sub_theta;
    const int *batch_flag;    // nullptr: every batch; else only the batches with a non-zero flag
};

template <int NP, int LMAX>
__global__ void __launch_bounds__((NP / 16 <= 8 ? NP / 16 : 8) * 64) deriv_sub_kernel(DerivSubArgs b) {
    const DerivMfmaArgs &a = b.m;
    constexpr int RT = NP / 16, KS = NP / 4;
    constexpr int NW = RT <= 8 ? RT : 8, TPW = RT / NW, NV = 1 + LMAX;
    static_assert(RT % NW == 0, "row tiles must divide evenly over the waves");
    __shared__ double red[2][NW][NV][16];       // per wave and column: ||new vector||^2 of the wave's rows; index LMAX = the chi chain
    __shared__ double redo[NW][LMAX][16][2];    // final overlaps
    if (b.batch_flag && a.flags[3] == 0) return;   // no batch was flagged in this evaluation
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, rg = lane >> 4;
    const int L = a.L;
    const size_t vplane = (size_t)NP * 16;
    double *vbase = a.vecs + (size_t)blockIdx.x * 2 * NV * 2 * vplane;
    for (int batch = blockIdx.x; batch < a.nbatch_total; batch += gridDim.x) {
        if (b.batch_flag && !b.batch_flag[batch]) continue;
        const int k = batch / a.batches_per_k;
        const int n0 = (batch - k * a.batches_per_k) * 16;
        const int n = n0 + c;
        const bool valid = n < a.N_T;
        const int nc = valid ? n : a.N_T - 1;
        const double *h0k = a.H0p + (size_t)k * RT * KS * 128;
        const double *hck = a.Hcp + (size_t)(a.hc_per_traj ? k : 0) * L * RT * KS * 128;
        const double dt = a.dts[nc];
        double e[LMAX], sh[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            sh[l] = (l < L && a.shape) ? a.shape[(size_t)l * a.N_T + nc] : 1.0;
            e[l] = l < L ? a.eps[(size_t)l * a.N_T + nc] * sh[l] : 0.;
        }
        // one sub-step count for the 16 cells of the batch (the columns advance in lockstep): the largest
        int nsub = valid ? deriv_substeps(b.rb, b.rb_k, b.sub_theta, k, a.hc_per_traj, L, a.eps, a.shape, a.N_T, nc, dt) : 1;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) nsub = max(nsub, __shfl_xor(nsub, off, 64));
        nsub = __builtin_amdgcn_readfirstlane(nsub);
        const double dts_ = dt / (double)nsub;
        const double rho = a.rho[k];
        // this wave's tiles of Psi(t_n), of the running chi and of the accumulated chi'_l
        double psr[TPW][4], psi_[TPW][4];
        d4 chr[TPW], chi_[TPW], gfr[LMAX][TPW], gfi[LMAX][TPW];
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (wave + tt * NW) + 4 * r + rg;
                const double2 p = a.fw[((size_t)k * (a.N_T + 1) + nc) * NP + row];
                const double2 x = a.bw[((size_t)k * (a.N_T + 1) + nc + 1) * NP + row];
                psr[tt][r] = p.x; psi_[tt][r] = p.y;
                chr[tt][r] = valid ? x.x : 0.; chi_[tt][r] = valid ? x.y : 0.;
            }
#pragma unroll
            for (int l = 0; l < LMAX; ++l) { gfr[l][tt] = (d4){0., 0., 0., 0.}; gfi[l][tt] = (d4){0., 0., 0., 0.}; }
        }
        const int all_done = ((1 << L) - 1) | (1 << LMAX);   // bit LMAX: the chi chain
        int converged = 1;
        unsigned long long orders = 0;
        for (int sub = 0; sub < nsub; ++sub) {
            __syncthreads();   // the scratch of the previous sub-step / batch is free
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t o = (size_t)(16 * (wave + tt * NW) + 4 * r + rg) * 16 + c;
                    vbase[0 * vplane + o] = chr[tt][r];
                    vbase[1 * vplane + o] = chi_[tt][r];
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) {
                        vbase[((1 + l) * 2 + 0) * vplane + o] = gfr[l][tt][r];
                        vbase[((1 + l) * 2 + 1) * vplane + o] = gfi[l][tt][r];
                    }
                }
            __syncthreads();
            d4 ocr[TPW], oci[TPW];   // sum_m alpha_m pw_m
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) { ocr[tt] = (d4){0., 0., 0., 0.}; oci[tt] = (d4){0., 0., 0., 0.}; }
            double alr = 0., ali = dts_;              // alpha_1 = i dt  (dtb = -dt)
            int done_mask = valid ? 0 : all_done;
            int cur = 0, conv_sub = 0, m_sub = a.max_order;
            for (int m = 1; m <= a.max_order; ++m) {
                const double *vc = vbase + (size_t)cur * NV * 2 * vplane;
                double *vn = vbase + (size_t)(cur ^ 1) * NV * 2 * vplane;
                const int slot = m & 1;
                double snn[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) snn[v] = 0.;
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt) {
                    const int rt = wave + tt * NW;
                    const double *h0p = h0k + (size_t)rt * KS * 128;
                    d4 pwr = {0., 0., 0., 0.}, pwi = {0., 0., 0., 0.};
                    d4 mur[LMAX], mui[LMAX], phr[LMAX], phi[LMAX];
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) {
                        mur[l] = (d4){0., 0., 0., 0.}; mui[l] = (d4){0., 0., 0., 0.};
                        phr[l] = (d4){0., 0., 0., 0.}; phi[l] = (d4){0., 0., 0., 0.};
                    }
#pragma unroll 2
                    for (int ks = 0; ks < KS; ++ks) {
                        double ar[NV], ai[NV];
                        ar[0] = h0p[(size_t)ks * 128 + lane];
                        ai[0] = h0p[(size_t)ks * 128 + 64 + lane];
#pragma unroll
                        for (int l = 0; l < LMAX; ++l) {
                            ar[1 + l] = l < L ? hck[(((size_t)l * RT + rt) * KS + ks) * 128 + lane] : 0.;
                            ai[1 + l] = l < L ? hck[(((size_t)l * RT + rt) * KS + ks) * 128 + 64 + lane] : 0.;
                        }
                        const size_t bo = (size_t)(4 * ks + rg) * 16 + c;
                        const double bwr = vc[0 * vplane + bo], bwi = vc[1 * vplane + bo];
                        pwr = MFMA64(ar[0], bwr, pwr);  pwi = MFMA64(ar[0], bwi, pwi);
                        pwr = MFMA64(ai[0], -bwi, pwr); pwi = MFMA64(ai[0], bwr, pwi);
#pragma unroll
                        for (int l = 0; l < LMAX; ++l) {
                            mur[l] = MFMA64(ar[1 + l], bwr, mur[l]);  mui[l] = MFMA64(ar[1 + l], bwi, mui[l]);
                            mur[l] = MFMA64(ai[1 + l], -bwi, mur[l]); mui[l] = MFMA64(ai[1 + l], bwr, mui[l]);
                        }
                        if (m > 1 || sub > 0) {   // the gradient slots are empty only at the very start
#pragma unroll
                            for (int l = 0; l < LMAX; ++l) {
                                const double br = vc[((1 + l) * 2 + 0) * vplane + bo], bi = vc[((1 + l) * 2 + 1) * vplane + bo];
                                phr[l] = MFMA64(ar[0], br, phr[l]);  phi[l] = MFMA64(ar[0], bi, phi[l]);
                                phr[l] = MFMA64(ai[0], -bi, phr[l]); phi[l] = MFMA64(ai[0], br, phi[l]);
#pragma unroll
                                for (int l2 = 0; l2 < LMAX; ++l2) {
                                    const double sr = e[l2] * br, si = e[l2] * bi;   // column-scaled B operand
                                    phr[l] = MFMA64(ar[1 + l2], sr, phr[l]);  phi[l] = MFMA64(ar[1 + l2], si, phi[l]);
                                    phr[l] = MFMA64(ai[1 + l2], -si, phr[l]); phi[l] = MFMA64(ai[1 + l2], sr, phi[l]);
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) {
                        pwr += e[l] * mur[l];
                        pwi += e[l] * mui[l];
                        phr[l] += sh[l] * mur[l];
                        phi[l] += sh[l] * mui[l];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const size_t o = (size_t)(16 * rt + 4 * r + rg) * 16 + c;
                        vn[0 * vplane + o] = pwr[r];
                        vn[1 * vplane + o] = pwi[r];
                        snn[LMAX] += pwr[r] * pwr[r] + pwi[r] * pwi[r];
                    }
                    ocr[tt] += alr * pwr - ali * pwi;   // += alpha_m pw_m
                    oci[tt] += alr * pwi + ali * pwr;
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const size_t o = (size_t)(16 * rt + 4 * r + rg) * 16 + c;
                            vn[((1 + l) * 2 + 0) * vplane + o] = phr[l][r];
                            vn[((1 + l) * 2 + 1) * vplane + o] = phi[l][r];
                            snn[l] += phr[l][r] * phr[l][r] + phi[l][r] * phi[l][r];
                        }
                        if (!((done_mask >> l) & 1)) {
                            gfr[l][tt] += alr * phr[l] - ali * phi[l];   // += alpha_m phi_m
                            gfi[l][tt] += alr * phi[l] + ali * phr[l];
                        }
                    }
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) {   // the 4 row groups of a column sit 16 lanes apart
                    snn[v] += __shfl_xor(snn[v], 16, 64);
                    snn[v] += __shfl_xor(snn[v], 32, 64);
                    if (lane < 16) red[slot][wave][v][c] = snn[v];
                }
                __syncthreads();
                const double al2 = alr * alr + ali * ali;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const bool used = v == LMAX || v < L;
                    const int bit = v == LMAX ? LMAX : v;
                    if (used && !((done_mask >> bit) & 1)) {
                        double nn = 0.;
#pragma unroll
                        for (int w = 0; w < NW; ++w) nn += red[slot][w][v][c];
                        if (m >= 2 && al2 * nn < a.tol * a.tol) done_mask |= 1 << bit;
                    }
                }
                cur ^= 1;
                if (__all(done_mask == all_done)) { conv_sub = 1; m_sub = m; break; }
                {
                    const double f = dts_ / (double)(m + 1);
                    const double nr = -ali * f, ni = alr * f;
                    alr = nr; ali = ni;
                }
            }
            converged &= conv_sub;
            orders += (unsigned long long)m_sub;
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) { chr[tt] += ocr[tt]; chi_[tt] += oci[tt]; }
        }
        // tau_grads = rho <chi'_l | Psi> = rho sum_rows conj(chi'_l) Psi
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            double orr = 0., oi = 0.;
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    orr += gfr[l][tt][r] * psr[tt][r] + gfi[l][tt][r] * psi_[tt][r];
                    oi += gfr[l][tt][r] * psi_[tt][r] - gfi[l][tt][r] * psr[tt][r];
                }
            orr += __shfl_xor(orr, 16, 64); oi += __shfl_xor(oi, 16, 64);
            orr += __shfl_xor(orr, 32, 64); oi += __shfl_xor(oi, 32, 64);
            if (lane < 16) { redo[wave][l][c][0] = orr; redo[wave][l][c][1] = oi; }
        }
        __syncthreads();
        if (tid < 16 && valid) {
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {
                if (l < L) {
                    double gr_ = 0., gi_ = 0.;
#pragma unroll
                    for (int w = 0; w < NW; ++w) { gr_ += redo[w][l][c][0]; gi_ += redo[w][l][c][1]; }
                    a.tg[((size_t)k * L + l) * a.N_T + n] = make_double2(rho * gr_, rho * gi_);
                }
            }
        }
        if (tid == 0) {
            if (!converged) atomicOr(&a.flags[0], 4);
            stat_add(a.stats, 8, orders * (unsigned long long)min(16, a.N_T - n0));
        }
    }
}

// ---------------------------------------------------------------------------------------
// Kernel 5c: the same overlaps <chi'_l | Psi> with 2 (1+L) instead of (1+L)^2 products per series order.
//
// chi'_l = L(B, E_l) chi with B = i dt H^dagger, E_l = i dt s_l mu_l^dagger (Frechet derivative of the backward
// step), so <chi'_l|Psi> = sum_{a,b} <B^a E_l B^b chi | Psi> / (a+b+1)!.  Moving B^a to the other side,
//     <chi'_l|Psi> = -i dt s_l  sum_a  < mu_l^dagger w_a | u_a > / (a+1),
//     u_a = A^a Psi / a!  (A = B^dagger = -i dt H),      w_a = chi + B w_{a+1} / (a+2)   (Horner, descending),
// i.e. ONE Krylov sequence on either side instead of 1+L coupled ones: pass 1 builds u_0, u_1, ... (products
// H0 u, mu_l u; stops when ||u_a|| has dropped below the tolerance for all 16 cells, which fixes the order M)
// and parks every u_a (16 KB per order at N = 64) in a block-private global area; pass 2 walks w_a down from
// a = M-1 (products H0^dagger w, mu_l^dagger w -- the mu_l^dagger w_a ARE the vectors the overlaps need),
// re-reading u_a as it goes.  The products are the batched fixed-matrix MFMA products of deriv_mfma_kernel
// (16 consecutive cells of a trajectory are the 16 MFMA columns, the cell-dependent pulse values scale
// accumulators per column).  Overlaps accumulate per lane over all orders; one reduction at the end.
// ---------------------------------------------------------------------------------------
struct Deriv2Args {
    const double *H0p, *Hcp;   // fragment-packed H0^dagger, mu_l^dagger   [K][RT][KS][2][64], [Kc][L][RT][KS][2][64]
    const double *H0q, *Hcq;   // fragment-packed H0, mu_l (not transposed)
    const double *eps, *shape, *dts;
    const double2 *fw, *bw;
    const double *rho;
    double2 *tg;
    double *park;              // [gridDim.x][maxm][2][NP][16]
    int *flags;
    unsigned long long *stats;
    int K, L, N_T, hc_per_traj, max_order, maxm, nbatch_total, batches_per_k;
    double tol;
    // matrix-free propagator: the forward sweep already summed u_a = A^a Psi / a! for every cell and parked the
    // terms (series_sweep_body); pass 1 is skipped for every batch whose 16 cells have a usable record
    const double2 *gpark;      // nullptr or [K][N_T][maxp][NP]
    const int *morder;         // [K][N_T] number of terms M of the cell (u_M below the tolerance), < 0: not usable
    int maxp;
    // batches with a cell that needs sub-steps (deriv_flag_kernel) are redone by deriv_sub_kernel afterwards: whatever
    // this kernel writes for them is overwritten, only its non-convergence flag has to stay down
    const int *batch_flag;     // nullptr or [nbatch_total]
    // deriv3_kernel at one and two tiles per side: a series that is not converged within the terms the kernel parks, while
    // max_order allows more, raises flags[7] instead of the non-convergence error -- deriv_kernel redoes the derivatives
    int deep_redo;
    // round 6: the degrees of the economized series lie behind the batch flags, batch_flag[nbatch_total + batch]
    // (deriv_econ_kernel; Hermitian operators only).  The assembly kernels find the tables behind their 1 / m; the
    // compiled ones (deriv3_kernel, deriv2_kernel) through econ_pairs: degree M at 64 (M - 16) doubles, (omega_a, sigma_a)
    int batch_econ;
    const double *econ_pairs;
#ifdef GRAPE_DIAG
    int ablate;                // diagnostic builds only: bit0 no parking traffic (results wrong)
#endif
};

// STREAM_L (more than four controls): the 1 + L products of an order are formed ONE MATRIX AT A TIME and folded into the
// running sum at once (H w = H0 w + sum_l e_l mu_l w; pass 2 also takes the overlap of mu_l^dagger w with u_a), so that the
// partial products of a single matrix are live instead of those of all 1 + L: with all of them live the kernel needs
// 24 (1 + L) accumulator registers and spills from L = 5 on (644 bytes of scratch per lane at L = 6).
// NP = 512 (matrix-free propagator for 256 < N <= 512): ONE vector block instead of the ping-pong pair -- two of them are
// 262 KB, the LDS has 160.  A wave then keeps the new rows of its row tiles in registers until every wave is done reading
// the old block (one more barrier per series order).
template <int NP, int LMAX, bool CACHE_A, bool STREAM_L = false>
__global__ void __launch_bounds__((NP / 16 <= 8 ? NP / 16 : 8) * 64) deriv2_kernel(Deriv2Args a) {
    extern __shared__ __attribute__((aligned(16))) double dsm2[];   // [2][2][NP*16] ping-pong vector block ([1][2][NP*16] at NP = 512)
    constexpr bool ONEBUF = NP > 256;
    constexpr int RT = NP / 16, KS = NP / 4;
    constexpr int NW = RT <= 8 ? RT : 8, TPW = RT / NW, NV = 1 + LMAX;
    static_assert(RT % NW == 0, "row tiles must divide evenly over the waves");
    static_assert(!CACHE_A || TPW == 1, "operand caching only for one tile per wave");
    static_assert(!(CACHE_A && STREAM_L), "the streamed form reads its operator fragments from L2");
    __shared__ double redn[2][NW][16];
    __shared__ double redo[NW][LMAX][16][2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, rg = lane >> 4;
    const int L = a.L;
    constexpr size_t vplane = (size_t)NP * 16;
    double *park = a.park + (size_t)blockIdx.x * a.maxm * 2 * vplane;

    for (int batch = blockIdx.x; batch < a.nbatch_total; batch += gridDim.x) {
        const int k = batch / a.batches_per_k;
        const int n0 = (batch - k * a.batches_per_k) * 16;
        const int n = n0 + c;
        const bool valid = n < a.N_T;
        const int nc = valid ? n : a.N_T - 1;
        const size_t kc = (size_t)(a.hc_per_traj ? k : 0) * L * RT * KS * 128;
        const double dt = a.dts[nc];
        double e[LMAX], sh[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            sh[l] = (l < L && a.shape) ? a.shape[(size_t)l * a.N_T + nc] : 1.0;
            e[l] = l < L ? a.eps[(size_t)l * a.N_T + nc] * sh[l] : 0.;
        }
        double car[CACHE_A ? NV : 1][CACHE_A ? KS : 1], cai[CACHE_A ? NV : 1][CACHE_A ? KS : 1];
        auto load_frags = [&](const double *h0k, const double *hck) __attribute__((always_inline)) {
            if constexpr (CACHE_A) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    car[0][ks] = h0k[((size_t)wave * KS + ks) * 128 + lane];
                    cai[0][ks] = h0k[((size_t)wave * KS + ks) * 128 + 64 + lane];
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) {
                        car[1 + l][ks] = l < L ? hck[(((size_t)l * RT + wave) * KS + ks) * 128 + lane] : 0.;
                        cai[1 + l][ks] = l < L ? hck[(((size_t)l * RT + wave) * KS + ks) * 128 + 64 + lane] : 0.;
                    }
                }
            }
        };
        // products of the (1+L) fixed matrices with the vector block `vc` for row tile rt: acc[v] (re, im)
        auto products = [&](const double *h0k, const double *hck, const double *vc, const int rt, d4 (&pr)[NV],
                            d4 (&pi)[NV]) __attribute__((always_inline)) {
            // complex products by the 3M scheme: P1 = Ar Br, P2 = Ai Bi, P3 = (Ar + Ai)(Br + Bi);
            // re = P1 - P2, im = P3 - P1 - P2 -- three real MFMA products instead of four (normwise stable; the
            // operand sums cost one VALU add each per k-step against 3 NV MFMAs of 64 cycles)
            d4 p3[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) { pr[v] = (d4){0., 0., 0., 0.}; pi[v] = (d4){0., 0., 0., 0.}; p3[v] = (d4){0., 0., 0., 0.}; }
            const double *vb = vc + (size_t)rg * 16 + c;
            double bwr = vb[0], bwi = vb[vplane];
            auto ks_step = [&](const int ks) __attribute__((always_inline)) {
                double ar[NV], ai[NV];
                if constexpr (CACHE_A) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) { ar[v] = car[v][CACHE_A ? ks : 0]; ai[v] = cai[v][CACHE_A ? ks : 0]; }
                } else {
                    ar[0] = h0k[((size_t)rt * KS + ks) * 128 + lane];
                    ai[0] = h0k[((size_t)rt * KS + ks) * 128 + 64 + lane];
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) {
                        ar[1 + l] = l < L ? hck[(((size_t)l * RT + rt) * KS + ks) * 128 + lane] : 0.;
                        ai[1 + l] = l < L ? hck[(((size_t)l * RT + rt) * KS + ks) * 128 + 64 + lane] : 0.;
                    }
                }
                const double br = bwr, bi = bwi, bs = bwr + bwi;
                if (ks + 1 < KS) {   // B operands of the next k-step: rows 4(ks+1) + rg of the block, column c
                    bwr = vb[(size_t)(4 * (ks + 1)) * 16];
                    bwi = vb[vplane + (size_t)(4 * (ks + 1)) * 16];
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    pr[v] = MFMA64(ar[v], br, pr[v]);  pi[v] = MFMA64(ai[v], bi, pi[v]);   // P1, P2
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) p3[v] = MFMA64(ar[v] + ai[v], bs, p3[v]);
            };
            if constexpr (CACHE_A) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) ks_step(ks);
            } else {
#pragma unroll 4
                for (int ks = 0; ks < KS; ++ks) ks_step(ks);
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const d4 p1 = pr[v], p2 = pi[v];
                pr[v] = p1 - p2;
                pi[v] = p3[v] - p1 - p2;
            }
        };

        // product of ONE fixed matrix (fragments at `frag`: [RT][KS][2][64]) with the vector block `vc` for row tile rt
        auto product1 = [&](const double *frag, const double *vc, const int rt, d4 &qr, d4 &qi) __attribute__((always_inline)) {
            d4 p1 = (d4){0., 0., 0., 0.}, p2 = p1, p3 = p1;
            const double *vb = vc + (size_t)rg * 16 + c;
            double bwr = vb[0], bwi = vb[vplane];
#pragma unroll 4
            for (int ks = 0; ks < KS; ++ks) {
                const double ar = frag[((size_t)rt * KS + ks) * 128 + lane], ai = frag[((size_t)rt * KS + ks) * 128 + 64 + lane];
                const double br = bwr, bi = bwi, bs = bwr + bwi;
                if (ks + 1 < KS) {
                    bwr = vb[(size_t)(4 * (ks + 1)) * 16];
                    bwi = vb[vplane + (size_t)(4 * (ks + 1)) * 16];
                }
                p1 = MFMA64(ar, br, p1); p2 = MFMA64(ai, bi, p2); p3 = MFMA64(ar + ai, bs, p3);
            }
            qr = p1 - p2;
            qi = p3 - p1 - p2;
        };

        // ---- terms already parked by the forward sweep? (same 16 cells in every wave: uniform decision) ----
        int Mc = 0;
        bool use_g = false;
        const int mcap = a.max_order < a.maxm ? a.max_order : a.maxm;
        if (a.gpark) {
            Mc = valid ? a.morder[(size_t)k * a.N_T + n] : 0;
            use_g = !__any(valid && (Mc < 2 || Mc > mcap));
        }
        int cur = 0, M = 0, converged = 0;
        // round 6: degree of the economized polynomial this batch is certified for (asm/gen_d3.py's header), 0: none
        const int deg = (a.batch_econ && a.batch_flag && a.econ_pairs && !use_g) ? a.batch_flag[a.nbatch_total + batch] : 0;
        const int capb = (deg && deg - 1 <= mcap) ? deg - 1 : mcap;
        const double *pairs = nullptr;
        if (use_g) {
            int mm = Mc;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) mm = max(mm, __shfl_xor(mm, off, 64));
            M = __builtin_amdgcn_readfirstlane(mm);
            converged = 1;
        } else {
        // ---- pass 1: u_0 = Psi(t_n), u_{a+1} = (-i dt / (a+1)) H u_a ----
        const double *h0q = a.H0q + (size_t)k * RT * KS * 128, *hcq = a.Hcq + kc;
        load_frags(h0q, hcq);
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (wave + tt * NW) + 4 * r + rg;
                const double2 p = a.fw[((size_t)k * (a.N_T + 1) + nc) * NP + row];
                const double pr_ = valid ? p.x : 0., pi_ = valid ? p.y : 0.;
                const size_t o = (size_t)row * 16 + c;
                dsm2[o] = pr_; dsm2[vplane + o] = pi_;
                park[o] = pr_; park[vplane + o] = pi_;
            }
        __syncthreads();
        for (int m = 1; m <= capb; ++m) {   // forms u_m
            const double *vc = dsm2 + (size_t)cur * 2 * vplane;
            double *vn = dsm2 + (size_t)(ONEBUF ? cur : cur ^ 1) * 2 * vplane;
            const double sfac = dt / (double)m;
            double nn = 0.;
            double keep_r[ONEBUF ? TPW : 1][4], keep_i[ONEBUF ? TPW : 1][4];
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) {
                const int rt = wave + tt * NW;
                d4 pr[STREAM_L ? 1 : NV], pi[STREAM_L ? 1 : NV];
                if constexpr (STREAM_L) {
                    product1(h0q, vc, rt, pr[0], pi[0]);
                    for (int l = 0; l < L; ++l) {
                        d4 qr, qi;
                        product1(hcq + (size_t)l * RT * KS * 128, vc, rt, qr, qi);
                        pr[0] += e[l] * qr; pi[0] += e[l] * qi;
                    }
                } else {
                    products(h0q, hcq, vc, rt, pr, pi);
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) { pr[0] += e[l] * pr[1 + l]; pi[0] += e[l] * pi[1 + l]; }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t o = (size_t)(16 * rt + 4 * r + rg) * 16 + c;
                    const double ur = sfac * pi[0][r], ui = -sfac * pr[0][r];   // (-i s)(x + i y) = s y - i s x
                    if constexpr (ONEBUF) { keep_r[ONEBUF ? tt : 0][r] = ur; keep_i[ONEBUF ? tt : 0][r] = ui; }
                    else { vn[o] = ur; vn[vplane + o] = ui; }
#ifdef GRAPE_DIAG
                    if (!(a.ablate & 1))
#endif
                    if (m < a.maxm) { park[(size_t)m * 2 * vplane + o] = ur; park[(size_t)m * 2 * vplane + vplane + o] = ui; }
                    nn += ur * ur + ui * ui;
                }
            }
            if constexpr (ONEBUF) {
                __syncthreads();   // every wave is done reading the block
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const size_t o = (size_t)(16 * (wave + tt * NW) + 4 * r + rg) * 16 + c;
                        vn[o] = keep_r[ONEBUF ? tt : 0][r]; vn[vplane + o] = keep_i[ONEBUF ? tt : 0][r];
                    }
            }
            nn += __shfl_xor(nn, 16, 64);
            nn += __shfl_xor(nn, 32, 64);
            if (lane < 16) redn[m & 1][wave][c] = nn;
            __syncthreads();
            double tot = 0.;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += redn[m & 1][w][c];
            if constexpr (!ONEBUF) cur ^= 1;
            M = m;
            // ||u_m|| < tol for every cell of the batch (identical decision in every wave: same LDS values)
            if (m >= 2 && __all(tot < a.tol * a.tol)) { converged = 1; break; }
        }
        if (!converged && capb != mcap) {   // certified: the deg - 1 orders formed are all the polynomial needs
            M = deg; converged = 1;
            pairs = a.econ_pairs + (size_t)(deg - 16) * 64;
        }
        }   // !use_g
        // orders a = 0..M-1 enter the sum; u_M is below the tolerance (or the cap was hit: flagged below)

        // ---- pass 2: w_{M-1} = chi(t_{n+1}), w_{a-1} = chi + (i dt / (a+1)) H^dagger w_a ----
        const double *h0p = a.H0p + (size_t)k * RT * KS * 128, *hcp = a.Hcp + kc;
        load_frags(h0p, hcp);
        double chr[TPW][4], chi_[TPW][4];
        __syncthreads();   // pass 1 is done with the vector block
        cur = 0;
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (wave + tt * NW) + 4 * r + rg;
                const double2 x = a.bw[((size_t)k * (a.N_T + 1) + nc + 1) * NP + row];
                chr[tt][r] = valid ? x.x : 0.; chi_[tt][r] = valid ? x.y : 0.;
                dsm2[(size_t)row * 16 + c] = chr[tt][r];
                dsm2[vplane + (size_t)row * 16 + c] = chi_[tt][r];
            }
        __syncthreads();
        double dr[LMAX], di[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { dr[l] = 0.; di[l] = 0.; }
        for (int aa = M - 1; aa >= 0; --aa) {
            const double *vc = dsm2 + (size_t)cur * 2 * vplane;
            double *vn = dsm2 + (size_t)(ONEBUF ? cur : cur ^ 1) * 2 * vplane;
            const double inv = pairs ? pairs[2 * aa] : 1.0 / (double)(aa + 1), sfac = dt * (pairs ? pairs[2 * aa + 1] : inv);
            double keep_r[ONEBUF ? TPW : 1][4], keep_i[ONEBUF ? TPW : 1][4];
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) {
                const int rt = wave + tt * NW;
                double ur[4], ui[4];   // u_aa tile of this wave (requested before the products)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (use_g) {   // the cell's own record; terms past its M are below the tolerance: zero
                        const double2 u = a.gpark[(((size_t)k * a.N_T + nc) * a.maxp + aa) * NP + 16 * rt + 4 * r + rg];
                        const bool in = valid && aa < Mc;
                        ur[r] = in ? u.x : 0.; ui[r] = in ? u.y : 0.;
                    } else {
                        const size_t o = (size_t)aa * 2 * vplane + (size_t)(16 * rt + 4 * r + rg) * 16 + c;
#ifdef GRAPE_DIAG
                        if (a.ablate & 1) { ur[r] = chr[tt][r]; ui[r] = chi_[tt][r]; } else
#endif
                        { ur[r] = park[o]; ui[r] = park[vplane + o]; }
                    }
                }
                d4 pr[STREAM_L ? 1 : NV], pi[STREAM_L ? 1 : NV];
                if constexpr (STREAM_L) {
                    product1(h0p, vc, rt, pr[0], pi[0]);
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) {
                        if (l < L) {
                            d4 qr, qi;
                            product1(hcp + (size_t)l * RT * KS * 128, vc, rt, qr, qi);
                            double sr = 0., si = 0.;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {   // conj(mu_l^dagger w) * u
                                sr += qr[r] * ur[r] + qi[r] * ui[r];
                                si += qr[r] * ui[r] - qi[r] * ur[r];
                            }
                            dr[l] += inv * sr; di[l] += inv * si;
                            pr[0] += e[l] * qr; pi[0] += e[l] * qi;
                        }
                    }
                } else {
                products(h0p, hcp, vc, rt, pr, pi);
#pragma unroll
                for (int l = 0; l < LMAX; ++l) {
                    double sr = 0., si = 0.;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {   // conj(mu_l^dagger w) * u
                        sr += pr[1 + l][r] * ur[r] + pi[1 + l][r] * ui[r];
                        si += pr[1 + l][r] * ui[r] - pi[1 + l][r] * ur[r];
                    }
                    dr[l] += inv * sr; di[l] += inv * si;
                    pr[0] += e[l] * pr[1 + l]; pi[0] += e[l] * pi[1 + l];
                }
                }
                if (aa > 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {   // chi + (i s)(x + i y) = chi - s y + i s x
                        const size_t o = (size_t)(16 * rt + 4 * r + rg) * 16 + c;
                        const double wr_ = chr[tt][r] - sfac * pi[0][r], wi_ = chi_[tt][r] + sfac * pr[0][r];
                        if constexpr (ONEBUF) { keep_r[ONEBUF ? tt : 0][r] = wr_; keep_i[ONEBUF ? tt : 0][r] = wi_; }
                        else { vn[o] = wr_; vn[vplane + o] = wi_; }
                    }
                }
            }
            if constexpr (ONEBUF) {
                __syncthreads();   // every wave is done reading the block
                if (aa > 0) {
#pragma unroll
                    for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const size_t o = (size_t)(16 * (wave + tt * NW) + 4 * r + rg) * 16 + c;
                            vn[o] = keep_r[ONEBUF ? tt : 0][r]; vn[vplane + o] = keep_i[ONEBUF ? tt : 0][r];
                        }
                }
            }
            __syncthreads();
            if constexpr (!ONEBUF) cur ^= 1;
        }
        // ---- tau_grads = rho (-i dt s_l) sum_a <mu_l^dagger w_a | u_a> / (a+1) ----
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            dr[l] += __shfl_xor(dr[l], 16, 64); di[l] += __shfl_xor(di[l], 16, 64);
            dr[l] += __shfl_xor(dr[l], 32, 64); di[l] += __shfl_xor(di[l], 32, 64);
            if (lane < 16) { redo[wave][l][c][0] = dr[l]; redo[wave][l][c][1] = di[l]; }
        }
        __syncthreads();
        if (tid < 16 && valid) {
            const double rho = a.rho[k];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {
                if (l < L) {
                    double Dr = 0., Di = 0.;
#pragma unroll
                    for (int w = 0; w < NW; ++w) { Dr += redo[w][l][c][0]; Di += redo[w][l][c][1]; }
                    const double f = rho * dt * sh[l];   // (-i f)(Dr + i Di) = f Di - i f Dr
                    a.tg[((size_t)k * L + l) * a.N_T + n] = make_double2(f * Di, -f * Dr);
                }
            }
        }
        if (tid == 0) {
            const bool redone = a.batch_flag && a.batch_flag[batch];
            if (!converged && !redone) atomicOr(&a.flags[0], 4);
            if (!redone) stat_add(a.stats, 8, (unsigned long long)M * (unsigned long long)min(16, a.N_T - n0));
        }
        __syncthreads();   // LDS block, reduction slots and parking area are reused by the next batch
    }
}

// ---------------------------------------------------------------------------------------
// State-dependent running cost of the family g_b(Psi) = <Psi|D|Psi> (test_state_running_cost.jl:32-40):
// xi_k(t_n) = -D_k Psi_k(t_n) and g_kn for every stored state (cell-parallel), then the trapezoid sum
// J_b = sum_k sum_n wq[n] g_kn (optimize.jl:727-750).
// ---------------------------------------------------------------------------------------
struct GbArgs {
    const double2 *Dt;   // [Kd][NP][NP] interleaved, TRANSPOSED (Dt[j][i] = D[i][j]: lane i reads contiguously)
    const double2 *fw;   // [K][N_T+1][NP]
    double2 *xi;         // [K][N_T+1][NP]
    double *gb;          // [K][N_T+1]
    int NP, N_T, d_per_traj;
};
__global__ void __launch_bounds__(256) gb_kernel(GbArgs a) {
    __shared__ double2 psi[256];
    __shared__ double part[256];
    const int NP = a.NP, tid = threadIdx.x;
    const int cell = blockIdx.x;                 // k * (N_T+1) + n
    const int k = cell / (a.N_T + 1);
    const double2 *Dt = a.Dt + (size_t)(a.d_per_traj ? k : 0) * NP * NP;
    if (tid < NP) psi[tid] = a.fw[(size_t)cell * NP + tid];
    __syncthreads();
    double ar = 0., ai = 0.;
    if (tid < NP) {
        for (int j = 0; j < NP; ++j) {
            const double2 d = Dt[(size_t)j * NP + tid], p = psi[j];
            ar += d.x * p.x - d.y * p.y;
            ai += d.x * p.y + d.y * p.x;
        }
        a.xi[(size_t)cell * NP + tid] = make_double2(-ar, -ai);
    }
    part[tid] = tid < NP ? psi[tid].x * ar + psi[tid].y * ai : 0.;   // Re conj(psi_i) (D psi)_i
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) part[tid] += part[tid + off];
        __syncthreads();
    }
    if (tid == 0) a.gb[cell] = part[0];
}
__global__ void __launch_bounds__(256) jb_reduce_kernel(const double *gb, const double *wq, int K, int N_T, double *out) {
    __shared__ double part[256];
    double s = 0.;
    const size_t tot = (size_t)K * (N_T + 1);
    for (size_t i = threadIdx.x; i < tot; i += 256) s += wq[i % (N_T + 1)] * gb[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0];
}

// ---------------------------------------------------------------------------------------
// Kernel 6: G[l*N_T + n] = -2 Re sum_k tau_grads[k][l][n]   (_grad_J_T_via_chi!, optimize.jl:574-584)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) grad_reduce_kernel(double2 *tg, int K, int LN, double *G, const double2 *z) {
    // 16 gradient entries per workgroup, 16 threads per entry over the trajectories (a thread per entry walking all K rows
    // is a chain of K dependent load latencies on a handful of CUs: 46 us at C3); partial sums meet in LDS in a fixed order
    __shared__ double part[16][17];
    const int j = threadIdx.x & 15, kg = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + j;
    double s = 0.;
    if (idx < LN) {
        if (z) {
            // concurrent sweeps: the derivative kernels saw the unit backward states; tau_grads = z_k <chi~'_l|Psi>
            for (int k = kg; k < K; k += 16) {
                const double2 t = tg[(size_t)k * LN + idx], zk = z[k];
                const double2 v = make_double2(zk.x * t.x - zk.y * t.y, zk.x * t.y + zk.y * t.x);
                tg[(size_t)k * LN + idx] = v;
                s += v.x;
            }
        } else {
            for (int k = kg; k < K; k += 16) s += tg[(size_t)k * LN + idx].x;
        }
    }
    part[kg][j] = s;
    __syncthreads();
    if (kg == 0 && idx < LN) {
        double t = 0.;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[q][j];
        G[idx] = -2.0 * t;
    }
}
