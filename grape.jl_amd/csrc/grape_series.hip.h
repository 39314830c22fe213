// grape_series.hip.h -- matrix-free polynomial propagator for the sweeps (prop_method = GRAPE_PROP_SERIES).
//
// The reference selects the short-time propagator with `prop_method` (src/workspace.jl:222-232 ->
// QuantumPropagators.init_prop): ExpProp materialises U_n = exp(-i H_n dt_n) (the Pade path of
// grape_kernels.hip.h); Cheby / Newton apply a polynomial of H_n to the state and never form U_n
// (README.md:55 and docs/src/tutorial.md:308 recommend them for larger systems).  This kernel is the
// matrix-free member of that family on MI355X: the state advances by the power series of the exponential
// on the *vector*,
//     forward : Psi_n     = sum_a u_a,  u_0 = Psi_{n-1},  u_{a+1} = (-i dt / (a+1)) H_n u_a        (optimize.jl:731-738)
//     backward: chi_{n-1} = sum_a w_a,  w_0 = chi_n,      w_{a+1} = (+i dt / (a+1)) H_n^dagger w_a  (optimize.jl:881)
// summed until ||u_a|| < tol ||Psi|| (converged to rounding, tol = 1e-17 by default), so the stored states
// equal the ExpProp ones to fp64 rounding and everything downstream (tau, chi boundary, derivative kernels,
// reductions) is shared with the ExpProp path.  O(N^2) per term instead of O(N^3) per cell.
//
// One workgroup of 8 NP threads per trajectory walks the serial recurrence (two waves per SIMD at N = 64: a term
// is a chain of LDS round trip, dependent fp64 FMAs, cross-lane reduction and barrier, and the second wave
// fills the latencies of the first).  Thread (row pair g, column chunk q) keeps a 2 x CW tile (rows 2g, 2g+1,
// CW = NP/16 columns) of H0_k (and of the control operators when L <= 2) in registers and rebuilds the tile of
// H_n = H0_k + sum_l eps_nl S_nl H_l once per time step.  The vector lives in LDS and a thread reads its CW
// elements per term (the LDS return path bounds layouts that read more); the 16 chunks of a row pair are the 16
// lanes of a DPP row and meet in a reduce-scatter on DPP moves: row_mirror hands each half one of the two
// rows -- the upper half holds its two rows swapped, so no lane has to select -- then row_half_mirror and two
// quad permutes finish the sum.  The stopping rule is a per-wave ballot (|u_a[i]|^2 <= tol^2 ||state||^2 / NP
// for every i), one LDS word per wave and term, read together with the next vector.
// Steps whose bound rho_n = (r0_k + sum_l |eps_nl| r_l) dt_n on the spectral radius exceeds `theta` are split into
// m = ceil(rho_n / theta) sub-steps (the vector analogue of scaling and squaring: bounds the largest term of
// the series and with it the cancellation error); r0_k, r_l are 2-norm estimates made at grape_create.
#pragma once

struct SeriesArgs {
    SweepArgs s;          // boundary data, storage, tau / rho / flags (shared with sweep_kernel)
    const double *H0;     // [K][2][NP*NP] planar row-major: H0f (forward) or H0t (backward: rows of H^T)
    const double *Hc;     // [Kc][L][2][NP*NP]: Hcf or Hct
    const double *eps, *shape, *dts;
    const double *rb;     // [K + Kc*L] 2-norm estimates: r0_k, then r_(kc,l)
    unsigned long long *stats;   // [10] += series terms summed over (sub-)steps, [11] += sub-steps
    double tol, theta;
    int L, hc_per_traj, max_order;
    // forward sweep only: the terms u_a of every cell, parked for the derivative kernel (deriv2_kernel skips its
    // first pass): park[k][n][a][NP], a < maxp, and the number of terms of the cell (-1: sub-steps or too many)
    double2 *park;
    int *morder;
    int maxp;
};

// 1 / (a + 1) of the series coefficients (scalar loads; an IEEE division costs a dozen dependent fp64 instructions)
__device__ __constant__ double c_series_inv[256] = {
#define SI4(n) 1.0 / ((n) + 1.0), 1.0 / ((n) + 2.0), 1.0 / ((n) + 3.0), 1.0 / ((n) + 4.0)
#define SI16(n) SI4(n), SI4((n) + 4), SI4((n) + 8), SI4((n) + 12)
#define SI64(n) SI16(n), SI16((n) + 16), SI16((n) + 32), SI16((n) + 48)
    SI64(0), SI64(64), SI64(128), SI64(192)
#undef SI64
#undef SI16
#undef SI4
};

#define DPP_ROW_ROR4 0x124
#define DPP_ROW_ROR8 0x128
typedef double v2d_t __attribute__((ext_vector_type(2)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v2d_t lds_v2d_t;   // volatile LDS accesses keep their address space
typedef __attribute__((address_space(3))) v4i_t lds_v4i_t;
typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
#ifndef SERIES_RPT
#define SERIES_RPT 4   // matrix rows per thread tile (2: 8 NP threads, 4: 4 NP threads per workgroup)
#endif

// fp64 value of another lane: two 32-bit DPP moves (every lane is a valid source for the permutations used here)
template <int CTRL>
__device__ __forceinline__ double dppm_f64(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, true),
                            __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, true));
}

#ifdef GRAPE_SERIES_DIAG
#define SDIAG_DECL unsigned long long sd_t[6] = {0, 0, 0, 0, 0, 0}, sd_last = 0, sd_wall0 = wall_clock64(), sd_c0 = clock64();
#define SDIAG_START() do { sd_last = clock64(); } while (0)
#define SDIAG(i) do { const unsigned long long t_ = clock64(); sd_t[i] += t_ - sd_last; sd_last = t_; } while (0)
#else
#define SDIAG_DECL
#define SDIAG_START() do {} while (0)
#define SDIAG(i) do {} while (0)
#endif

template <int NP, bool BACKWARD, int LR>   // LR = controls whose tiles stay in registers (0: streamed from L2)
__device__ __forceinline__ void series_sweep_body(const SeriesArgs &a, const int k) {
    constexpr int NCH = 16, RPT = SERIES_RPT, CW = NP / NCH, NTH = NP * NCH / RPT, NW = (NTH + 63) / 64;
    static_assert(NW <= 8, "flag bytes of a term are read as one 64-bit LDS access");
    __shared__ double2 vec[2][NP];
    __shared__ double red[2][NW];
    __shared__ __attribute__((aligned(8))) unsigned char flg[2][8];
    __shared__ double sc[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = tid & (NCH - 1), g = tid >> 4;
    // element j of a vector sits at slot (j % CW) * 16 + j / CW: the 16 chunks of one read are contiguous
    auto slot = [](int j) { return (j % CW) * NCH + j / CW; };
    // tile row r of a thread is matrix row RPT g + (r ^ rsw): the rows are permuted per lane such that every step of
    // the reduce-scatter keeps the low tile rows and gives up the high ones -- no lane has to select
    const int rsw = RPT == 2 ? (q >= 8 ? 1 : 0) : (2 * (q & 1)) ^ ((q >> 1) & 1);
    const int myrow = RPT * g + rsw;
    const bool owner = RPT == 2 ? (q & 7) == 0 : q < 4;
    const SweepArgs &sa = a.s;
    const int N_T = sa.N_T, L = a.L;
    double2 *st = sa.store + (size_t)k * (N_T + 1) * NP;
    if (tid < 16) ((unsigned char *)flg)[tid] = 0;

    // ---- static tiles: columns q*CW .. q*CW+CW-1; backward reads rows of H^T and conjugates ----
    constexpr double SG = BACKWARD ? -1.0 : 1.0;
    double h0r[RPT][CW], h0i[RPT][CW], mur[LR > 0 ? LR : 1][RPT][CW], mui[LR > 0 ? LR : 1][RPT][CW];
    const double *h0 = a.H0 + (size_t)k * 2 * NP * NP;
    const double *hc = a.Hc + (size_t)(a.hc_per_traj ? k : 0) * L * 2 * NP * NP;
    int ridx[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) ridx[r] = (RPT * g + (r ^ rsw)) * NP + q * CW;
#pragma unroll
    for (int r = 0; r < RPT; ++r)
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            h0r[r][c] = h0[ridx[r] + c];
            h0i[r][c] = SG * h0[NP * NP + ridx[r] + c];
#pragma unroll
            for (int l = 0; l < LR; ++l) {
                mur[l][r][c] = l < L ? hc[(size_t)l * 2 * NP * NP + ridx[r] + c] : 0.;
                mui[l][r][c] = l < L ? SG * hc[(size_t)l * 2 * NP * NP + NP * NP + ridx[r] + c] : 0.;
            }
        }
    const double r0 = a.rb[k];
    const double *rl = a.rb + sa.K + (size_t)(a.hc_per_traj ? k : 0) * L;

    // ---- boundary state (same as sweep_kernel) ----
    if (!BACKWARD) {
        if (tid < NP) {
            const double2 v = tid < sa.N ? sa.psi0[(size_t)k * sa.N + tid] : make_double2(0., 0.);
            vec[0][slot(tid)] = v;
            st[tid] = v;
        }
        if (tid == 0) sc[0] = 1.0;
    } else {
        // chi_k(T) = coeff_k * target_k, rho_k = ||chi_k||, chi_k /= rho_k (optimize.jl:848-868)
        double2 v = make_double2(0., 0.);
        if (tid < sa.N) {
            v = chi_boundary(sa, k, tid);
            if (sa.xi) {   // chi_k(T) += lambda_b dt/2 xi_k(T)   (optimize.jl:856-866)
                const double2 x_ = sa.xi[((size_t)k * (N_T + 1) + N_T) * NP + tid];
                const double c = sa.lambda_b * sa.wq[N_T];
                v.x += c * x_.x; v.y += c * x_.y;
            }
        }
        {   // NP <= 64: wave 0 holds every element
            const double n2 = wave_sum_dpp(tid < NP ? v.x * v.x + v.y * v.y : 0.);
            if (tid == 0) sc[0] = sqrt(n2);
        }
        __syncthreads();
        const double rho = sc[0];
        if (tid == 0 && !sa.unit_chi) {
            sa.rho[k] = rho;
            if (rho < sa.chi_min_norm) atomicOr(&sa.flags[0], 2);
        }
        if (tid < NP) {
            const double ir = rho > 0. ? 1.0 / rho : 0.;
            v.x *= ir; v.y *= ir;
            vec[0][slot(tid)] = v;
            st[(size_t)N_T * NP + tid] = v;
        }
    }
    {   // ||state||^2 for the relative stopping rule
        double n2 = 0.;
        __syncthreads();
        if (tid < NP) { const double2 v = vec[0][slot(tid)]; n2 = v.x * v.x + v.y * v.y; }
        n2 = wave_sum_dpp(n2);
        if (tid == 0) sc[1] = n2;
        __syncthreads();
    }
    double nrm2 = sc[1];
    const double rho_k = sc[0];
    int cur = 0;
    unsigned long long terms = 0, substeps = 0;
    bool failed = false;
    SDIAG_DECL

    for (int step = 0; step < N_T; ++step) {
        SDIAG_START();
        const int n = BACKWARD ? N_T - 1 - step : step;
        const double dt_full = a.dts[n];
        // ---- tile of H_n (or of H_n^dagger) and the bound on its spectral radius ----
        double hr[RPT][CW], hi[RPT][CW];
#pragma unroll
        for (int r = 0; r < RPT; ++r)
#pragma unroll
            for (int c = 0; c < CW; ++c) { hr[r][c] = h0r[r][c]; hi[r][c] = h0i[r][c]; }
        double bound = r0;
        if constexpr (LR > 0) {
#pragma unroll
            for (int l = 0; l < LR; ++l) {
                if (l < L) {
                    const double e = a.eps[(size_t)l * N_T + n] * (a.shape ? a.shape[(size_t)l * N_T + n] : 1.0);
                    bound += fabs(e) * rl[l];
#pragma unroll
                    for (int r = 0; r < RPT; ++r)
#pragma unroll
                        for (int c = 0; c < CW; ++c) {
                            hr[r][c] = fma(e, mur[l][r][c], hr[r][c]);
                            hi[r][c] = fma(e, mui[l][r][c], hi[r][c]);
                        }
                }
            }
        } else {
            for (int l = 0; l < L; ++l) {
                const double e = a.eps[(size_t)l * N_T + n] * (a.shape ? a.shape[(size_t)l * N_T + n] : 1.0);
                bound += fabs(e) * rl[l];
                const double *m = hc + (size_t)l * 2 * NP * NP;
#pragma unroll
                for (int r = 0; r < RPT; ++r)
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        hr[r][c] = fma(e, m[ridx[r] + c], hr[r][c]);
                        hi[r][c] = fma(SG * e, m[NP * NP + ridx[r] + c], hi[r][c]);
                    }
            }
        }
        int msub = (int)ceil(bound * dt_full / a.theta);
        msub = msub < 1 ? 1 : (msub > 4096 ? 4096 : msub);
        const double dt = dt_full / (double)msub;

        double sr = 0., si = 0.;
        for (int sub = 0; sub < msub; ++sub) {
            // row owners accumulate the sum of the series, starting from u_0 = current state
            if (owner) { const double2 v = vec[cur][slot(myrow)]; sr = v.x; si = v.y; }
            double2 *pk = nullptr;
            if constexpr (!BACKWARD) {
                if (a.park && msub == 1) {
                    pk = a.park + ((size_t)k * N_T + n) * a.maxp * NP + myrow;
                    if (owner) pk[0] = make_double2(sr, si);
                }
            }
            const double thr_el = a.tol * a.tol * nrm2 / (double)NP;
            bool conv = false;
            int aord = 0;
            for (; aord < a.max_order; ++aord) {
                // the vector and the verdict on the term it holds travel together (volatile: the loads stay ahead of
                // the branch, one LDS round trip per term instead of two)
                const unsigned long long big = *(const volatile lds_u64_t *)flg[cur];
                double2 x[CW];
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    const v2d_t t = *(const volatile lds_v2d_t *)&vec[cur][c * NCH + q];
                    x[c] = make_double2(t.x, t.y);
                }
                SDIAG(0);
                if (aord > 0 && !big) { conv = true; break; }
                double pr[RPT], pi[RPT];
#pragma unroll
                for (int r = 0; r < RPT; ++r) { pr[r] = 0.; pi[r] = 0.; }
#pragma unroll
                for (int c = 0; c < CW; ++c)
#pragma unroll
                    for (int r = 0; r < RPT; ++r) {
                        pr[r] = fma(hr[r][c], x[c].x, pr[r]);
                        pi[r] = fma(hr[r][c], x[c].y, pi[r]);
                        pr[r] = fma(-hi[r][c], x[c].y, pr[r]);
                        pi[r] = fma(hi[r][c], x[c].x, pi[r]);
                    }
#ifdef GRAPE_SERIES_DIAG
                asm volatile("" : "+v"(pr[0]), "+v"(pi[0]), "+v"(pr[1]), "+v"(pi[1]));
#endif
                SDIAG(1);
                // reduce-scatter over the 16 lanes of the row group (see rsw), then the lanes that hold the same row sum
                double yr, yi;
                if constexpr (RPT == 2) {
                    yr = pr[0] + dppm_f64<DPP_ROW_MIRROR>(pr[1]);
                    yi = pi[0] + dppm_f64<DPP_ROW_MIRROR>(pi[1]);
                    yr += dppm_f64<DPP_ROW_HALF_MIRROR>(yr); yi += dppm_f64<DPP_ROW_HALF_MIRROR>(yi);
                    yr += dppm_f64<DPP_QUAD_XOR1>(yr); yi += dppm_f64<DPP_QUAD_XOR1>(yi);
                    yr += dppm_f64<DPP_QUAD_XOR2>(yr); yi += dppm_f64<DPP_QUAD_XOR2>(yi);
                } else {
                    const double y0r = pr[0] + dppm_f64<DPP_QUAD_XOR1>(pr[2]), y0i = pi[0] + dppm_f64<DPP_QUAD_XOR1>(pi[2]);
                    const double y1r = pr[1] + dppm_f64<DPP_QUAD_XOR1>(pr[3]), y1i = pi[1] + dppm_f64<DPP_QUAD_XOR1>(pi[3]);
                    yr = y0r + dppm_f64<DPP_QUAD_XOR2>(y1r); yi = y0i + dppm_f64<DPP_QUAD_XOR2>(y1i);
                    yr += dppm_f64<DPP_ROW_ROR4>(yr); yi += dppm_f64<DPP_ROW_ROR4>(yi);
                    yr += dppm_f64<DPP_ROW_ROR8>(yr); yi += dppm_f64<DPP_ROW_ROR8>(yi);
                }
                const double f = dt * c_series_inv[aord & 255];
                // forward: (-i f)(yr + i yi) = f yi - i f yr ; backward: (+i f)(yr + i yi) = -f yi + i f yr
                const double ur = BACKWARD ? -f * yi : f * yi;
                const double ui = BACKWARD ? f * yr : -f * yr;
                if (owner) {
                    vec[cur ^ 1][slot(myrow)] = make_double2(ur, ui);
                    sr += ur; si += ui;
                    if constexpr (!BACKWARD) {
                        if (pk && aord + 1 < a.maxp) pk[(size_t)(aord + 1) * NP] = make_double2(ur, ui);
                    }
                }
                // eight lanes hold a copy of each row's term: the ballot covers every row eight times, harmless
                const unsigned long long bal = __ballot(ur * ur + ui * ui > thr_el);
                if (lane == 0) flg[cur ^ 1][wave] = bal != 0ull;
                SDIAG(2);
                __syncthreads();
                SDIAG(3);
                cur ^= 1;
            }
            if (!conv) failed = true;
            if constexpr (!BACKWARD) {
                // u_0 .. u_aord were summed and u_aord is below the tolerance: M = aord terms for the derivative
                if (a.morder && tid == 0) a.morder[(size_t)k * N_T + n] = (pk && conv && aord < a.maxp) ? aord : -1;
            }
            terms += (unsigned long long)aord;
            ++substeps;
            // publish the new state as u_0 of the next (sub-)step; vec[cur ^ 1] is no longer read by anyone
            double n2 = 0.;
            if (owner) {
                if (BACKWARD && sa.xi && n > 0 && sub == msub - 1) {
                    // chi(t_n) += lambda_b Dt_n / rho_k xi_k(t_n)   (optimize.jl:897-908)
                    const double2 x_ = sa.xi[((size_t)k * (N_T + 1) + n) * NP + myrow];
                    const double c = sa.lambda_b * sa.wq[n] / rho_k;
                    sr += c * x_.x; si += c * x_.y;
                }
                vec[cur ^ 1][slot(myrow)] = make_double2(sr, si);
                n2 = sr * sr + si * si;
            }
            n2 = wave_sum_dpp(n2);
            if (lane == 0) red[cur ^ 1][wave] = n2;
            __syncthreads();
            nrm2 = 0.;
#pragma unroll
            for (int w = 0; w < NW; ++w) nrm2 += red[cur ^ 1][w];
            cur ^= 1;
        }
        if (owner) st[(size_t)(BACKWARD ? n : n + 1) * NP + myrow] = make_double2(sr, si);
        SDIAG(4);
    }
#ifdef GRAPE_SERIES_DIAG
    if (tid == 0 && k == 0) {
        printf("series diag (%s, block 0 wave 0): loads+flags %llu | fma %llu | reduce+write %llu | barrier %llu | step tail %llu cycles; "
               "total %llu cycles in %llu wall ticks (100 MHz) => %.0f MHz\n", BACKWARD ? "bwd" : "fwd", sd_t[0], sd_t[1], sd_t[2], sd_t[3],
               sd_t[4], clock64() - sd_c0, wall_clock64() - sd_wall0,
               (double)(clock64() - sd_c0) / (double)(wall_clock64() - sd_wall0) * 100.0);
    }
#endif
    if (!BACKWARD) {
        // tau_k = <target_k | Psi_k(T)>  (optimize.jl:753)
        if (wave == 0) {
            double pr = 0., pi = 0.;
            if (lane < sa.N) {
                const double2 t = sa.target[(size_t)k * sa.N + lane];
                const double2 p = vec[cur][slot(lane)];
                pr = t.x * p.x + t.y * p.y;
                pi = t.x * p.y - t.y * p.x;
            }
            pr = wave_sum_dpp(pr);
            pi = wave_sum_dpp(pi);
            if (lane == 0) sa.tau[k] = make_double2(pr, pi);
        }
    }
    if (tid == 0) {
        if (failed) atomicOr(&sa.flags[0], 16);
        stat_add(a.stats, 10, terms);
        stat_add(a.stats, 11, substeps);
    }
}

template <int NP, bool BACKWARD, int LR>
__global__ void __launch_bounds__(NP * 16 / SERIES_RPT) series_sweep_kernel(SeriesArgs a) {
    series_sweep_body<NP, BACKWARD, LR>(a, blockIdx.x);
}

// both sweeps in one launch (see sweep_pair_kernel): blocks [0, K) forward, [K, 2K) backward from the unit targets
template <int NP, int LR>
__global__ void __launch_bounds__(NP * 16 / SERIES_RPT) series_pair_kernel(SeriesArgs af, SeriesArgs ab) {
    if ((int)blockIdx.x < af.s.K) series_sweep_body<NP, false, LR>(af, blockIdx.x);
    else series_sweep_body<NP, true, LR>(ab, blockIdx.x - af.s.K);
}
