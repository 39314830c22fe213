// grape_deriv3.hip.h -- derivative overlaps <chi'_l | Psi> of the gradient (same two-pass series as deriv2_kernel,
// grape_kernels.hip.h; /root/reference/src/optimize.jl:876-911 and :604-653), ONE WAVE PER BATCH of 16 consecutive cells.
//
// deriv2_kernel gives the four waves of a workgroup one row tile each of the SAME batch: every order ends with the new vector
// block going through LDS and a barrier (2 K cycles of bubble per order-pass, 38 of them per batch), and the operator
// fragments a wave needs 19 times over are cached in the accumulation half of the register file (v_accvgpr_read in front of
// every matrix instruction that uses them: MfmaUtil 70 %).  Here a wave owns a whole batch:
//   * the new vector block never leaves the registers -- in the C/D layout of v_mfma_f64_16x16x4, register r of row tile t
//     IS the B operand of k-step 16 t + 4 r (Strip, grape_kernels.hip.h), so the result of one order is the right operand
//     of the next; no LDS round trip, no barrier, no cross-wave reduction of norms and overlaps;
//   * the operators H0_k, mu_1, mu_2 of the workgroup's trajectory are in LDS, shared by its four waves (four batches of
//     one trajectory at a time).  Hermitian operators only: the upper block triangle is stored (10 of 16 tiles, re and im,
//     131 KB for three matrices at N = 64 -- the full matrices would need 196 KB); a tile below the diagonal is read
//     transposed from its mirror image, and its conjugation costs nothing: P2 is issued with the negation bit of its left
//     operand (the BLGP field of v_mfma_f64), P3 takes the difference re - im as left operand.  Pass 2 needs H^dagger = H: the same tiles.
//     A general (non-Hermitian) drift H0_k beside Hermitian control operators -- effective Hamiltonians with decay -- is
//     stored with all its tiles (H0G); pass 2 then reads EVERY tile of it the mirrored way.
//   * k loops: matrix instructions, LDS reads and ONE vector addition per left operand (the operand sum of the 3M scheme;
//     a third LDS plane does not fit).
// Everything else (stopping rule, parked terms u_a, order of the additions within a cell) is deriv2_kernel's.
#pragma once
#include "grape_t18.hip.h"

struct Deriv3Args {
    Deriv2Args d;             // H0p / Hcp / H0q / Hcq unused
    const double *H0f, *Hcf;  // planar row-major operators [K][2][NP*NP], [Kc][L][2][NP*NP] (ExpmArgs layout)
    int wpt;                  // workgroups per trajectory
    int skip_if_flagged;      // N <= 32: leave the launch at once when deriv_flag_kernel found a batch that needs sub-steps
                              // (flags[3] != 0) -- deriv_kernel, launched behind this one, then does every cell
};

template <int NT>
struct D3Lds {
    static constexpr int LDT = 17, TILE = 16 * LDT;            // doubles per tile plane (odd stride: the 16 rows of a fragment fall into different banks)
    static constexpr int NTILE = NT * (NT + 1) / 2, MAT = NTILE * 2 * TILE;   // doubles per matrix
    static constexpr int tile(int ti, int tj) { return ti * NT - ti * (ti - 1) / 2 + (tj - ti); }   // ti <= tj, row by row
};

// right operand of a product: column strip of the vector block with the sums of the 3M scheme
// (the sum re + im of the 3M scheme is formed per k-step: one vector addition beside the NT of the left operands, against 32
// registers for a third strip component -- the kernel is at the register limit)
template <int NT>
struct D3Vec {
    d4 re[NT], im[NT];
};
template <int NT>
__device__ __forceinline__ void d3_finish(D3Vec<NT> &) {}
// v_mfma_f64 reads its BLGP field as negation bits of (A, B, C): c - a b without a vector instruction
#define MFMA64_NEGA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 1)

// storage of a matrix in LDS and the way a product reads it
enum { D3_HERM = 0,      // Hermitian matrix, upper tiles: M v (tiles below the diagonal from their mirror images)
       D3_FULL = 1,      // general matrix, all NT x NT tiles row by row: M v
       D3_FULL_ADJ = 2   // ... : M^dagger v (every tile from its transposed position, conjugated)
};
// q (re, im) = M v (or M^dagger v) for the matrix at `mat` (LDS)
template <int NT, int MODE = D3_HERM>
__device__ __forceinline__ void d3_product(Strip<NT> &q, const double *mat, const D3Vec<NT> &v, const int lane) {
    using LY = D3Lds<NT>;
    typedef const double __attribute__((address_space(3))) *lds_cptr;
    constexpr int LDT = LY::LDT, TILE = LY::TILE;
    // per-lane bases of a fragment: direct tile element [m][4 r + kq], mirrored tile element [4 r + kq][m]
    lds_cptr bd = (lds_cptr)(mat + (lane & 15) * LDT + (lane >> 4));
    lds_cptr bm = (lds_cptr)(mat + (lane >> 4) * LDT + (lane & 15));
    // is the fragment of (row tile rt, k tile kt) read directly, or transposed and conjugated from tile (kt, rt)?
    auto direct = [](int rt, int kt) constexpr { return MODE == D3_FULL ? true : MODE == D3_FULL_ADJ ? false : rt <= kt; };
    auto tile_of = [](int ti, int tj) constexpr { return MODE == D3_HERM ? LY::tile(ti, tj) : ti * NT + tj; };
    auto frag_re = [&](int rt, int kt, int r) __attribute__((always_inline)) -> double {
        return direct(rt, kt) ? bd[(tile_of(rt, kt) * 2) * TILE + 4 * r] : bm[(tile_of(kt, rt) * 2) * TILE + 4 * r * LDT];
    };
    auto frag_im = [&](int rt, int kt, int r) __attribute__((always_inline)) -> double {   // (mirrored: the stored value, sign handled below)
        return direct(rt, kt) ? bd[(tile_of(rt, kt) * 2 + 1) * TILE + 4 * r] : bm[(tile_of(kt, rt) * 2 + 1) * TILE + 4 * r * LDT];
    };
    d4 p1[NT], p2[NT], p3[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { p1[t] = (d4){0., 0., 0., 0.}; p2[t] = (d4){0., 0., 0., 0.}; p3[t] = (d4){0., 0., 0., 0.}; }
    double are[NT], aim[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) { are[rt] = frag_re(rt, 0, 0); aim[rt] = frag_im(rt, 0, 0); }
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool more = !(kt == NT - 1 && r == 3);
            const int ktn = r < 3 ? kt : kt + 1, rn = r < 3 ? r + 1 : 0;
            const double bre = v.re[kt][r], bim = v.im[kt][r], bsm = bre + bim;
            double as[NT];
#pragma unroll
            for (int rt = 0; rt < NT; ++rt) as[rt] = direct(rt, kt) ? are[rt] + aim[rt] : are[rt] - aim[rt];
#pragma unroll
            for (int rt = 0; rt < NT; ++rt) p1[rt] = MFMA64(are[rt], bre, p1[rt]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int rt = 0; rt < NT; ++rt) are[rt] = frag_re(rt, ktn, rn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rt = 0; rt < NT; ++rt) p2[rt] = direct(rt, kt) ? MFMA64(aim[rt], bim, p2[rt]) : MFMA64_NEGA(aim[rt], bim, p2[rt]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int rt = 0; rt < NT; ++rt) aim[rt] = frag_im(rt, ktn, rn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rt = 0; rt < NT; ++rt) p3[rt] = MFMA64(as[rt], bsm, p3[rt]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        q.re[t] = p1[t] - p2[t];
        q.im[t] = p3[t] - p1[t] - p2[t];
    }
}

// H0G: the drift H0_k is a general matrix (effective non-Hermitian Hamiltonians: decay terms) while the control operators
// are Hermitian -- H0_k is stored with all its tiles, pass 1 reads it directly and pass 2 as its adjoint
template <int NT, int LMAX, bool H0G = false>
__global__ void __launch_bounds__(256) deriv3_kernel(Deriv3Args g) {
    using LY = D3Lds<NT>;
    constexpr int NP = 16 * NT;
    constexpr int H0SZ = H0G ? NT * NT * 2 * LY::TILE : LY::MAT;   // doubles of the drift in LDS; control l follows at H0SZ + l MAT
    extern __shared__ __attribute__((aligned(16))) double d3sm[];   // [1 + LMAX][NTILE][2][16][17]
    const Deriv2Args &a = g.d;
    if (g.skip_if_flagged && a.flags[3] != 0) return;   // (uniform)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, rg = lane >> 4;
    const int L = a.L;
    constexpr size_t vplane = (size_t)NP * 16, pp = (size_t)NP * NP;
    double *park = a.park + ((size_t)blockIdx.x * 4 + wave) * a.maxm * 2 * vplane;
    const int nslots = a.K * g.wpt;
    const int mcap = a.max_order < a.maxm ? a.max_order : a.maxm;
    int k_loaded = -1;
    for (int slot = blockIdx.x; slot < nslots; slot += gridDim.x) {
        const int k = slot / g.wpt, part = slot - k * g.wpt;
        if (k != k_loaded) {   // (uniform) the operators of trajectory k: upper tiles of H0_k and of the control operators
            __syncthreads();
            const int row = tid >> 4, col = tid & 15;
            for (int m = 0; m <= L; ++m) {
                const double *src = m == 0 ? g.H0f + (size_t)k * 2 * pp
                                           : g.Hcf + ((size_t)(a.hc_per_traj ? k : 0) * L + (m - 1)) * 2 * pp;
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) {
                        const bool full = H0G && m == 0;
                        if (tj < ti && !full) continue;
                        const size_t o = (size_t)(16 * ti + row) * NP + 16 * tj + col;
                        const int q_ = full ? ti * NT + tj : LY::tile(ti, tj);
                        double *dst = d3sm + (m == 0 ? (size_t)0 : (size_t)H0SZ + (size_t)(m - 1) * LY::MAT) + (size_t)q_ * 2 * LY::TILE
                                      + row * LY::LDT + col;
                        dst[0] = src[o];
                        dst[LY::TILE] = src[pp + o];
                    }
            }
            __syncthreads();
            k_loaded = k;
        }
        for (int bq = part * 4 + wave; bq < a.batches_per_k; bq += 4 * g.wpt) {
            const int batch = k * a.batches_per_k + bq;
            const int n0 = bq * 16, n = n0 + c;
            const bool valid = n < a.N_T;
            const int nc = valid ? n : a.N_T - 1;
            const double dt = a.dts[nc];
            double e[LMAX], sh[LMAX];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {
                sh[l] = (l < L && a.shape) ? a.shape[(size_t)l * a.N_T + nc] : 1.0;
                e[l] = l < L ? a.eps[(size_t)l * a.N_T + nc] * sh[l] : 0.;
            }
            // H v = H0 v + sum_l e_l mu_l v, one matrix at a time; `each(l, q)` sees mu_l v before it is folded in
            // (adj: pass 2 applies H^dagger -- the same tiles for Hermitian operators, the adjoint reading of a general drift)
            auto apply = [&](const D3Vec<NT> &v, Strip<NT> &sum, auto each, auto adj) __attribute__((always_inline)) {
                if constexpr (!H0G) d3_product<NT, D3_HERM>(sum, d3sm, v, lane);
                else if constexpr (decltype(adj)::value) d3_product<NT, D3_FULL_ADJ>(sum, d3sm, v, lane);
                else d3_product<NT, D3_FULL>(sum, d3sm, v, lane);
#pragma unroll
                for (int l = 0; l < LMAX; ++l) {
                    if (l < L) {
                        Strip<NT> q;
                        d3_product<NT, D3_HERM>(q, d3sm + (size_t)H0SZ + (size_t)l * LY::MAT, v, lane);
                        each(l, q);
#pragma unroll
                        for (int t = 0; t < NT; ++t) { sum.re[t] += e[l] * q.re[t]; sum.im[t] += e[l] * q.im[t]; }
                    }
                }
            };
            // ---- pass 1: u_0 = Psi(t_n), u_{a+1} = (-i dt / (a+1)) H u_a ----
            D3Vec<NT> v;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * t + 4 * r + rg;
                    const double2 p = a.fw[((size_t)k * (a.N_T + 1) + nc) * NP + row];
                    v.re[t][r] = valid ? p.x : 0.; v.im[t][r] = valid ? p.y : 0.;
                    const size_t o = (size_t)row * 16 + c;
                    park[o] = v.re[t][r]; park[vplane + o] = v.im[t][r];
                }
            d3_finish<NT>(v);
            int M = 0, converged = 0;
            // round 6: degree of the economized polynomial this batch is certified for (asm/gen_d3.py's header), 0: none
            const int deg = (a.batch_econ && a.batch_flag && a.econ_pairs) ? a.batch_flag[a.nbatch_total + batch] : 0;
            const int capb = (deg && deg - 1 <= mcap) ? deg - 1 : mcap;
            const double *pairs = nullptr;
            for (int m = 1; m <= capb; ++m) {   // forms u_m
                Strip<NT> sum;
                apply(v, sum, [](int, const Strip<NT> &) {}, std::false_type());
                const double sfac = dt / (double)m;
                double nn = 0.;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double ur = sfac * sum.im[t][r], ui = -sfac * sum.re[t][r];   // (-i s)(x + i y) = s y - i s x
                        v.re[t][r] = ur; v.im[t][r] = ui;
                        if (m < a.maxm) {
                            const size_t o = (size_t)m * 2 * vplane + (size_t)(16 * t + 4 * r + rg) * 16 + c;
                            park[o] = ur; park[vplane + o] = ui;
                        }
                        nn += ur * ur + ui * ui;
                    }
                d3_finish<NT>(v);
                nn += __shfl_xor(nn, 16, 64);
                nn += __shfl_xor(nn, 32, 64);
                M = m;
                if (m >= 2 && __all(nn < a.tol * a.tol)) { converged = 1; break; }   // ||u_m|| < tol for every cell of the batch
            }
            if (!converged && capb != mcap) {   // certified: the deg - 1 orders formed are all the polynomial needs
                M = deg; converged = 1;
                pairs = a.econ_pairs + (size_t)(deg - 16) * 64;
            }
            // ---- pass 2: w_{M-1} = chi(t_{n+1}), w_{a-1} = chi + (i dt / (a+1)) H^dagger w_a ----
            Strip<NT> chi;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * t + 4 * r + rg;
                    const double2 x = a.bw[((size_t)k * (a.N_T + 1) + nc + 1) * NP + row];
                    chi.re[t][r] = valid ? x.x : 0.; chi.im[t][r] = valid ? x.y : 0.;
                    v.re[t][r] = chi.re[t][r]; v.im[t][r] = chi.im[t][r];
                }
            d3_finish<NT>(v);
            double dr[LMAX], di[LMAX];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) { dr[l] = 0.; di[l] = 0.; }
            for (int aa = M - 1; aa >= 0; --aa) {
                const double inv = pairs ? pairs[2 * aa] : 1.0 / (double)(aa + 1), sfac = dt * (pairs ? pairs[2 * aa + 1] : inv);
                Strip<NT> u;   // u_aa (requested before the products)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const size_t o = (size_t)aa * 2 * vplane + (size_t)(16 * t + 4 * r + rg) * 16 + c;
                        u.re[t][r] = park[o]; u.im[t][r] = park[vplane + o];
                    }
                Strip<NT> sum;
                apply(v, sum, [&](int l, const Strip<NT> &q) {
                    double sr = 0., si = 0.;
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {   // conj(mu_l^dagger w) * u
                            sr += q.re[t][r] * u.re[t][r] + q.im[t][r] * u.im[t][r];
                            si += q.re[t][r] * u.im[t][r] - q.im[t][r] * u.re[t][r];
                        }
#pragma unroll
                    for (int ll = 0; ll < LMAX; ++ll)
                        if (ll == l) { dr[ll] += inv * sr; di[ll] += inv * si; }
                }, std::true_type());
                if (aa > 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {   // chi + (i s)(x + i y) = chi - s y + i s x
                        v.re[t] = chi.re[t] - sfac * sum.im[t];
                        v.im[t] = chi.im[t] + sfac * sum.re[t];
                    }
                    d3_finish<NT>(v);
                }
            }
            // ---- tau_grads = rho (-i dt s_l) sum_a <mu_l^dagger w_a | u_a> / (a+1) ----
            const double rho = a.rho[k];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {
                if (l < L) {
                    double Dr = dr[l], Di = di[l];
                    Dr += __shfl_xor(Dr, 16, 64); Di += __shfl_xor(Di, 16, 64);
                    Dr += __shfl_xor(Dr, 32, 64); Di += __shfl_xor(Di, 32, 64);
                    if (lane < 16 && valid) {
                        const double f = rho * dt * sh[l];   // (-i f)(Dr + i Di) = f Di - i f Dr
                        a.tg[((size_t)k * L + l) * a.N_T + n] = make_double2(f * Di, -f * Dr);
                    }
                }
            }
            if (lane == 0) {
                const bool redone = a.batch_flag && a.batch_flag[batch];
                // (not converged within the terms this kernel parks while taylor_grad_max_order allows more: flags[7] asks for
                // deriv_kernel, which is launched behind this kernel and honours any order; round-3 advisor finding)
                if (!converged && !redone) {
                    if (a.deep_redo && a.max_order > mcap) atomicAdd(&a.flags[7], 1);
                    else atomicOr(&a.flags[0], 4);
                }
                if (!redone) stat_add(a.stats, 8, (unsigned long long)M * (unsigned long long)min(16, a.N_T - n0));
            }
        }
    }
}
