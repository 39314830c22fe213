// grape_t18.hip.h -- inverse-free exponential of a skew-Hermitian A = -i dt H (Hermitian generators), gfx950.
//
// Replaces the `exp` inside ExpProp's prop_step! (/root/reference/src/optimize.jl:732, 881, 972) for Hermitian generators;
// the order-13 Pade kernel of grape_kernels.hip.h stays the path of every other generator and the parity reference.
//
// Why: the Pade approximant needs the solve (V - U) X = V + U -- four serial 16 x 16 tile inversions per 64 x 64 cell,
// 47 K of the 125 K cycles of a cell with the matrix pipe almost idle.  A polynomial needs no solve, and for a NORMAL
// matrix its error is a scalar statement on the spectrum: with H Hermitian and spec(H dt) in [-beta, beta],
//     || p(-i H dt) - exp(-i H dt) ||_2 = max_{|lam| <= beta} | p(-i lam) - exp(-i lam) |.
// p is the degree-18 Chebyshev truncation of exp on the segment i[-2, 2] (error 1.6e-17, tools/t18_coeffs.py), evaluated
// with FIVE matrix products by the scheme of Bader, Blanes, Casas (2019) for degree 18:
//     A2 = A A, A3 = A A2, A6 = A3 A3                  (Hermitian / skew-Hermitian: NT - 1 of NT row tiles per strip)
//     B1 = a1 A + a2 A2 + a3 A3,  B5 = e2 A2 + e3 A3 + e6 A6,  B4 = d0 I + d1 A + d2 A2 + d3 A3 + d6 A6
//     A9 = B1 B5 + B4                                   (general product)
//     B3 = c0 I + c1 A + c2 A2 + c3 A3 + c6 A6,  B2 = b0 I + b1 A + b2 A2 + b3 A3 + b6 A6   (d0 = 0, b0 = 1)
//     p(A) = B2 + (B3 + A9) A9                          (general product)
// 768 matrix instructions per wave at N = 64 against 816 + 264 (products + solve) of the Pade route, and none of them
// waits for a tile inversion.
// The spectral bound is rigorous and costs two column-sum passes over register strips: rho(H dt)^2 <= ||A2||_1 and
// rho(H dt)^6 <= ||A6||_1 (Hermitian matrices: ||M||_2 <= ||M||_1; |re| + |im| stands in for the modulus, which can only
// enlarge the bound).  beta = min(sqrt ||A2||_1, ||A6||_1^(1/6)); beta > 2: A is scaled by 2^-s (the powers by 2^-ks,
// exact) and the result squared s times.  At the headline configuration ||A||_1 = 4.2 but beta = 1.17: no squaring.
//
// Cost model behind the layout (tools/mfma_filler_probe.hip, profiles/r03_mfma_f64_filler_probe.txt; one wave per SIMD):
// a v_mfma_f64_16x16x4 occupies the vector ALU for its 64 cycles -- NOTHING the wave issues to the VALU runs underneath
// it (PMC SQ_VALU_MFMA_COEXEC_CYCLES = 0): the first vector instruction between two matrix instructions costs 24
// cycles, every further one 4.6-4.9, a v_accvgpr_read/write 8.4; only LDS reads (up to ~8 per gap) and scalar
// instructions are free.  Hence
//   * the k loops contain matrix instructions and LDS reads ONLY: the operand sums of the 3M scheme come from a third
//     LDS plane (re + im, written once by whoever writes the left operand) and a third strip component (formed once per
//     right operand) instead of 5 additions per k-step;
//   * the linear combinations are formed in ONE pass over the powers, right after the third product (each power is
//     read by the VALU once instead of four or five times -- with 450 live registers most of them live in accumulation
//     registers and every VALU pass over them is a stream of v_accvgpr_read); B4 and B2 enter their products as start
//     values of the accumulators.
#pragma once
#include "grape_kernels.hip.h"

// diagnostic builds only (tools/t18_ablate.sh): -DT18_STOP=n leaves the cell after phase n (results are wrong; the
// differences of the launch times are the cost of the phases -- in-kernel stamps perturb this kernel by a third)
#ifdef T18_STOP
#define T18_STOP_AT(n, Uout, Src) do { if (T18_STOP == (n)) { _Pragma("unroll") for (int t_ = 0; t_ < NT; ++t_) { (Uout).re[t_] = (Src).re[t_ % (sizeof((Src).re) / sizeof((Src).re[0]))]; (Uout).im[t_] = (Src).im[t_ % (sizeof((Src).im) / sizeof((Src).im[0]))]; } s_out = 0; bad = false; return; } } while (0)
#else
#define T18_STOP_AT(n, Uout, Src) do {} while (0)
#endif

#include "grape_t18_coeffs.h"

// coefficient sets: Chebyshev truncation on i[-2, 2] (Hermitian generators) / Taylor polynomial (general matrices)
template <bool HERM> struct T18Coef;
template <> struct T18Coef<true> {
    static constexpr double A1 = T18_A1, A2 = T18_A2, A3 = T18_A3, B1 = T18_B1, B2 = T18_B2, B3 = T18_B3, B6 = T18_B6,
        C0 = T18_C0, C1 = T18_C1, C2 = T18_C2, C3 = T18_C3, C6 = T18_C6, D0 = T18_D0, D1 = T18_D1, D2 = T18_D2, D3 = T18_D3,
        D6 = T18_D6, E2 = T18_E2, E3 = T18_E3, E6 = T18_E6, B0 = T18_B0, THETA = T18_THETA;
};
template <> struct T18Coef<false> {
    static constexpr double A1 = T18T_A1, A2 = T18T_A2, A3 = T18T_A3, B1 = T18T_B1, B2 = T18T_B2, B3 = T18T_B3, B6 = T18T_B6,
        C0 = T18T_C0, C1 = T18T_C1, C2 = T18T_C2, C3 = T18T_C3, C6 = T18T_C6, D0 = T18T_D0, D1 = T18T_D1, D2 = T18T_D2, D3 = T18T_D3,
        D6 = T18T_D6, E2 = T18T_E2, E3 = T18T_E3, E6 = T18T_E6, B0 = T18T_B0, THETA = T18T_THETA;
};

// LDS carve: ONE left-operand region of three planes (re, im, re + im; leading dimension NP + 2), two exchange areas
// (partial sums of the doubly computed tile, mirrored tiles) and the reduction scratch
template <int NT>
struct T18Lds {
    static constexpr int NP = 16 * NT, LD = NP + 2, NTH = NT * 64;
    static constexpr int PL = NP * LD;          // doubles per plane
    // (the area of the partial sums exists at four tiles per side only -- without it the three-tile layout is 72 KB and two
    // workgroups fit a CU)
    static constexpr int E1 = 3 * PL, E2 = E1 + (NT == 4 ? NT * 512 : 0), RED = E2 + NT * 512;
    static constexpr int TOTAL = RED + NTH + 8 + NP;   // doubles
};

// a right operand: rotated column strip (slot s of wave w = row tile (w + s) % NT) with the sums re + im of the 3M scheme
template <int NT>
struct Strip3M {
    d4 re[NT], im[NT], sm[NT];
};
// the three partial products of the 3M scheme: re = p1 - p2, im = p3 - p1 - p2
template <int NS>
struct Acc3 {
    d4 p1[NS], p2[NS], p3[NS];
};
template <int NS>
__device__ __forceinline__ void acc3_zero(Acc3<NS> &q) {
#pragma unroll
    for (int t = 0; t < NS; ++t) { q.p1[t] = (d4){0., 0., 0., 0.}; q.p2[t] = (d4){0., 0., 0., 0.}; q.p3[t] = (d4){0., 0., 0., 0.}; }
}

// max_j sum_i (|re| + |im|) over the wave's 16 columns of a complete rotated strip (all NT slots): an upper bound of the
// largest column sum of moduli of this column strip; uniform over the wave
template <int NT, class S>
__device__ __forceinline__ double t18_colsum_max(const S &s) {
    double c = 0.;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) c += fabs(s.re[t][r]) + fabs(s.im[t][r]);
    c += __shfl_xor(c, 16, 64);   // the four lane rows hold the same column
    c += __shfl_xor(c, 32, 64);
    c = fmax(c, dpp_f64<DPP_QUAD_XOR1>(c));
    c = fmax(c, dpp_f64<DPP_QUAD_XOR2>(c));
    c = fmax(c, dpp_f64<DPP_ROW_HALF_MIRROR>(c));
    c = fmax(c, dpp_f64<DPP_ROW_MIRROR>(c));
    return readlane_f64(c, 0);
}

struct T18NoHook {
    __device__ __forceinline__ void operator()(int, int) const {}
};

// q[0..NS-1] += X * B for rotated strips: X in the three LDS planes R (natural layout), B a Strip3M.  The k loop issues
// matrix instructions and LDS reads only (see the cost model above): the operands of k-step ks+1 are requested one
// plane at a time behind the matrix instructions that consumed that plane's registers.  hook(sk, r) runs in front of
// k-step r of k-block sk (late arrival of a mirrored tile, stores of the previous cell's result, requests for the next cell's
// operator tiles).  Measured (tools/ab_lib.py, C3): the 16 result stores of a lane as four per k-block 17.33 ms, as one per
// k-step 17.51 ms (every store brings its address arithmetic, i.e. a matrix -> vector -> matrix transition of 24 cycles,
// into the k-step), all 16 behind the last product 17.60 ms.
// HALF_LAST (NT = 4, squares of a (skew-)Hermitian matrix, X == B): the last computed slot, tile (w+2, w), is also computed
// -- as its conjugate transpose -- by wave w+2; the two are adjoint term by term, so each wave sums only the first two
// k-blocks of its rotated k order and the caller adds the partner's partial sum (gemm_rot in grape_kernels.hip.h).
template <int LD, int NS, int NT, bool HALF_LAST, class Hook>
__device__ __forceinline__ void t18_gemm(Acc3<NS> &q, const double *__restrict__ R, const Strip3M<NT> &B, int wave, int lane,
                                         Hook hook) {
    constexpr int PL = 16 * NT * LD;
    // one base address per slot for the planes re / im (the im plane is an immediate offset of 8 PL <= 33792 bytes away)
    // and ONE MORE per slot for the third plane: 16 PL bytes exceed the 16-bit offset field of ds_read, and with a single
    // base the compiler kept a table of per-k-step addresses in accumulation registers -- three v_accvgpr_read per
    // k-step, i.e. vector instructions between the matrix instructions of every k-step (8 % of a product)
    // (pointers in the LDS address space by TYPE: behind the register pin below the compiler no longer sees where a generic
    // pointer came from and reads the third plane with flat_load -- 64-bit addresses, the vector-memory path and its counter)
    typedef const double __attribute__((address_space(3))) *lds_cptr;
    lds_cptr xp[NS], xq[NS];
#pragma unroll
    for (int so = 0; so < NS; ++so) {
        xp[so] = (lds_cptr)(R + (lane & 15) * LD + (lane >> 4) + 16 * ((wave + so) % NT) * LD);
        xq[so] = xp[so] + 2 * PL;
        // (NT = 4: keep it a register of its own -- folded back into xp[so] + constant it is the table again; NT = 3: all
        // three planes are within the offset field of one base)
        if constexpr (8 * (3 * PL) > 65535) asm volatile("" : "+v"(xq[so]));
    }
    double are[NS], aim[NS], asu[NS];
    {
        const int k0 = 16 * wave;
#pragma unroll
        for (int so = 0; so < NS; ++so) { are[so] = xp[so][k0]; aim[so] = xp[so][PL + k0]; asu[so] = xq[so][k0]; }
    }
#pragma unroll
    for (int sk = 0; sk < NT; ++sk) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hook(sk, r);
            const int kn = (r < 3) ? 16 * ((wave + sk) % NT) + 4 * (r + 1) : 16 * ((wave + sk + 1) % NT);   // next k column
            const bool more = !(sk == NT - 1 && r == 3);
            const int ns = (HALF_LAST && 2 * sk >= NT) ? NS - 1 : NS;                                    // slots of this k-step
            const int nsn = (HALF_LAST && (2 * sk >= NT || (2 * (sk + 1) >= NT && r == 3))) ? NS - 1 : NS;   // ... of the next one
            const double bre = B.re[sk][r], bim = B.im[sk][r], bsm = B.sm[sk][r];
#pragma unroll
            for (int so = 0; so < NS; ++so)
                if (so < ns) q.p1[so] = MFMA64(are[so], bre, q.p1[so]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int so = 0; so < NS; ++so)
                    if (so < nsn) are[so] = xp[so][kn];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int so = 0; so < NS; ++so)
                if (so < ns) q.p2[so] = MFMA64(aim[so], bim, q.p2[so]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int so = 0; so < NS; ++so)
                    if (so < nsn) aim[so] = xp[so][PL + kn];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int so = 0; so < NS; ++so)
                if (so < ns) q.p3[so] = MFMA64(asu[so], bsm, q.p3[so]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int so = 0; so < NS; ++so)
                    if (so < nsn) asu[so] = xq[so][kn];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// slots 0..NS-1 of a rotated strip to their natural positions in the three planes
template <int LD, int NS, int NT, class S>
__device__ __forceinline__ void t18_store_slots(double *R, const S &s, int wave, int lane) {
    constexpr int PL = 16 * NT * LD;
    double *x = R + (lane >> 4) * LD + 16 * wave + (lane & 15);
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        const int tb = (wave + sl) % NT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = (16 * tb + 4 * r) * LD;
            x[o] = s.re[sl][r];
            x[PL + o] = s.im[sl][r];
            x[2 * PL + o] = s.re[sl][r] + s.im[sl][r];
        }
    }
}
// (signed) conjugate transpose of the slot-1 tile (row block w+1, column block w) into position (w, w+1), three planes
template <int LD, int NT>
__device__ __forceinline__ void t18_store_adjoint(double *R, const d4 &tre, const d4 &tim, int wave, int lane, double sgn) {
    constexpr int PL = 16 * NT * LD;
    const int tb = (wave + 1) % NT, c = lane & 15, rg = lane >> 4;
    double *x = R + (16 * wave + c) * LD + 16 * tb + rg;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double vr = sgn * tre[r], vi = -sgn * tim[r];
        x[4 * r] = vr;
        x[PL + 4 * r] = vi;
        x[2 * PL + 4 * r] = vr + vi;
    }
}
template <int LD, int NT>
__device__ __forceinline__ void t18_load_strip(const double *R, Strip3M<NT> &s, int wave, int lane) {
    constexpr int PL = 16 * NT * LD;
    const double *x = R + (lane >> 4) * LD + 16 * wave + (lane & 15);
#pragma unroll
    for (int sl = 0; sl < NT; ++sl) {
        const int tb = (wave + sl) % NT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = (16 * tb + 4 * r) * LD;
            s.re[sl][r] = x[o]; s.im[sl][r] = x[PL + o]; s.sm[sl][r] = x[2 * PL + o];
        }
    }
}
// mirrored slot NT-1 of a (skew-)Hermitian matrix whose planes are complete
template <int LD, int NT>
__device__ __forceinline__ void t18_load_slot_last(const double *R, Strip3M<NT> &s, int wave, int lane) {
    constexpr int PL = 16 * NT * LD;
    const int tb = (wave + NT - 1) % NT;
    const double *x = R + ((lane >> 4) + 16 * tb) * LD + 16 * wave + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s.re[NT - 1][r] = x[4 * r * LD]; s.im[NT - 1][r] = x[PL + 4 * r * LD]; s.sm[NT - 1][r] = x[2 * PL + 4 * r * LD];
    }
}
// exchange areas (NT waves x 2 planes x 256 doubles): see rot_exch_write / rot_exch_add / rot_exch_read
template <int NT, int SLOT>
__device__ __forceinline__ void t18_exch_add(const double *area, Strip3M<NT> &s, int wave, int lane) {
    const double *src = area + wave * 512 + lane * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) { s.re[SLOT][r] += src[r]; s.im[SLOT][r] += src[256 + r]; }
}
template <int NT>
__device__ __forceinline__ void t18_exch_read_last(const double *area, Strip3M<NT> &s, int wave, int lane) {
    const double *src = area + wave * 512 + lane * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) { s.re[NT - 1][r] = src[r]; s.im[NT - 1][r] = src[256 + r]; }
    s.sm[NT - 1] = s.re[NT - 1] + s.im[NT - 1];
}
// slots 0..NS-1 of S from the partial products
template <int NS, int NT>
__device__ __forceinline__ void t18_combine(const Acc3<NS> &q, Strip3M<NT> &s) {
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        s.re[t] = q.p1[t] - q.p2[t];
        s.im[t] = q.p3[t] - q.p1[t] - q.p2[t];
    }
}

// Formation of a cell's A = -i dt H in the three planes (upper block triangle fetched, both triangles written; see
// expm_form_a_herm64): issue() requests the operator tiles, commit() combines them with the pulse values.
template <int NTH, int NT>
struct T18FormA {
    static constexpr int NP = 16 * NT, LD = NP + 2, HALF = NP * NP / 2, PL = NP * LD;
    static constexpr int NTILE = NT * (NT + 1) / 2, NPAIR = 128 * NTILE, NU = (NPAIR + NTH - 1) / NTH;
    const ExpmArgs &a;
    double *R;
    int cell, t;
    double2 hr[NU], hi[NU], c0r[NU], c0i[NU], c1r[NU], c1i[NU];
    __device__ __forceinline__ T18FormA(const ExpmArgs &a_, double *R_, int cell_, int t_) : a(a_), R(R_), cell(cell_), t(t_) {}
    // tile q of the upper block triangle, row by row: (ti, tj), ti <= tj
    static constexpr int tile_i(int q) {
        int ti = 0, r = q;
        for (int it = 0; it < NT - 1; ++it)
            if (r >= NT - ti) { r -= NT - ti; ++ti; }
        return ti;
    }
    static constexpr int tile_j(int q) {
        int ti = 0, r = q;
        for (int it = 0; it < NT - 1; ++it)
            if (r >= NT - ti) { r -= NT - ti; ++ti; }
        return ti + r;
    }
    __device__ __forceinline__ void locate(int u, int &i, int &j, bool &diag) const {
        if constexpr (NTH == 256) {
            // element pair t + 256 u: tile 2u + (t >> 7), position t & 127 inside it -- the tile indices are compile-time
            // constants up to one select (the general form below costs ~120 vector instructions per cell for index
            // arithmetic, and nothing issued to the vector ALU is free in this kernel)
            const bool hi = (t >> 7) != 0;
            const int idx = t & 127;
            const int q0 = 2 * u < NTILE ? 2 * u : NTILE - 1, q1 = 2 * u + 1 < NTILE ? 2 * u + 1 : NTILE - 1;
            const int ti = hi ? tile_i(q1) : tile_i(q0), tj = hi ? tile_j(q1) : tile_j(q0);
            i = 16 * ti + (idx >> 3);
            j = 16 * tj + 2 * (idx & 7);
            diag = ti == tj;
            return;
        }
        const int ep = min(t + u * NTH, NPAIR - 1), q = ep >> 7, idx = ep & 127;   // (surplus threads repeat the last pair)
        int ti = 0, r = q;
#pragma unroll
        for (int it = 0; it < NT - 1; ++it)
            if (r >= NT - ti) { r -= NT - ti; ++ti; }
        const int tj = ti + r;
        i = 16 * ti + (idx >> 3);
        j = 16 * tj + 2 * (idx & 7);
        diag = ti == tj;
    }
    __device__ __forceinline__ void issue() {
        const int kc = cell / a.N_T;
        const int k = a.rep ? a.rep[kc] : kc;
        const double2 *h0 = (const double2 *)(a.H0f + (size_t)k * 2 * NP * NP);
        // (a.Sf: the controls of this time step are already summed -- ONE "control" with coefficient 1, see ExpmArgs)
        const double2 *hc = a.Sf ? (const double2 *)(a.Sf + (size_t)(a.hc_per_traj ? cell : cell - kc * a.N_T) * 2 * NP * NP)   // (per-trajectory control operators: summed per CELL)
                                 : (const double2 *)(a.Hcf + (size_t)(a.hc_per_traj ? k : 0) * a.L * 2 * NP * NP);
        const size_t o1 = (a.L > 1 && !a.Sf) ? (size_t)2 * HALF : 0;   // (one control: the same tile again, unused)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            int i, j; bool dg;
            locate(u, i, j, dg);
            const int off = (i * NP + j) >> 1;
            hr[u] = h0[off]; hi[u] = h0[HALF + off];
            c0r[u] = hc[off]; c0i[u] = hc[HALF + off];
            c1r[u] = hc[o1 + off]; c1i[u] = hc[o1 + HALF + off];
        }
    }
    __device__ __forceinline__ void commit() {
        const int kc = cell / a.N_T, n = cell - kc * a.N_T;
        const int k = a.rep ? a.rep[kc] : kc;
        const double2 *hc = (const double2 *)(a.Hcf + (size_t)(a.hc_per_traj ? k : 0) * a.L * 2 * NP * NP);
        const double dt = a.dts[n];
        const int L = a.Sf ? 1 : a.L;
        double e[8];
        for (int l = 0; l < L; ++l) {
            e[l] = a.eps[(size_t)l * a.N_T + n];
            if (a.shape) e[l] *= a.shape[(size_t)l * a.N_T + n];
        }
        if (a.Sf) e[0] = 1.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            int i, j; bool dg;
            locate(u, i, j, dg);
            double2 xr = hr[u], xi = hi[u];
            xr.x = fma(e[0], c0r[u].x, xr.x); xr.y = fma(e[0], c0r[u].y, xr.y);
            xi.x = fma(e[0], c0i[u].x, xi.x); xi.y = fma(e[0], c0i[u].y, xi.y);
            if (L > 1) {
                xr.x = fma(e[1], c1r[u].x, xr.x); xr.y = fma(e[1], c1r[u].y, xr.y);
                xi.x = fma(e[1], c1i[u].x, xi.x); xi.y = fma(e[1], c1i[u].y, xi.y);
            }
            const int off = (i * NP + j) >> 1;
            for (int l = 2; l < L; ++l) {   // (more than two controls, per trajectory: fetched here)
                const double2 cr = hc[(size_t)l * 2 * HALF + off], ci = hc[(size_t)l * 2 * HALF + HALF + off];
                xr.x = fma(e[l], cr.x, xr.x); xr.y = fma(e[l], cr.y, xr.y);
                xi.x = fma(e[l], ci.x, xi.x); xi.y = fma(e[l], ci.y, xi.y);
            }
            const double ar0 = dt * xi.x, ar1 = dt * xi.y, ai0 = -dt * xr.x, ai1 = -dt * xr.y;
            double *p = R + i * LD + j;
            p[0] = ar0; p[1] = ar1;
            p[PL] = ai0; p[PL + 1] = ai1;
            p[2 * PL] = ar0 + ai0; p[2 * PL + 1] = ar1 + ai1;
            if (!dg) {   // mirrored tile: a_ji = -conj(a_ij)
                double *m = R + j * LD + i;
                m[0] = -ar0; m[LD] = -ar1;
                m[PL] = ai0; m[PL + LD] = ai1;
                m[2 * PL] = ai0 - ar0; m[2 * PL + LD] = ai1 - ar1;
            }
        }
    }
};

// General (non-Hermitian) generators: every element is fetched; NBATCH batches of element pairs over NTH threads (each
// batch keeps 6 x 16 bytes per pair and thread in flight: four batches stay inside the register budget of this kernel)
template <int NTH, int NT>
__device__ __forceinline__ void t18_form_a_general(const ExpmArgs &a, double *R, const int cell, const int t) {
    constexpr int NP = 16 * NT, LD = NP + 2, HALF = NP * NP / 2, PL = NP * LD;
    constexpr int NBATCH = 4;
    constexpr int NU = (HALF / NBATCH + NTH - 1) / NTH;   // pairs per thread and batch
    const int kc = cell / a.N_T, n = cell - kc * a.N_T;
    const int k = a.rep ? a.rep[kc] : kc;
    const double2 *h0 = (const double2 *)(a.H0f + (size_t)k * 2 * NP * NP);
    const double2 *hc = a.Sf ? (const double2 *)(a.Sf + (size_t)(a.hc_per_traj ? cell : n) * 2 * NP * NP)
                             : (const double2 *)(a.Hcf + (size_t)(a.hc_per_traj ? k : 0) * a.L * 2 * NP * NP);
    const int L = a.Sf ? 1 : a.L;   // (summed controls: one "control" with coefficient 1)
    const size_t o1 = L > 1 ? (size_t)2 * HALF : 0;
    const double dt = a.dts[n];
    double e[8];
    for (int l = 0; l < L; ++l) {
        e[l] = a.eps[(size_t)l * a.N_T + n];
        if (a.shape) e[l] *= a.shape[(size_t)l * a.N_T + n];
    }
    if (a.Sf) e[0] = 1.0;
    for (int b = 0; b < NBATCH; ++b) {
        double2 hr[NU], hi[NU], c0r[NU], c0i[NU], c1r[NU], c1i[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int off = min(b * (HALF / NBATCH) + t + u * NTH, HALF - 1);
            hr[u] = h0[off]; hi[u] = h0[HALF + off];
            c0r[u] = hc[off]; c0i[u] = hc[HALF + off];
            c1r[u] = hc[o1 + off]; c1i[u] = hc[o1 + HALF + off];
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int off = min(b * (HALF / NBATCH) + t + u * NTH, HALF - 1);
            double2 xr = hr[u], xi = hi[u];
            xr.x = fma(e[0], c0r[u].x, xr.x); xr.y = fma(e[0], c0r[u].y, xr.y);
            xi.x = fma(e[0], c0i[u].x, xi.x); xi.y = fma(e[0], c0i[u].y, xi.y);
            if (L > 1) {
                xr.x = fma(e[1], c1r[u].x, xr.x); xr.y = fma(e[1], c1r[u].y, xr.y);
                xi.x = fma(e[1], c1i[u].x, xi.x); xi.y = fma(e[1], c1i[u].y, xi.y);
            }
            for (int l = 2; l < L; ++l) {
                const double2 cr = hc[(size_t)l * 2 * HALF + off], ci = hc[(size_t)l * 2 * HALF + HALF + off];
                xr.x = fma(e[l], cr.x, xr.x); xr.y = fma(e[l], cr.y, xr.y);
                xi.x = fma(e[l], ci.x, xi.x); xi.y = fma(e[l], ci.y, xi.y);
            }
            const int i = (2 * off) / NP, j = 2 * off - i * NP;
            const double ar0 = dt * xi.x, ar1 = dt * xi.y, ai0 = -dt * xr.x, ai1 = -dt * xr.y;
            double *p = R + i * LD + j;
            p[0] = ar0; p[1] = ar1;
            p[PL] = ai0; p[PL + 1] = ai1;
            p[2 * PL] = ar0 + ai0; p[2 * PL + 1] = ar1 + ai1;
        }
    }
}

// ||A||_1 of the A in the planes (credited statistics only, cells whose operator-norm bound does not certify the order):
// all threads; returns the norm to everybody
template <int NT>
__device__ __forceinline__ double t18_norm1(double *smem, const int tid) {
    using LY = T18Lds<NT>;
    constexpr int NP = LY::NP, LD = LY::LD, NTH = LY::NTH, PARTS = NTH / NP;
    double *red = smem + LY::RED;
    const int j = tid % NP, part = tid / NP;
    double sum = 0.;
    for (int i = part; i < NP; i += PARTS) {
        const double xr = smem[i * LD + j], xi = smem[LY::PL + i * LD + j];
        sum += fast_sqrt(xr * xr + xi * xi);
    }
    red[tid] = sum;
    __syncthreads();
    if (tid < 64) {
        double c = 0.;
        if (tid < NP)
            for (int p = 0; p < PARTS; ++p) c += red[p * NP + tid];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c = fmax(c, __shfl_xor(c, off, 64));
        if (tid == 0) red[NTH] = c;
    }
    __syncthreads();
    const double nA = red[NTH];
    __syncthreads();   // red is written again inside the cell
    return nA;
}

// matrix instructions per wave of one cell without squarings / of one squaring (executed-work counter)
template <int NT>
struct T18Count {
    static constexpr int NS = NT > 1 ? NT - 1 : 1;
    static constexpr int SQH = NT == 4 ? 3 * 4 * (NT * NS - NT / 2) : 3 * 4 * NT * NS;   // Hermitian square (half of the doubly computed tile)
    static constexpr int HP = 3 * 4 * NT * NS;                                           // Hermitian-result product
    static constexpr int GP = 3 * 4 * NT * NT;                                           // general product
    static constexpr int CELL = 2 * SQH + HP + 2 * GP;   // Hermitian generators
    static constexpr int CELL_GENERAL = 5 * GP;
};

// One cell: A = -i dt H (skew-Hermitian) is in the planes; returns U = exp(A) as a ROTATED column strip (slot s of wave w
// = row tile (w + s) % NT, all NT slots) and the number of squarings that were applied.  On return other waves may
// still be reading the planes.
// SYM: the tile symmetry of Hermitian generators is used (NT >= 3: NT - 1 of NT row tiles of the three powers from the matrix
// instructions); CHEB: Hermitian generators (Chebyshev coefficient set, spectral scaling) -- without SYM for NT <= 2, where no
// wave computes an off-diagonal tile it could mirror.
template <int NT, bool SYM, bool CHEB, class HookFirst, class HookLast>
__device__ __forceinline__ void expm_t18_cell(double *smem, const int wave, const int lane, Strip3M<NT> &U, int &s_out,
                                              bool &bad, HookFirst hook_first, HookLast hook_last) {
    using LY = T18Lds<NT>;
    using CF = T18Coef<CHEB>;
    // Hermitian generators: NT - 1 of NT row tiles of the three powers come from the matrix instructions, the last one is
    // the mirrored tile; general matrices: all NT
    constexpr int LD = LY::LD, NS = SYM ? NT - 1 : NT;
    constexpr bool HALF = SYM && NT == 4;
    static_assert(!SYM || (CHEB && NT >= 3), "tile symmetry needs Hermitian generators and at least three tiles per side");
    double *R = smem, *e1 = smem + LY::E1, *e2 = smem + LY::E2, *red = smem + LY::RED;
    const T18NoHook nohook;
    Strip3M<NT> As, A2, A3, A6;
    t18_load_strip<LD, NT>(R, As, wave, lane);
    T18_STOP_AT(1, U, As);
    // ---- A2 = A A ----
    {
        Acc3<NS> q;
        acc3_zero(q);
        t18_gemm<LD, NS, NT, HALF>(q, R, As, wave, lane, hook_first);
        t18_combine<NS, NT>(q, A2);
    }
    if constexpr (SYM) {
        if constexpr (NT == 4) rot_exch_write<NT, 2>(e1, A2.re[2], A2.im[2], wave, lane, 1.0);   // partial sum of tile (w+2, w) -> wave w+2
        rot_exch_write<NT>(e2, A2.re[1], A2.im[1], wave, lane, 1.0);                             // mirrored tile -> wave w+1
#pragma unroll
        for (int t = 0; t < 2; ++t) A2.sm[t] = A2.re[t] + A2.im[t];
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) A2.sm[t] = A2.re[t] + A2.im[t];
    }
    T18_STOP_AT(2, U, A2);
    STAMP(2);
    // ---- A3 = A A2 (Hermitian generators: the exchanged tiles of A2 arrive while the first k-blocks run) ----
    {
        Acc3<NS> q;
        acc3_zero(q);
        t18_gemm<LD, NS, NT, false>(q, R, A2, wave, lane, [&](int sk, int r) {
            if constexpr (SYM) {
                if (sk == 2 && r == 0) {
                    __syncthreads();
                    if constexpr (NT == 4) { t18_exch_add<NT, 2>(e1, A2, wave, lane); A2.sm[2] = A2.re[2] + A2.im[2]; }
                    t18_exch_read_last<NT>(e2, A2, wave, lane);
                }
            }
        });
        t18_combine<NS, NT>(q, A3);
    }
    T18_STOP_AT(3, U, A3);
    STAMP(3);
    const double n2w = t18_colsum_max<NT>(A2);
    if (lane == 0) red[wave] = n2w;
    if constexpr (!CHEB) {   // general matrices: ||A||_1 and ||A3||_1 as well
        const double n1w = t18_colsum_max<NT>(As), n3w = t18_colsum_max<NT>(A3);
        if (lane == 0) { red[NT + wave] = n3w; red[2 * NT + wave] = n1w; }
    }
    __syncthreads();                                                          // everybody is done reading A
    t18_store_slots<LD, NS, NT>(R, A3, wave, lane);                           // planes = A3 (Hermitian: with the mirrored tiles)
    if constexpr (SYM) t18_store_adjoint<LD, NT>(R, A3.re[1], A3.im[1], wave, lane, -1.0);
#pragma unroll
    for (int t = 0; t < NS; ++t) A3.sm[t] = A3.re[t] + A3.im[t];
    __syncthreads();
    if constexpr (SYM) t18_load_slot_last<LD, NT>(R, A3, wave, lane);
    T18_STOP_AT(4, U, A3);
    STAMP(4);
    // ---- A6 = A3 A3 ----
    {
        Acc3<NS> q;
        acc3_zero(q);
        t18_gemm<LD, NS, NT, HALF>(q, R, A3, wave, lane, nohook);
        t18_combine<NS, NT>(q, A6);
    }
    T18_STOP_AT(5, U, A6);
    STAMP(5);
    if constexpr (SYM) {
        if constexpr (NT == 4) rot_exch_write<NT, 2>(e1, A6.re[2], A6.im[2], wave, lane, 1.0);
        rot_exch_write<NT>(e2, A6.re[1], A6.im[1], wave, lane, 1.0);
        __syncthreads();                                                      // (also: everybody is done reading the planes)
        if constexpr (NT == 4) t18_exch_add<NT, 2>(e1, A6, wave, lane);
        t18_exch_read_last<NT>(e2, A6, wave, lane);
    }
    if constexpr (CHEB) {
        const double n6w = t18_colsum_max<NT>(A6);
        if (lane == 0) red[NT + wave] = n6w;
    }
    // the scaling needs the norms of every wave (and everybody has to be done reading the planes): one barrier
    __syncthreads();
    T18_STOP_AT(6, U, A6);
    STAMP(6);
    int s = 0;
    {
        double n2 = red[0], nq = red[NT], n1 = CHEB ? 0.0 : red[2 * NT];
#pragma unroll
        for (int w = 1; w < NT; ++w) {
            n2 = fmax(n2, red[w]); nq = fmax(nq, red[NT + w]);
            if constexpr (!CHEB) n1 = fmax(n1, red[2 * NT + w]);
        }
        n2 *= 1.0 + 1e-9; nq *= 1.0 + 1e-9;   // (rounding of the computed powers)
        if constexpr (CHEB) {   // beta = min(sqrt ||A2||, ||A6||^(1/6)) <= theta 2^s
            double t2 = CF::THETA * CF::THETA, t6 = t2 * t2 * t2;
            while (!(n2 <= t2 || nq <= t6) && s < 64) { ++s; t2 *= 4.0; t6 *= 64.0; }
        } else {                // alpha = min(||A||, max(||A2||^(1/2), ||A3||^(1/3))) <= theta 2^s
            double t1 = CF::THETA, t2 = t1 * t1, t3 = t2 * t1;
            while (!(n1 <= t1 || (n2 <= t2 && nq <= t3)) && s < 64) { ++s; t1 *= 2.0; t2 *= 4.0; t3 *= 8.0; }
        }
    }
    bad = s >= 64;   // NaN / overflow in the generator
    if (bad) s = 0;
    s = __builtin_amdgcn_readfirstlane(s);
    if (s > 0) {     // A <- A / 2^s: the powers are scaled exactly, the coefficients stay compile-time constants
        const double f1 = ldexp(1.0, -s), f2 = f1 * f1, f3 = f2 * f1, f6 = f3 * f3;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            As.re[t] *= f1; As.im[t] *= f1; A2.re[t] *= f2; A2.im[t] *= f2;
            A3.re[t] *= f3; A3.im[t] *= f3; A6.re[t] *= f6; A6.im[t] *= f6;
        }
    }
    s_out = s;
    // ---- ONE pass over the powers: B1 -> planes, B5 (right operand), B4 (start value of A9), B3 and B2 (kept) ----
    Strip3M<NT> B5;
    Acc3<NT> q;
    Strip<NT> B3, B2;
    const int cdiag = lane & 15, rgd = lane >> 4;   // the diagonal tile is slot 0: row 4r + rg == column c
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        Strip3M<NT> &B1 = U;   // (U is free until the last product: its slots carry B1 to the planes)
        B1.re[t] = CF::A1 * As.re[t] + CF::A2 * A2.re[t] + CF::A3 * A3.re[t];
        B1.im[t] = CF::A1 * As.im[t] + CF::A2 * A2.im[t] + CF::A3 * A3.im[t];
        B5.re[t] = CF::E2 * A2.re[t] + CF::E3 * A3.re[t] + CF::E6 * A6.re[t];
        B5.im[t] = CF::E2 * A2.im[t] + CF::E3 * A3.im[t] + CF::E6 * A6.im[t];
        B5.sm[t] = B5.re[t] + B5.im[t];
        d4 b4r = CF::D1 * As.re[t] + CF::D2 * A2.re[t] + CF::D3 * A3.re[t] + CF::D6 * A6.re[t];
        const d4 b4i = CF::D1 * As.im[t] + CF::D2 * A2.im[t] + CF::D3 * A3.im[t] + CF::D6 * A6.im[t];
        B3.re[t] = CF::C1 * As.re[t] + CF::C2 * A2.re[t] + CF::C3 * A3.re[t] + CF::C6 * A6.re[t];
        B3.im[t] = CF::C1 * As.im[t] + CF::C2 * A2.im[t] + CF::C3 * A3.im[t] + CF::C6 * A6.im[t];
        B2.re[t] = CF::B1 * As.re[t] + CF::B2 * A2.re[t] + CF::B3 * A3.re[t] + CF::B6 * A6.re[t];
        B2.im[t] = CF::B1 * As.im[t] + CF::B2 * A2.im[t] + CF::B3 * A3.im[t] + CF::B6 * A6.im[t];
        if (t == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * r + rgd == cdiag) { b4r[r] += CF::D0; B3.re[0][r] += CF::C0; B2.re[0][r] += CF::B0; }   // (d0 = 0, b0 = 1: see grape_t18_coeffs.h)
        }
        // A9 = B1 B5 + B4 through the start values: re = p1 - p2, im = p3 - p1 - p2
        q.p1[t] = b4r; q.p2[t] = (d4){0., 0., 0., 0.}; q.p3[t] = b4r + b4i;
    }
    t18_store_slots<LD, NT, NT>(R, U, wave, lane);                            // planes = B1 (free since the last barrier)
    __syncthreads();
    T18_STOP_AT(7, U, B5);
    STAMP(7);
    // ---- A9 = B1 B5 + B4 ----
    t18_gemm<LD, NT, NT, false>(q, R, B5, wave, lane, nohook);
    Strip3M<NT> A9;
    t18_combine<NT, NT>(q, A9);
#pragma unroll
    for (int t = 0; t < NT; ++t) A9.sm[t] = A9.re[t] + A9.im[t];
    T18_STOP_AT(8, U, A9);
    STAMP(8);
    // ---- p = B2 + (B3 + A9) A9 ----
    __syncthreads();                                                          // everybody is done reading B1
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        U.re[t] = A9.re[t] + B3.re[t];
        U.im[t] = A9.im[t] + B3.im[t];
        q.p1[t] = B2.re[t]; q.p2[t] = (d4){0., 0., 0., 0.}; q.p3[t] = B2.re[t] + B2.im[t];
    }
    t18_store_slots<LD, NT, NT>(R, U, wave, lane);                            // planes = B3 + A9
    __syncthreads();
    T18_STOP_AT(9, U, A9);
    STAMP(9);
    t18_gemm<LD, NT, NT, false>(q, R, A9, wave, lane, hook_last);
    t18_combine<NT, NT>(q, U);
    STAMP(10);
    // ---- squarings ----
    for (int it = 0; it < s; ++it) {
        __syncthreads();
        t18_store_slots<LD, NT, NT>(R, U, wave, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) U.sm[t] = U.re[t] + U.im[t];
        __syncthreads();
        acc3_zero(q);
        t18_gemm<LD, NT, NT, false>(q, R, U, wave, lane, nohook);
        t18_combine<NT, NT>(q, U);
    }
}

// ---- four products instead of five: degree 16, Hermitian generators with rho(H dt) <= T16_THETA, no scaling ----
//     A2 = A A                                                   (Hermitian square, as above)
//     y0 = (c1 A2 + c2 A) A2                                     (general products from here on: the operands mix the
//     y1 = (y0 + c3 A2 + c4 A)(y0 + c5 A2) + c6 y0 + c7 A2        Hermitian and the skew-Hermitian powers)
//     p  = (y1 + c8 A2 + c9 A)(y1 + c10 y0 + c11 A) + c12 y1 + c13 y0 + c14 A2 + c15 A + c16 I
// (grape_t18_coeffs.h).  3.6 instead of 4.25 general-product equivalents per cell, one exchange of mirrored tiles instead
// of three, and the rounding error of the order-13 Pade approximant (1.2e-16 max element at N = 64, rho = 1; the degree-18
// coefficient set gives 1.1e-15: its constant terms cancel from -4.3 to 1).
// The form is evaluated AS WRITTEN: folding c3 A2 + c4 A into the start value of the second product (and likewise for
// the third) would make each left operand the previous result and spare two combination passes, but the last
// combination then cancels 25 A against 24 A -- 1.4e-15, measured (tools/t16_rounding.py).
// The spectral bound comes from y0 itself: with spec(H dt) = {lam}, y0 = c1 H^4 + i c2 H^3 (dt absorbed), so
//     ||y0||_F^2 = c1^2 m8 + c2^2 m6,   Re <A2, y0>_F = -c1 m6,   m_p = sum lam^p >= rho^p
// -- two sums over the register strips give m8 = sum lam^8 without any further product, and rho <= m8^(1/8)
// (1.17-1.24 rho at N = 64 for a semicircle spectrum, where sqrt ||A2||_1 is 1.8 rho).  The cell is evaluated whatever the
// bound says and the verdict returned: a cell with m8 > theta^8 (and ||A2||_1 > theta^2) has to be redone by the degree-18
// route -- in ANOTHER launch.  With the fallback inside this cell (an early return and the other cell function behind it)
// the register allocation of both routes fell apart: 780 bytes of scratch per lane and 24.7 instead of 16.5 ms at the
// headline configuration, where no cell needs it.
template <int NT>
struct T16Count {
    static constexpr int CELL = T18Count<NT>::SQH + 3 * T18Count<NT>::GP;    // matrix instructions per wave (tile symmetry)
    static constexpr int CELL_NOSYM = 4 * T18Count<NT>::GP;
};

// SYM as in expm_t18_cell: the tile symmetry of the Hermitian square (NT >= 3); without it all NT row tiles of A2 come
// from the matrix instructions (NT <= 2, where no wave computes an off-diagonal tile it could mirror).
template <int NT, bool SYM, class HookFirst, class HookLast>
__device__ __forceinline__ bool expm_t16_cell(double *smem, const int wave, const int lane_in, Strip3M<NT> &U, HookFirst hook_first,
                                              HookLast hook_last) {
    // (per-lane addresses are recomputed per stage from a pinned copy of the lane index: shared between the stages they
    // stay alive through the whole cell -- some 50 registers that this cell does not have)
#define T16_FRESH_LANE(l) int l = lane_in; asm volatile("" : "+v"(l))
    using LY = T18Lds<NT>;
    constexpr int LD = LY::LD, NS = SYM ? NT - 1 : NT;
    constexpr bool HALF = SYM && NT == 4;
    static_assert(!SYM || NT >= 3, "tile symmetry needs at least three tiles per side");
    double *R = smem, *e1 = smem + LY::E1, *e2 = smem + LY::E2, *red = smem + LY::RED;
    const T18NoHook nohook;
    Strip3M<NT> As, A2, Y0;
    int lane = lane_in;
    t18_load_strip<LD, NT>(R, As, wave, lane);
    // ---- A2 = A A ----
    {
        Acc3<NS> q;
        acc3_zero(q);
        t18_gemm<LD, NS, NT, HALF>(q, R, As, wave, lane, hook_first);
        t18_combine<NS, NT>(q, A2);
    }
    if constexpr (SYM) {
        if constexpr (NT == 4) rot_exch_write<NT, 2>(e1, A2.re[2], A2.im[2], wave, lane, 1.0);   // partial sum of tile (w+2, w) -> wave w+2
        rot_exch_write<NT>(e2, A2.re[1], A2.im[1], wave, lane, 1.0);                             // mirrored tile -> wave w+1
    }
    __syncthreads();                                                          // (also: everybody is done reading A)
    if constexpr (SYM) {
        if constexpr (NT == 4) t18_exch_add<NT, 2>(e1, A2, wave, lane);
        t18_exch_read_last<NT>(e2, A2, wave, lane);
    }
    Acc3<NT> q;
    { T16_FRESH_LANE(l2); lane = l2; }
    // ---- y0 = (c1 A2 + c2 A) A2: the combination goes to the planes, A2 -- alive to the end anyway -- is the right operand ----
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        U.re[t] = T16_C1 * A2.re[t] + T16_C2 * As.re[t];
        U.im[t] = T16_C1 * A2.im[t] + T16_C2 * As.im[t];
        A2.sm[t] = A2.re[t] + A2.im[t];
    }
    t18_store_slots<LD, NT, NT>(R, U, wave, lane);
    __syncthreads();
    acc3_zero(q);
    t18_gemm<LD, NT, NT, false>(q, R, A2, wave, lane, nohook);
    t18_combine<NT, NT>(q, Y0);
    // ---- spectral bound ----
    {
        double f = 0., g = 0.;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f = fma(Y0.re[t][r], Y0.re[t][r], f); f = fma(Y0.im[t][r], Y0.im[t][r], f);
                g = fma(A2.re[t][r], Y0.re[t][r], g); g = fma(A2.im[t][r], Y0.im[t][r], g);
            }
        f = wave_sum(f);
        g = wave_sum(g);
        const double n2w = t18_colsum_max<NT>(A2);
        if (lane == 0) { red[wave] = f; red[NT + wave] = g; red[2 * NT + wave] = n2w; }
    }
    __syncthreads();                                                          // (also: everybody is done reading the planes)
    bool ok;
    {
        double F = red[0], G = red[NT], n2 = red[2 * NT];
#pragma unroll
        for (int w = 1; w < NT; ++w) { F += red[w]; G += red[NT + w]; n2 = fmax(n2, red[2 * NT + w]); }
        const double m6 = -G / T16_C1;                                        // sum lam^6
        const double m8 = (F - (T16_C2 * T16_C2) * m6) / (T16_C1 * T16_C1);   // sum lam^8 (cancellation: 55 ulp of F, far inside the 1e-9)
        constexpr double th2 = T16_THETA * T16_THETA, th8 = th2 * th2 * th2 * th2;
        ok = (n2 * (1.0 + 1e-9) <= th2) || (m6 >= 0. && m8 * (1.0 + 1e-9) <= th8);   // (NaN: neither)
    }
    ok = __builtin_amdgcn_readfirstlane((int)ok) != 0;
    { T16_FRESH_LANE(l3); lane = l3; }
    // ---- y1 = (y0 + c3 A2 + c4 A)(y0 + c5 A2) + c6 y0 + c7 A2 ----
    // (the right operand takes the place of y0, which comes back as operand - c5 A2 behind the product: with y0 AND the
    // operand alive the product runs 32 doubles per lane above what the register file holds beside A, A2 and the accumulators)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        U.re[t] = Y0.re[t] + T16_C3 * A2.re[t] + T16_C4 * As.re[t];
        U.im[t] = Y0.im[t] + T16_C3 * A2.im[t] + T16_C4 * As.im[t];
        const d4 vr = T16_C6 * Y0.re[t] + T16_C7 * A2.re[t], vi = T16_C6 * Y0.im[t] + T16_C7 * A2.im[t];
        q.p1[t] = vr; q.p2[t] = (d4){0., 0., 0., 0.}; q.p3[t] = vr + vi;       // re = p1 - p2, im = p3 - p1 - p2
        Y0.re[t] += T16_C5 * A2.re[t];
        Y0.im[t] += T16_C5 * A2.im[t];
        Y0.sm[t] = Y0.re[t] + Y0.im[t];
    }
    t18_store_slots<LD, NT, NT>(R, U, wave, lane);                            // planes = y0 + c3 A2 + c4 A
    __syncthreads();
    t18_gemm<LD, NT, NT, false>(q, R, Y0, wave, lane, nohook);
    Strip3M<NT> &Y1 = U;
    t18_combine<NT, NT>(q, Y1);
    // ---- p = (y1 + c8 A2 + c9 A)(y1 + c10 y0 + c11 A) + c12 y1 + c13 y0 + c14 A2 + c15 A + c16 I ----
    __syncthreads();                                                          // everybody is done reading the planes
    { T16_FRESH_LANE(l4); lane = l4; }
    {
        const int cdiag = lane & 15, rgd = lane >> 4;   // the diagonal tile is slot 0: row 4r + rg == column c
        constexpr int PL = LY::PL;
        double *x = R + (lane >> 4) * LD + 16 * wave + (lane & 15);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const d4 y0r = Y0.re[t] - T16_C5 * A2.re[t], y0i = Y0.im[t] - T16_C5 * A2.im[t];
            const d4 xr = Y1.re[t] + T16_C8 * A2.re[t] + T16_C9 * As.re[t], xi = Y1.im[t] + T16_C8 * A2.im[t] + T16_C9 * As.im[t];
            const int tb = (wave + t) % NT;
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                     // planes = y1 + c8 A2 + c9 A, tile by tile
                const int o = (16 * tb + 4 * r) * LD;
                x[o] = xr[r]; x[PL + o] = xi[r]; x[2 * PL + o] = xr[r] + xi[r];
            }
            d4 vr = T16_C12 * Y1.re[t] + T16_C13 * y0r + T16_C14 * A2.re[t] + T16_C15 * As.re[t];
            const d4 vi = T16_C12 * Y1.im[t] + T16_C13 * y0i + T16_C14 * A2.im[t] + T16_C15 * As.im[t];
            if (t == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * r + rgd == cdiag) vr[r] += T16_C16;
            }
            q.p1[t] = vr; q.p2[t] = (d4){0., 0., 0., 0.}; q.p3[t] = vr + vi;
            Y0.re[t] = Y1.re[t] + T16_C10 * y0r + T16_C11 * As.re[t];         // right operand in place
            Y0.im[t] = Y1.im[t] + T16_C10 * y0i + T16_C11 * As.im[t];
            Y0.sm[t] = Y0.re[t] + Y0.im[t];
        }
    }
    __syncthreads();
    t18_gemm<LD, NT, NT, false>(q, R, Y0, wave, lane, hook_last);
    t18_combine<NT, NT>(q, U);
    return ok;
#undef T16_FRESH_LANE
}

// store of a rotated strip as U_kn (row-major interleaved complex)
template <int NT>
__device__ __forceinline__ void t18_store_u(const ExpmArgs &a, const int cell, const int wave, const int lane, const Strip3M<NT> &T) {
    constexpr int NP = 16 * NT;
    double2 *Uc = a.U + (size_t)cell * NP * NP;
    const int col = 16 * wave + (lane & 15), rg = lane >> 4;
#pragma unroll
    for (int sl = 0; sl < NT; ++sl) {
        const int tb = (wave + sl) % NT;
#pragma unroll
        for (int r = 0; r < 4; ++r) Uc[(16 * tb + 4 * r + rg) * NP + col] = make_double2(T.re[sl][r], T.im[sl][r]);
    }
}

// one slot (row tile) of the above: four 16-byte stores per lane in front of a k-block of the NEXT cell's first product
template <int NT>
__device__ __forceinline__ void t18_store_u_slot(const ExpmArgs &a, const int cell, const int wave, const int lane, const Strip<NT> &T,
                                                 const int sl) {
    constexpr int NP = 16 * NT;
    // wave-uniform part of the address (cell, row tile, register) in scalar registers, one per-lane offset for all stores:
    // address arithmetic on the vector ALU would sit between the matrix instructions of the product these stores ride in
    const int tb = (wave + sl) % NT;
    double2 *Ut = a.U + ((size_t)cell * NP + 16 * tb) * NP;
    const unsigned loff = (unsigned)((lane >> 4) * NP + 16 * wave + (lane & 15));   // (unsigned: scalar base + 32-bit lane offset)
#pragma unroll
    for (int r = 0; r < 4; ++r) (Ut + (size_t)(4 * r) * NP)[loff] = make_double2(T.re[sl][r], T.im[sl][r]);
}

// Persistent launch: one workgroup per CU walks a round-robin share of the cells of its XCD (neighbouring cells of the
// same trajectories run concurrently on one XCD: H0_k stays in that XCD's L2).  The credited statistics (Pade order and
// squarings Julia's exp! would use) come from the 1-norm bound of the operators or, outside its certifying
// window, from the measured norm, exactly as in expm_persistent; the executed work is counted separately.
// T16 (Hermitian generators): every cell takes the four-product route; those whose spectral bound is beyond T16_THETA
// are appended to a.cell_list (flags[4] counts them, flags[5] the cells tried) and redone by a launch of the five-product
// variant with a.listed set, which the host issues behind this one (no cell listed: a launch that ends at once).  From the
// two counts the host decides whether the next evaluation tries the four-product route again.
// Three tiles per side, four-product route: 256 registers and 72 KB, TWO workgroups per CU -- three waves leave a SIMD idle
// and a lone wave cannot hide its own vector and LDS phases; with six waves on four SIMDs two of them are matrix-bound.
template <int NT, bool SYM, bool CHEB, bool T16 = false>
__global__ void __launch_bounds__(NT * 64, (T16 && NT == 3) ? 2 : 1) expm_t18_kernel(ExpmArgs a) {
    static_assert(!T16 || CHEB, "the four-product route is a Hermitian-generator route");
    using LY = T18Lds<NT>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid0 = threadIdx.x, lane0 = tid0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int ncell = a.K * a.N_T;
    const int x = blockIdx.x & 7, per_x = gridDim.x >> 3;
    const int lo = (int)((long)x * ncell / 8), hi = (int)((long)(x + 1) * ncell / 8);
    // the cells of this workgroup: positions idx0, idx0 + step, ... < end of its XCD's range of cells or, for a launch
    // that works through the hand-over list of the four-product variant, of that list
    // (the plan of this evaluation, t16_plan_kernel: with more than a quarter of the cells predicted beyond the range of the
    // four-product route that route is not tried -- its kernel leaves, the launch behind it walks all cells, not a list)
    const bool skip16 = a.cell_list != nullptr && t16_skipped(a.flags, ncell);
    if (T16 && skip16) return;
    const bool listed = a.listed != 0 && !skip16;
    const int step = listed ? (int)gridDim.x : per_x;
    const int end = listed ? min(__builtin_amdgcn_readfirstlane(a.flags[4]), ncell) : hi;
    const int idx0 = listed ? (int)blockIdx.x : lo + ((int)blockIdx.x >> 3);
    auto cell_at = [&](int i) -> int { return listed ? __builtin_amdgcn_readfirstlane(a.cell_list[i]) : i; };
    int st_s = 0, st_max = 0, st_ord[5] = {0, 0, 0, 0, 0}, st_sq = 0, st_cells = 0, st_t16 = 0;
    bool any_bad = false;
    // Software pipeline over the cells of this workgroup: the result of cell c is stored between the matrix instructions
    // of the first product of cell c+1, the operator tiles of cell c+1 are requested in front of the last product of
    // cell c (lowest register pressure of the cell) and combined into the planes behind it.
    Strip<NT> Uprev;
    strip_zero(Uprev);
    int prev = -1;
    const int first = idx0 < end ? cell_at(idx0) : 0;
    if (idx0 < end) {
        if constexpr (SYM) {
            T18FormA<64 * NT, NT> fa(a, smem, first, tid0);
            fa.issue();
            fa.commit();
        } else {
            t18_form_a_general<64 * NT, NT>(a, smem, first, tid0);
        }
    }
    __syncthreads();
    for (int idx = idx0; idx < end; idx += step) {
        const int cell = cell_at(idx);
        int lane = lane0, tid = tid0;
        asm volatile("" : "+v"(lane), "+v"(tid));   // per-lane addresses are recomputed per cell (see expm_persistent)
#ifdef GRAPE_DIAG
        if (tid == 0) g_diag_off[blockIdx.x & 1023] = (idx != idx0 + step);
        __syncthreads();
#endif
        STAMP(0);
        // credited work: what Julia's exp! would do for this cell (order and squarings from ||A||_1)
        int sj = 0, oj = 0;
        if (!listed) {   // (a listed cell was credited by the launch that listed it)
            const double bound = expm_norm_bound(a, cell);
            double nA = bound;
            if (!(bound > 2.1 && bound <= 5.4)) nA = t18_norm1<NT>(smem, tid);
            if (nA > 5.4) {
                const double r = nA / 5.4;
                const int e = ilogb(r);
                sj = (r == ldexp(1.0, e)) ? e : e + 1;
            }
            oj = nA > 2.1 ? 4 : nA > 0.95 ? 3 : nA > 0.25 ? 2 : nA > 0.015 ? 1 : 0;
        }
        Strip3M<NT> U;
        int s;
        bool bad;
        STAMP(1);
        const bool have_next = idx + step < end;
        const int next = have_next ? cell_at(idx + step) : cell;
        if constexpr (SYM) {
            T18FormA<64 * NT, NT> fa(a, smem, next, tid);   // (no next cell: the same tiles again, not committed)
            auto store_prev = [&](int sk, int r) { if (r == 0 && prev >= 0) t18_store_u_slot<NT>(a, prev, wave, lane, Uprev, sk); };
            auto fetch_next = [&](int sk, int r) { if (sk == 1 && r == 0) fa.issue(); };   // (fenced by scheduling barriers on both sides)
            if constexpr (T16) {
                s = 0; bad = false;
                const bool ok16 = expm_t16_cell<NT, true>(smem, wave, lane, U, store_prev, fetch_next);
                if (tid == 0 && a.cellflag) a.cellflag[cell] = ok16 ? 0 : 1;   // the verdict (round 6: deriv_econ_kernel reads it)
                if (ok16) st_t16 += 1;
                else if (tid == 0) a.cell_list[atomicAdd(&a.flags[4], 1)] = cell;   // to be redone by the five-product launch
            } else {
                expm_t18_cell<NT, true, true>(smem, wave, lane, U, s, bad, store_prev, fetch_next);
            }
            STAMP(11);
            __syncthreads();   // everybody is done reading the planes
            STAMP(12);
            if (have_next) fa.commit();
        } else {
            // (general matrices: the result is stored at once -- carried into the next cell's first product it costs 64
            // registers there, and this variant runs at the register limit: 84 bytes of scratch per lane with it)
            if constexpr (T16) {
                s = 0; bad = false;
                const bool ok16 = expm_t16_cell<NT, false>(smem, wave, lane, U, T18NoHook(), T18NoHook());
                if (tid == 0 && a.cellflag) a.cellflag[cell] = ok16 ? 0 : 1;   // the verdict (deriv_econ_kernel)
                if (ok16) st_t16 += 1;
                else if (tid == 0) a.cell_list[atomicAdd(&a.flags[4], 1)] = cell;
            } else {
                expm_t18_cell<NT, false, CHEB>(smem, wave, lane, U, s, bad, T18NoHook(), T18NoHook());
            }
#pragma unroll
            for (int sl = 0; sl < NT; ++sl) {
                Strip<NT> Us;
                Us.re[sl] = U.re[sl]; Us.im[sl] = U.im[sl];
                t18_store_u_slot<NT>(a, cell, wave, lane, Us, sl);
            }
            __syncthreads();
            if (have_next) t18_form_a_general<64 * NT, NT>(a, smem, next, tid);   // (all elements: not fetched ahead)
        }
        if constexpr (SYM) {
#pragma unroll
            for (int t = 0; t < NT; ++t) { Uprev.re[t] = U.re[t]; Uprev.im[t] = U.im[t]; }
            prev = cell;
        }
        any_bad |= bad;
        st_s += sj; st_max = max(st_max, sj); st_ord[oj] += listed ? 0 : 1;
        st_sq += s; st_cells += 1;
        __syncthreads();
        STAMP(13);
    }
    if (prev >= 0) {
#pragma unroll
        for (int sl = 0; sl < NT; ++sl) t18_store_u_slot<NT>(a, prev, wave, lane0, Uprev, sl);
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
        // executed matrix instructions (all waves), squarings and cells of this path (four-product variant: every cell is
        // executed, the cells handed over are counted by the launch that redoes them)
        unsigned long long mi = (unsigned long long)st_cells * (T16 ? (SYM ? T16Count<NT>::CELL : T16Count<NT>::CELL_NOSYM) : SYM ? T18Count<NT>::CELL : T18Count<NT>::CELL_GENERAL)
                                + (unsigned long long)st_sq * T18Count<NT>::GP;
        if constexpr (T16) {
            stat_add(a.stats, 15, (unsigned long long)st_t16);
            atomicAdd(&a.flags[5], st_cells);
        }
        stat_add(a.stats, 12, (unsigned long long)NT * mi);
        stat_add(a.stats, 13, (unsigned long long)st_sq);
        stat_add(a.stats, 14, (unsigned long long)(T16 ? st_t16 : st_cells));
        if (any_bad) atomicOr(&a.flags[0], 64);
    }
}
