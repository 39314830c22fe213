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
//     B3 = c0 I + c1 A + c2 A2 + c3 A3 + c6 A6,  B2 = b1 A + b2 A2 + b3 A3 + b6 A6
//     p(A) = B2 + (B3 + A9) A9                          (general product)
// 768 matrix instructions per wave at N = 64 against 816 + 264 (products + solve) of the Pade route, and none of them
// waits for a tile inversion.
// The spectral bound is rigorous and costs two column-sum passes over register strips: rho(H dt)^2 <= ||A2||_1 and
// rho(H dt)^6 <= ||A6||_1 (Hermitian matrices: ||M||_2 <= ||M||_1; |re| + |im| stands in for the modulus, which can only
// enlarge the bound).  beta = min(sqrt ||A2||_1, ||A6||_1^(1/6)); beta > 2: A is scaled by 2^-s (the powers by 2^-ks,
// exact) and the result squared s times.  At the headline configuration ||A||_1 = 4.2 but beta = 1.17: no squaring.
#pragma once
#include "grape_kernels.hip.h"

#define T18_THETA 2.0
// a1, a2, a3, b1, b2, b3, b6, c0, c1, c2, c3, c6, d0, d1, d2, d3, d6, e2, e3, e6  (tools/t18_coeffs.py 2.0)
#define T18_A1 -0.10036558103014462001
#define T18_A2 -0.007456351650625886579
#define T18_A3 -0.00083091953191006175088
#define T18_B1 0.24166417193309948294
#define T18_B2 1.1119704726210786376
#define T18_B3 0.29736195952844853785
#define T18_B6 -0.000564510422238531483
#define T18_C0 -4.2636626654470864734
#define T18_C1 1.7157463766850012865
#define T18_C2 0.073686948027488562391
#define T18_C3 -0.0033650385206633560936
#define T18_C6 0.000033927981037541774044
#define T18_D0 -0.22288835997489735785
#define T18_D1 -0.24222749901747747758
#define T18_D2 0.050668391204088569683
#define T18_D3 0.023404567895744140748
#define T18_D6 -0.000010355013205937047443
#define T18_E2 -0.13912895765004587534
#define T18_E3 -0.013910627366173824328
#define T18_E6 -0.000014649629174709440602

// max_j sum_i (|re| + |im|) over the wave's 16 columns of a complete rotated strip (all NT slots): an upper bound of the
// largest column sum of moduli of this column strip; uniform over the wave
template <int NT>
__device__ __forceinline__ double t18_colsum_max(const Strip<NT> &S) {
    double c = 0.;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) c += fabs(S.re[t][r]) + fabs(S.im[t][r]);
    c += __shfl_xor(c, 16, 64);   // the four lane rows hold the same column
    c += __shfl_xor(c, 32, 64);
    c = fmax(c, dpp_f64<DPP_QUAD_XOR1>(c));
    c = fmax(c, dpp_f64<DPP_QUAD_XOR2>(c));
    c = fmax(c, dpp_f64<DPP_ROW_HALF_MIRROR>(c));
    c = fmax(c, dpp_f64<DPP_ROW_MIRROR>(c));
    return readlane_f64(c, 0);
}

struct T18NoHook {
    __device__ __forceinline__ void operator()(int) const {}
};

// acc[0..NS-1] += X * B for rotated strips, X in LDS planes (natural layout), 3M scheme and software pipeline exactly as
// gemm_rot; the right operand comes from a functor bop(sk, r, bre, bim) (slot sk, register r: k-step 16 ((w + sk) % NT) + 4 r)
// so that a linear combination of strips is formed on the fly, and hook(sk) runs in front of k-block sk (late arrival
// of a mirrored tile, staged prefetch of the next cell).
template <int LD, int NS, int NT, bool HALF_LAST, class BOp, class Hook>
__device__ __forceinline__ void t18_gemm(Strip<NT> &acc, const double *__restrict__ Xre, const double *__restrict__ Xim,
                                         int wave, int lane, BOp bop, Hook hook) {
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
        hook(sk);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kn = (r < 3) ? 16 * ((wave + sk) % NT) + 4 * (r + 1) : 16 * ((wave + sk + 1) % NT);   // next k column
            const bool more = !(sk == NT - 1 && r == 3);
            double bre, bim;
            bop(sk, r, bre, bim);
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

// matrix instructions per wave of one cell without squarings / of one squaring (executed-work counter)
template <int NT>
struct T18Count {
    static constexpr int NS = NT - 1;
    static constexpr int SQH = NT == 4 ? 3 * 4 * (NT * NS - NT / 2) : 3 * 4 * NT * NS;   // Hermitian square (half of the doubly computed tile)
    static constexpr int HP = 3 * 4 * NT * NS;                                           // Hermitian-result product
    static constexpr int GP = 3 * 4 * NT * NT;                                           // general product
    static constexpr int CELL = 2 * SQH + HP + 2 * GP;
};

// One cell: A = -i dt H (skew-Hermitian) is in the A region of the LDS; returns U = exp(A) as a ROTATED column strip
// (slot s of wave w = row tile (w + s) % NT, all NT slots) and the number of squarings that were applied.
template <int NT, class Hook = T18NoHook>
__device__ __forceinline__ void expm_t18_cell(double *smem, const int wave, const int lane, Strip<NT> &T, int &s_out,
                                              bool &bad, Hook hook = Hook()) {
    using LY = ExpmLds<NT>;
    constexpr int NP = LY::NP, LD = LY::LD, NS = NT - 1;
    double *Are = smem, *Aim = Are + NP * LD, *Xre = smem + LY::REG, *Xim = Xre + NP * LD;
    double *exch = smem + 2 * LY::REG, *red = exch + LY::DV;
    Strip<NT> As, A2, A3, A6;
    rot_load_strip<LD, NT>(Are, Aim, As, wave, lane);
    // ---- A2 = A A (Hermitian) ----
    strip_zero(A2);
    t18_gemm<LD, NS, NT, NT == 4>(A2, Are, Aim, wave, lane,
        [&](int sk, int r, double &br, double &bi) { br = As.re[sk][r]; bi = As.im[sk][r]; }, T18NoHook());
    if constexpr (NT == 4) {   // slot 2 += (partial sum of wave w+2)^dagger, through the idle X region
        rot_exch_write<NT, 2>(Xre, A2.re[2], A2.im[2], wave, lane, 1.0);
        __syncthreads();
        rot_exch_add<NT, 2>(Xre, A2, wave, lane);
    }
    // the mirrored tile of A2 travels while the first NT - 1 k-blocks of the next product run
    rot_exch_write<NT>(exch, A2.re[1], A2.im[1], wave, lane, 1.0);
    // ---- A3 = A A2 (skew-Hermitian) ----
    strip_zero(A3);
    t18_gemm<LD, NS, NT, false>(A3, Are, Aim, wave, lane,
        [&](int sk, int r, double &br, double &bi) { br = A2.re[sk][r]; bi = A2.im[sk][r]; },
        [&](int sk) { if (sk == NT - 1) { __syncthreads(); rot_exch_read<NT>(exch, A2, wave, lane); } });
    // (every wave is past the barrier inside the product: the exchange tiles in the X region have been consumed)
    rot_store_slots<LD, NS, NT>(Xre, Xim, A3, wave, lane);                    // X = A3, with the mirrored tiles
    rot_store_adjoint<LD, NT>(Xre, Xim, A3.re[1], A3.im[1], wave, lane, -1.0);
    const double n2w = t18_colsum_max<NT>(A2);
    __syncthreads();                                                          // A is dead from here on
    rot_load_slot3<LD, NT>(Xre, Xim, A3, wave, lane);
    // ---- A6 = A3 A3 (Hermitian) ----
    strip_zero(A6);
    t18_gemm<LD, NS, NT, NT == 4>(A6, Xre, Xim, wave, lane,
        [&](int sk, int r, double &br, double &bi) { br = A3.re[sk][r]; bi = A3.im[sk][r]; }, hook);
    if constexpr (NT == 4) {
        rot_exch_write<NT, 2>(exch, A6.re[2], A6.im[2], wave, lane, 1.0);
        __syncthreads();
        rot_exch_add<NT, 2>(exch, A6, wave, lane);
        __syncthreads();
    }
    rot_exch_write<NT>(exch, A6.re[1], A6.im[1], wave, lane, 1.0);
    if (lane == 0) red[wave] = n2w;
    __syncthreads();                                                          // (also: everybody is done reading X = A3)
    rot_exch_read<NT>(exch, A6, wave, lane);
    const double n6w = t18_colsum_max<NT>(A6);
    if (lane == 0) red[NT + wave] = n6w;
    // B1 needs the scaling, the scaling needs the norm of A6 of every wave: one more barrier
    __syncthreads();
    double n2 = red[0], n6 = red[NT];
#pragma unroll
    for (int w = 1; w < NT; ++w) { n2 = fmax(n2, red[w]); n6 = fmax(n6, red[NT + w]); }
    n2 *= 1.0 + 1e-9; n6 *= 1.0 + 1e-9;   // (rounding of the computed powers)
    int s = 0;
    {
        double t2 = T18_THETA * T18_THETA, t6 = t2 * t2 * t2;
        while (!(n2 <= t2 || n6 <= t6) && s < 64) { ++s; t2 *= 4.0; t6 *= 64.0; }
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
    // ---- B1 -> X region ----
    {
        Strip<NT> B;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            B.re[t] = T18_A1 * As.re[t] + T18_A2 * A2.re[t] + T18_A3 * A3.re[t];
            B.im[t] = T18_A1 * As.im[t] + T18_A2 * A2.im[t] + T18_A3 * A3.im[t];
        }
        rot_store_slots<LD, NT, NT>(Xre, Xim, B, wave, lane);
    }
    // ---- A9 = B1 B5 + B4 ----  (B4 is added behind the product: as a start value it would occupy 64 more registers
    // while the four strips, the three partial-product sets and the operands are live)
    Strip<NT> A9;
    strip_zero(A9);
    const int cdiag = lane & 15, rgd = lane >> 4;   // the diagonal tile is slot 0: row 4r + rg == column c
    __syncthreads();
    t18_gemm<LD, NT, NT, false>(A9, Xre, Xim, wave, lane,
        [&](int sk, int r, double &br, double &bi) {
            br = T18_E2 * A2.re[sk][r] + T18_E3 * A3.re[sk][r] + T18_E6 * A6.re[sk][r];
            bi = T18_E2 * A2.im[sk][r] + T18_E3 * A3.im[sk][r] + T18_E6 * A6.im[sk][r];
        }, hook);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        A9.re[t] += T18_D1 * As.re[t] + T18_D2 * A2.re[t] + T18_D3 * A3.re[t] + T18_D6 * A6.re[t];
        A9.im[t] += T18_D1 * As.im[t] + T18_D2 * A2.im[t] + T18_D3 * A3.im[t] + T18_D6 * A6.im[t];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (4 * r + rgd == cdiag) A9.re[0][r] += T18_D0;
    // ---- p = B2 + (B3 + A9) A9 ----
    __syncthreads();                                                          // everybody is done reading X = B1
    {
        Strip<NT> Lm;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            Lm.re[t] = A9.re[t] + T18_C1 * As.re[t] + T18_C2 * A2.re[t] + T18_C3 * A3.re[t] + T18_C6 * A6.re[t];
            Lm.im[t] = A9.im[t] + T18_C1 * As.im[t] + T18_C2 * A2.im[t] + T18_C3 * A3.im[t] + T18_C6 * A6.im[t];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * r + rgd == cdiag) Lm.re[0][r] += T18_C0;
        rot_store_slots<LD, NT, NT>(Xre, Xim, Lm, wave, lane);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        T.re[t] = T18_B1 * As.re[t] + T18_B2 * A2.re[t] + T18_B3 * A3.re[t] + T18_B6 * A6.re[t];
        T.im[t] = T18_B1 * As.im[t] + T18_B2 * A2.im[t] + T18_B3 * A3.im[t] + T18_B6 * A6.im[t];
    }
    __syncthreads();
    t18_gemm<LD, NT, NT, false>(T, Xre, Xim, wave, lane,
        [&](int sk, int r, double &br, double &bi) { br = A9.re[sk][r]; bi = A9.im[sk][r]; }, hook);
    // ---- squarings ----
    for (int it = 0; it < s; ++it) {
        __syncthreads();
        rot_store_slots<LD, NT, NT>(Xre, Xim, T, wave, lane);
        __syncthreads();
        Strip<NT> Sq;
        strip_zero(Sq);
        t18_gemm<LD, NT, NT, false>(Sq, Xre, Xim, wave, lane,
            [&](int sk, int r, double &br, double &bi) { br = T.re[sk][r]; bi = T.im[sk][r]; }, T18NoHook());
        T = Sq;
    }
}

// store of a rotated strip as U_kn (row-major interleaved complex)
template <int NT>
__device__ __forceinline__ void t18_store_u(const ExpmArgs &a, const int cell, const int wave, const int lane, const Strip<NT> &T) {
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

// Persistent launch: one workgroup per CU walks a round-robin share of the cells of its XCD (neighbouring cells of the
// same trajectories run concurrently on one XCD: H0_k stays in that XCD's L2).  The credited statistics (Pade order and
// squarings Julia's exp! would use) come from the 1-norm bound of the operators or, outside its certifying
// window, from the measured norm, exactly as in expm_persistent; the executed work is counted separately.
template <int NT>
__global__ void __launch_bounds__(NT * 64) expm_t18_kernel(ExpmArgs a) {
    using LY = ExpmLds<NT>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *red = smem + 2 * LY::REG + LY::DV;
    const int tid0 = threadIdx.x, lane0 = tid0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int ncell = a.K * a.N_T;
    const int x = blockIdx.x & 7, per_x = gridDim.x >> 3;
    const int lo = (int)((long)x * ncell / 8), hi = (int)((long)(x + 1) * ncell / 8);
    int st_s = 0, st_max = 0, st_ord[5] = {0, 0, 0, 0, 0}, st_sq = 0, st_cells = 0;
    bool any_bad = false;
    for (int cell = lo + ((int)blockIdx.x >> 3); cell < hi; cell += per_x) {
        int lane = lane0, tid = tid0;
        asm volatile("" : "+v"(lane), "+v"(tid));   // per-lane addresses are recomputed per cell (see expm_persistent)
        const double bound = expm_norm_bound(a, cell);
        expm_form_a_herm64<64 * NT, NT>(a, cell, smem, tid);
        __syncthreads();
        // credited work: what Julia's exp! would do for this cell (order and squarings from ||A||_1)
        double nA = bound;
        if (!(bound > 2.1 && bound <= 5.4)) {
            expm_norm_partial<NT>(smem, tid, LY::NTH / LY::NP);
            __syncthreads();
            expm_norm_combine<NT>(smem, tid, LY::NTH / LY::NP);
            __syncthreads();
            nA = red[LY::NTH];
            __syncthreads();   // red is written again inside the cell
        }
        int sj = 0;
        if (nA > 5.4) {
            const double r = nA / 5.4;
            const int e = ilogb(r);
            sj = (r == ldexp(1.0, e)) ? e : e + 1;
        }
        const int oj = nA > 2.1 ? 4 : nA > 0.95 ? 3 : nA > 0.25 ? 2 : nA > 0.015 ? 1 : 0;
        Strip<NT> T;
        int s;
        bool bad;
        expm_t18_cell<NT>(smem, wave, lane, T, s, bad);
        t18_store_u<NT>(a, cell, wave, lane, T);
        any_bad |= bad;
        st_s += sj; st_max = max(st_max, sj); st_ord[oj] += 1;
        st_sq += s; st_cells += 1;
        __syncthreads();   // the next cell writes the A region (read by slow waves as exchange area / product operand)
    }
    if (tid0 == 0) {
        stat_add(a.stats, 0, (unsigned long long)st_s);
#pragma unroll
        for (int o = 0; o < 5; ++o)
            if (st_ord[o]) stat_add(a.stats, 3 + o, (unsigned long long)st_ord[o]);
        if (st_max > 0) atomicMax(&a.flags[1], st_max);
        // executed matrix instructions (all waves), squarings and cells of this path
        stat_add(a.stats, 12, (unsigned long long)NT * ((unsigned long long)st_cells * T18Count<NT>::CELL + (unsigned long long)st_sq * T18Count<NT>::GP));
        stat_add(a.stats, 13, (unsigned long long)st_sq);
        stat_add(a.stats, 14, (unsigned long long)st_cells);
        if (any_bad) atomicOr(&a.flags[0], 64);
    }
}
