// grape_large.hip.h -- blocked variant of the path for 64 < N <= 256 (NP = 128 or 256).
//
// Same algorithm as grape_kernels.hip.h, but a cell's matrices no longer fit one workgroup's
// registers/LDS: they live in HBM/L2 as planar (re plane | im plane) row-major NP x NP arrays, cells
// are processed in chunks, and every product of the Pade evaluation is one launch of a batched,
// MFMA-tiled block GEMM (64x64 output block per workgroup, K loop over 64-wide blocks, the same
// LDS-left-operand / register-strip-right-operand tile engine `gemm_xb<4,66>` as the fused kernel).
// The Pade system is solved by block Gauss-Jordan at 64-block granularity: the 64x64 diagonal block is
// inverted by the fused kernel's in-register solver (P = I), row scaling and rank-64 updates are block
// GEMMs.  Sweeps use 1024-thread workgroups; the derivative overlaps use deriv_mfma_kernel.
#pragma once
#include "grape_kernels.hip.h"

struct LgView {          // a block view into a batch of planar matrices
    double *p;           // re plane of cell 0; the im plane follows at +plane
    size_t cell_stride;  // doubles between consecutive cells
    size_t plane;        // doubles between re and im plane
    int ld;              // leading dimension (row stride)
    int rb, cb;          // block offset of the view (units of 64 rows / columns)
};

struct LgGemmArgs {
    LgView X, Y, C;      // C[bi][bj] = alpha * sum_kb X[bi][kb] Y[kb][bj] + beta * C[bi][bj] + sum_i coef[i] Add_i + cI * I
    LgView Add[4];
    double coef[4];
    double alpha, beta, cI;
    // second output of the same launch (C2.p != nullptr): C2 = C + sum_i coef2[i] Add_i + cI2 * I -- the polynomial route
    // gets A9 = B1 B5 + B4 and B3 + A9 from one product (B3 and B4 combine the same four blocks)
    LgView C2;
    double coef2[4], cI2;
    int add_pow[4];      // with scale_s: coef[i] and coef2[i] are multiplied by 2^(-add_pow[i] * scale_s[cell]) (Add_i = A^add_pow[i]
                         // of the UNSCALED A: the scaling A / 2^s of scaling and squaring, exact in binary)
    int nadd, kblocks;
    int nbi, nbj, ncell; // output blocks per cell and cells in this launch
    const int *s_cell;   // optional: squarings per cell; cells with s_cell[cell] <= sq_iter copy X instead
    int sq_iter;
    int herm;            // +1 / -1: the product is Hermitian / skew-Hermitian (nbi == nbj): only the blocks bi <= bj are
                         // computed, each workgroup also stores the (signed) conjugate transpose of its block at (bj, bi)
    const int *scale_s;  // optional: alpha *= 2^(-scale_pow * scale_s[cell]) -- the scaling A / 2^s of scaling and squaring
    int scale_pow;       //   applied where A is consumed (A*A: 2, A*T: 1) instead of in a pass over A; exact in binary
    double2 *Uout;       // optional: the result goes to U[cell][row][col] (interleaved complex) instead of C
    int u_np;
    int skip_bi;         // block row that is left alone (-1: none): the trailing update of a Gauss-Jordan step covers all
                         // block rows but the pivot row in ONE launch
    int u_if_smax0;      // last product of the polynomial route: the result goes to Uout when *smax_ptr == 0 (no cell of the
                         // evaluation so far needs a squaring), to C otherwise
    const int *smax_ptr; // squarings only: the launch belongs to iteration sq_iter of a plan of fixed length; it exits at
                         // once when sq_iter >= *smax_ptr (the largest squaring count so far, known only on the device)
                         // and writes U when it is the last one -- no host read-back of the count between launches
};

__device__ __forceinline__ double *lg_ptr(const LgView &v, int cell, int brow, int bcol) {
    return v.p + (size_t)cell * v.cell_stride + (size_t)(v.rb + brow) * 64 * v.ld + (size_t)(v.cb + bcol) * 64;
}

__global__ void __launch_bounds__(256, 2) lg_gemm_kernel(LgGemmArgs a) {  // 2 workgroups per CU: one loads while the other issues MFMAs
    constexpr int LD = 66;
    __shared__ double Xre[64 * LD], Xim[64 * LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware 1-D grid: the nbi*nbj blocks of one cell share blockIdx % 8, i.e. one XCD and its L2,
    // so that the row panels of X and the column panels of Y are fetched from HBM once per cell.
    const int per_cell = a.herm ? a.nbi * (a.nbi + 1) / 2 : a.nbi * a.nbj;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int cell = (q / per_cell) * 8 + xcd;
    if (cell >= a.ncell) return;
    const int rem = q % per_cell;
    int bi, bj;
    if (a.herm) {   // upper triangle, row by row
        int r = rem;
        bi = 0;
        while (r >= a.nbi - bi) { r -= a.nbi - bi; ++bi; }
        bj = bi + r;
    } else {
        bi = rem / a.nbj; bj = rem - bi * a.nbj;
    }
    if (bi == a.skip_bi) return;
    double2 *Uout = a.Uout;
    if (a.u_if_smax0) {
        if (*a.smax_ptr != 0) Uout = nullptr;
    } else if (a.smax_ptr) {
        const int sm = *a.smax_ptr;
        if (a.sq_iter >= sm) return;
        if (a.sq_iter != sm - 1) Uout = nullptr;
    }
    const int col = 16 * wave + (lane & 15), rg = lane >> 4;
    double *c = lg_ptr(a.C, cell, bi, bj);
    if (a.s_cell && a.s_cell[cell] <= a.sq_iter) {   // no (further) squaring for this cell: C = X
        const double *x = lg_ptr(a.X, cell, bi, bj);
        for (int idx = tid; idx < 64 * 64; idx += 256) {
            const int i = idx >> 6, j = idx & 63;
            const double vr = x[(size_t)i * a.X.ld + j], vi = x[a.X.plane + (size_t)i * a.X.ld + j];
            if (Uout) {
                Uout[(size_t)cell * a.u_np * a.u_np + (size_t)(bi * 64 + i) * a.u_np + bj * 64 + j] = make_double2(vr, vi);
            } else {
                c[(size_t)i * a.C.ld + j] = vr;
                c[a.C.plane + (size_t)i * a.C.ld + j] = vi;
            }
        }
        return;
    }
    const double alpha = a.scale_s ? ldexp(a.alpha, -a.scale_pow * a.scale_s[cell]) : a.alpha;
    double cf[4], cf2[4];
    for (int q = 0; q < a.nadd; ++q) {
        const double f = (a.scale_s && a.add_pow[q]) ? ldexp(1.0, -a.add_pow[q] * a.scale_s[cell]) : 1.0;
        cf[q] = a.coef[q] * f;
        cf2[q] = a.coef2[q] * f;
    }
    double *c2o = a.C2.p ? lg_ptr(a.C2, cell, bi, bj) : nullptr;
    Strip3<4> acc3;   // 3M partial products, combined once after the K loop
    strip3_zero(acc3);
    for (int kb = 0; kb < a.kblocks; ++kb) {
        const double *x = lg_ptr(a.X, cell, bi, kb);
        const double *y = lg_ptr(a.Y, cell, kb, bj);
        __syncthreads();
        for (int idx = tid; idx < 64 * 32; idx += 256) {   // 16-byte loads: 2 columns at a time
            const int i = idx >> 5, j = (idx & 31) * 2;
            const double2 vr = *(const double2 *)(x + (size_t)i * a.X.ld + j);
            const double2 vi = *(const double2 *)(x + a.X.plane + (size_t)i * a.X.ld + j);
            Xre[i * LD + j] = vr.x; Xre[i * LD + j + 1] = vr.y;
            Xim[i * LD + j] = vi.x; Xim[i * LD + j + 1] = vi.y;
        }
        Strip<4> B;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t o = (size_t)(16 * t + 4 * r + rg) * a.Y.ld + col;
                B.re[t][r] = y[o];
                B.im[t][r] = y[a.Y.plane + o];
            }
        __syncthreads();
        gemm_xb3<4, LD>(acc3, Xre, Xim, B, lane);
    }
    Strip<4> acc;
    strip_zero(acc);
    strip3_add_to(acc3, acc);
    const int grow0 = (a.C.rb + bi) * 64, gcol = (a.C.cb + bj) * 64 + col;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + 4 * r + rg;
            const size_t o = (size_t)row * a.C.ld + col;
            double vr = alpha * acc.re[t][r], vi = alpha * acc.im[t][r];
            if (a.beta != 0.0) { vr += a.beta * c[o]; vi += a.beta * c[a.C.plane + o]; }
            double wr = 0., wi = 0.;
            for (int q = 0; q < a.nadd; ++q) {
                const double *ad = lg_ptr(a.Add[q], cell, bi, bj);
                const size_t oa = (size_t)row * a.Add[q].ld + col;
                const double xr = ad[oa], xi = ad[a.Add[q].plane + oa];
                vr += cf[q] * xr;
                vi += cf[q] * xi;
                wr += cf2[q] * xr;
                wi += cf2[q] * xi;
            }
            if (a.cI != 0.0 && grow0 + row == gcol) vr += a.cI;
            if (c2o) {
                const size_t o2 = (size_t)row * a.C2.ld + col;
                c2o[o2] = vr + wr + ((grow0 + row == gcol) ? a.cI2 : 0.0);
                c2o[a.C2.plane + o2] = vi + wi;
            }
            if (Uout) {
                Uout[(size_t)cell * a.u_np * a.u_np + (size_t)(bi * 64 + row) * a.u_np + bj * 64 + col] = make_double2(vr, vi);
            } else {
                c[o] = vr;
                c[a.C.plane + o] = vi;
            }
            acc.re[t][r] = vr;   // kept for the mirrored block
            acc.im[t][r] = vi;
        }
    if (a.herm && bi != bj) {
        // mirrored block (bj, bi) = sgn * conj(transpose): staged through the (now idle) LDS tile so that the
        // global stores stay row-contiguous
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + 4 * r + rg;
                Xre[row * LD + col] = acc.re[t][r];
                Xim[row * LD + col] = acc.im[t][r];
            }
        __syncthreads();
        double *c2 = lg_ptr(a.C, cell, bj, bi);
        const double sg = (double)a.herm;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + 4 * r + rg;   // element (row, col) of the mirrored block = conj of (col, row)
                const size_t o = (size_t)row * a.C.ld + col;
                c2[o] = sg * Xre[col * LD + row];
                c2[a.C.plane + o] = -sg * Xim[col * LD + row];
            }
    }
}

// out = sum_i coef[i] * In_i  (whole NP x NP matrices, both planes)
struct LgLincombArgs {
    double *out;
    const double *in[3];
    double coef[3];
    int nin;
    size_t n;   // doubles per cell (2 * NP * NP) * cells
};
// two combinations of the same inputs in one pass (W and Z of the order-13 polynomials; P = V+U and Q = V-U)
struct LgLincomb2Args {
    double *out0, *out1;
    const double *in[3];
    double c0[3], c1[3];
    int nin;
    size_t n;
};
__global__ void lg_lincomb2_kernel(LgLincomb2Args a) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
        double v0 = 0., v1 = 0.;
        for (int q = 0; q < a.nin; ++q) {
            const double x = a.in[q][i];
            v0 += a.c0[q] * x;
            v1 += a.c1[q] * x;
        }
        a.out0[i] = v0;
        a.out1[i] = v1;
    }
}
__global__ void lg_lincomb_kernel(LgLincombArgs a) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
        double v = 0.;
        for (int q = 0; q < a.nin; ++q) v += a.coef[q] * a.in[q][i];
        a.out[i] = v;
    }
}

// A = -i dt (H0_k + sum_l e_l H_l) for the cells [cell0, cell0 + ncell) of the chunk; then ||A||_1,
// squaring count s and the scaling A *= 2^-s.  One 256-thread workgroup per cell.
struct LgFormArgs {
    const double *H0f, *Hcf, *eps, *shape, *dts;
    double *A;            // [ncell][2][NP*NP]
    int *s_cell;          // [ncell]
    unsigned long long *stats;
    int *flags;
    int NP, L, N_T, hc_per_traj, cell0;
    const int *rep;       // nullptr or representative trajectory per generator class
    double *norm1;        // polynomial route (lg_t18_decide_kernel decides the scaling): ||A||_1 per cell goes here, the
                          // Pade squaring count is only CREDITED (statistics), s_cell and flags[1] are left alone
    const double *Sf;     // nullptr or [N_T][2][NP*NP]: S_n = sum_l eps_ln shape_ln H_l of every time step (ctrl_sum_kernel, once
                          // per evaluation; control operators shared by the trajectories).  The cell then reads H0_k and S_n --
                          // two operators instead of 1 + L: the kernel was bound by what one CU pulls from the L2 (10 MB per cell at
                          // C5: 0.41 ms per chunk of 635 cells where the 0.67 GB it writes take 0.13; round 5)
};
// 1024 threads per cell: thread (part, j) forms rows part, part + 4, ... of column j.  (With 256 threads -- one per column,
// 256 rows each -- a launch lasted as long as ONE workgroup's chain of 32 load round trips: 0.63 ms per chunk of 889 cells,
// 1 TB/s for a kernel that only writes; round 3.)
__global__ void __launch_bounds__(1024) lg_form_kernel(LgFormArgs a) {
    __shared__ double colsum[1024];
    __shared__ double wmax[4];
    __shared__ double snorm;
    const int tid = threadIdx.x & 255, part = threadIdx.x >> 8, NP = a.NP;
    const int cell = a.cell0 + blockIdx.x;
    const int kc = cell / a.N_T, n = cell - kc * a.N_T;
    const int k = a.rep ? a.rep[kc] : kc;
    const double dt = a.dts[n];
    const size_t pp = (size_t)NP * NP;
    const double *h0 = a.H0f + (size_t)k * 2 * pp;
    const double *hc = a.Hcf + (size_t)(a.hc_per_traj ? k : 0) * a.L * 2 * pp;
    double *A = a.A + (size_t)blockIdx.x * 2 * pp;
    double e[8];
    for (int l = 0; l < a.L; ++l) {
        e[l] = a.eps[(size_t)l * a.N_T + n];
        if (a.shape) e[l] *= a.shape[(size_t)l * a.N_T + n];
    }
    // thread (part, j) owns every fourth row of column j (NP <= 256).  ONE pass over the (L2-resident) operators:
    // A is written unscaled together with the column sums of |a_ij| for the 1-norm; the scaling A / 2^s of scaling and
    // squaring is applied as a per-cell power of two where A is consumed (LgGemmArgs::scale_s: A*A and A*T) --
    // exact in binary, and the operators are read once instead of twice.  Rows are unrolled by 8 so that 8 (1 + L)
    // independent loads per plane are in flight (the loop is otherwise bound by one load latency per row).
    const double *sn = a.Sf ? a.Sf + (size_t)n * 2 * pp : nullptr;
    auto element = [&](int i, double &ar, double &ai) __attribute__((always_inline)) {
        const size_t o = (size_t)i * NP + tid;
        double hr = h0[o], hi = h0[pp + o];
        if (sn) {
            hr += sn[o]; hi += sn[pp + o];
        } else {
            for (int l = 0; l < a.L; ++l) {
                hr = fma(e[l], hc[(size_t)l * 2 * pp + o], hr);
                hi = fma(e[l], hc[(size_t)l * 2 * pp + pp + o], hi);
            }
        }
        ar = dt * hi; ai = -dt * hr;
    };
    double cs = 0.;
    if (tid < NP) {
#pragma unroll 8
        for (int i = part; i < NP; i += 4) {
            double ar, ai;
            element(i, ar, ai);
            cs += sqrt(ar * ar + ai * ai);
            const size_t o = (size_t)i * NP + tid;
            A[o] = ar;
            A[pp + o] = ai;
        }
    }
    colsum[threadIdx.x] = tid < NP ? cs : 0.;
    __syncthreads();
    if (threadIdx.x < 256) {   // column sums of the four row parts, then their maximum: waves 0..3 by shuffles, the four waves through LDS
        double m = tid < NP ? (colsum[tid] + colsum[256 + tid]) + (colsum[512 + tid] + colsum[768 + tid]) : 0.;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) m = fmax(m, __shfl_xor(m, off, 64));
        if ((tid & 63) == 0) wmax[tid >> 6] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) snorm = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
    __syncthreads();
    const double nA = snorm;
    int s = 0;
    if (nA > 5.4) {
        const double r = nA / 5.4;
        const int ex = ilogb(r);
        s = (r == ldexp(1.0, ex)) ? ex : ex + 1;
    }
    if (threadIdx.x == 0) {
        stat_add(a.stats, 0, (unsigned long long)s);
        stat_add(a.stats, 7, 1ull);   // the blocked path always evaluates (or is credited with) the order-13 approximant
        if (a.norm1) {
            a.norm1[blockIdx.x] = nA;
        } else {
            a.s_cell[blockIdx.x] = s;
            atomicMax(&a.flags[1], s);
        }
    }
}

// Round 5: formation with the operators in registers.  lg_form_kernel (one workgroup per cell) reads H0_k and S_n for every
// cell and writes A: 33.6 GB per C5-shard evaluation at 4.5 TB/s, 7.4 ms (+ 0.7 ms for the S_n).  Here a workgroup owns
// LG_FORM_ROWS rows of the matrix for a SLICE OF CELLS: thread j keeps its column's elements of H0_k and of the (at most
// four, shared) control operators in registers and only writes -- 16.8 GB, plus the partial column sums of |a_ij| (1/16 of
// that), which lg_norm1_kernel turns into ||A||_1 per cell (same sums, same order for every launch geometry).
constexpr int LG_FORM_ROWS = 8;
template <int LMAX>
__global__ void __launch_bounds__(256) lg_form2_kernel(LgFormArgs a, int ncell, double *normpart) {
    const int NP = a.NP, j = threadIdx.x, rg = blockIdx.x, i0 = rg * LG_FORM_ROWS, ngroups = NP / LG_FORM_ROWS;
    if (j >= NP) return;
    const size_t pp = (size_t)NP * NP;
    const int per = (ncell + gridDim.y - 1) / gridDim.y, c_lo = blockIdx.y * per, c_hi = min(ncell, c_lo + per);
    double hr[LG_FORM_ROWS], hi[LG_FORM_ROWS], cr[LMAX][LG_FORM_ROWS], ci[LMAX][LG_FORM_ROWS];
    int k_have = -1;
    for (int c = c_lo; c < c_hi; ++c) {
        const int cell = a.cell0 + c, kc = cell / a.N_T, n = cell - kc * a.N_T;
        const int k = a.rep ? a.rep[kc] : kc;
        if (k != k_have) {   // (a chunk holds the cells of one trajectory, rarely of two)
            const double *h0 = a.H0f + (size_t)k * 2 * pp;
#pragma unroll
            for (int r = 0; r < LG_FORM_ROWS; ++r) {
                const size_t o = (size_t)(i0 + r) * NP + j;
                hr[r] = h0[o]; hi[r] = h0[pp + o];
            }
            if (k_have < 0) {
#pragma unroll
                for (int l = 0; l < LMAX; ++l)
#pragma unroll
                    for (int r = 0; r < LG_FORM_ROWS; ++r) {
                        const size_t o = (size_t)l * 2 * pp + (size_t)(i0 + r) * NP + j;
                        cr[l][r] = l < a.L ? a.Hcf[o] : 0.0; ci[l][r] = l < a.L ? a.Hcf[pp + o] : 0.0;
                    }
            }
            k_have = k;
        }
        const double dt = a.dts[n];
        double e[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            e[l] = l < a.L ? a.eps[(size_t)l * a.N_T + n] : 0.0;
            if (a.shape && l < a.L) e[l] *= a.shape[(size_t)l * a.N_T + n];
        }
        double *A = a.A + (size_t)c * 2 * pp;
        double cs = 0.;
#pragma unroll
        for (int r = 0; r < LG_FORM_ROWS; ++r) {
            double xr = hr[r], xi = hi[r];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) { xr = fma(e[l], cr[l][r], xr); xi = fma(e[l], ci[l][r], xi); }
            const double ar = dt * xi, ai = -dt * xr;
            cs += sqrt(ar * ar + ai * ai);
            const size_t o = (size_t)(i0 + r) * NP + j;
            __builtin_nontemporal_store(ar, A + o);
            __builtin_nontemporal_store(ai, A + pp + o);
        }
        normpart[((size_t)c * ngroups + rg) * NP + j] = cs;
    }
}
// ||A||_1 per cell from the partial column sums of lg_form2_kernel (one workgroup per cell, thread j owns column j): what the
// tail of lg_form_kernel does
__global__ void __launch_bounds__(256) lg_norm1_kernel(LgFormArgs a, const double *normpart) {
    __shared__ double wmax[4];
    const int NP = a.NP, tid = threadIdx.x, ngroups = NP / LG_FORM_ROWS;
    double m = 0.;
    if (tid < NP) {
        const double *np_ = normpart + (size_t)blockIdx.x * ngroups * NP + tid;
        for (int g = 0; g < ngroups; ++g) m += np_[(size_t)g * NP];   // fixed order
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((tid & 63) == 0) wmax[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        const double nA = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
        int s = 0;
        if (nA > 5.4) {
            const double r = nA / 5.4;
            const int ex = ilogb(r);
            s = (r == ldexp(1.0, ex)) ? ex : ex + 1;
        }
        stat_add(a.stats, 0, (unsigned long long)s);
        stat_add(a.stats, 7, 1ull);
        a.norm1[blockIdx.x] = nA;
    }
}

// ---- polynomial route of the blocked path (five products, no solve; scheme and coefficients: grape_t18_coeffs.h) ----
// Scaling of one cell from the powers: Hermitian generators beta = min(||A2||_1^(1/2), ||A6||_1^(1/6)) against theta = 2
// (spectral bound, see grape_t18.hip.h); general matrices alpha = min(||A||_1, max(||A2||_1^(1/2), ||A3||_1^(1/3))) against
// theta = 1.09.  P and Q are the two powers whose column sums are needed (A2 and A6, or A2 and A3); |re| + |im| stands in
// for the modulus (an upper bound of the norm is all the theory needs).  One 256-thread workgroup per cell, thread j
// owns column j.
struct LgT18ScaleArgs {
    const double *P, *Q;   // [ncell][2][NP*NP]
    const double *norm1;   // ||A||_1 per cell (general matrices), nullptr for Hermitian generators
    int *s_cell;
    int *flags;
    int *smax;             // the largest squaring count of the evaluation so far, per lane of chunks (flags + 1 for the first lane)
    unsigned long long *stats;
    int NP, qpow;          // Q = A^qpow (6 or 3)
    double theta;
    unsigned long long mfma_per_cell, mfma_per_sq;   // executed matrix instructions (all waves): statistics
};
__global__ void __launch_bounds__(256) lg_t18_scale_kernel(LgT18ScaleArgs a) {
    __shared__ double cs[2][256];
    const int tid = threadIdx.x, NP = a.NP;
    const size_t pp = (size_t)NP * NP;
    const double *P = a.P + (size_t)blockIdx.x * 2 * pp, *Q = a.Q + (size_t)blockIdx.x * 2 * pp;
    double sp = 0., sq = 0.;
    if (tid < NP) {
#pragma unroll 8
        for (int i = 0; i < NP; ++i) {
            const size_t o = (size_t)i * NP + tid;
            sp += fabs(P[o]) + fabs(P[pp + o]);
            sq += fabs(Q[o]) + fabs(Q[pp + o]);
        }
    }
    cs[0][tid] = sp; cs[1][tid] = sq;
    __syncthreads();
    if (tid == 0) {
        double np_ = 0., nq = 0.;
        for (int j = 0; j < NP; ++j) { np_ = fmax(np_, cs[0][j]); nq = fmax(nq, cs[1][j]); }
        np_ *= 1.0 + 1e-9; nq *= 1.0 + 1e-9;   // rounding of the computed powers
        int s = 0;
        bool bad = false;
        if (!a.norm1) {   // beta <= theta 2^s  <=>  ||A2|| <= (theta 2^s)^2  or  ||A6|| <= (theta 2^s)^6
            double t2 = a.theta * a.theta, t6 = t2 * t2 * t2;
            while (!(np_ <= t2 || nq <= t6) && s < 64) { ++s; t2 *= 4.0; t6 *= 64.0; }
            bad = s >= 64;
        } else {          // alpha <= theta 2^s  <=>  ||A|| <= theta 2^s  or  (||A2|| <= (theta 2^s)^2 and ||A3|| <= (theta 2^s)^3)
            const double n1 = a.norm1[blockIdx.x];
            double t1 = a.theta, t2 = t1 * t1, t3 = t2 * t1;
            while (!(n1 <= t1 || (np_ <= t2 && nq <= t3)) && s < 64) { ++s; t1 *= 2.0; t2 *= 4.0; t3 *= 8.0; }
            bad = s >= 64;
        }
        if (bad) { s = 0; atomicOr(&a.flags[0], 64); }   // NaN / overflow in the generator
        a.s_cell[blockIdx.x] = s;
        atomicMax(a.smax, s);
        stat_add(a.stats, 12, a.mfma_per_cell + (unsigned long long)s * a.mfma_per_sq);
        stat_add(a.stats, 13, (unsigned long long)s);
        stat_add(a.stats, 14, 1ull);
    }
}

// The five linear combinations of the scaled powers (A / 2^s)^k = A^k 2^(-k s) in ONE streaming pass (4 reads, 5 writes per
// element at the HBM rate): B1, B5 (operands of the fourth product), B4, B3, B2 (added in the epilogues of the fourth and
// fifth product: one or two extra blocks per output block -- with the four powers combined in the epilogue itself the
// two general products took 2.1 ms per chunk instead of 1.1, their block loads being issued one element at a time).
// B4 and B3 include their multiples of the identity.
struct LgT18OperandsArgs {
    const double *A, *A2, *A3, *A6;
    double *B1, *B5, *B4, *B3, *B2;
    const int *s_cell;
    double a[3], e[3], b[5], c[5], d[5];   // b, c, d: I, A, A2, A3, A6
    int NP;
    size_t per_cell;   // 2 * NP * NP
    size_t n;          // per_cell * cells
};
__global__ void lg_t18_operands_kernel(LgT18OperandsArgs a) {
    const size_t pp = a.per_cell / 2;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t cell = i / a.per_cell, o = i - cell * a.per_cell;
        const int s = a.s_cell[cell];
        const double f1 = ldexp(1.0, -s), f2 = f1 * f1, f3 = f2 * f1, f6 = f3 * f3;
        const double x1 = f1 * a.A[i], x2 = f2 * a.A2[i], x3 = f3 * a.A3[i], x6 = f6 * a.A6[i];
        const bool diag = o < pp && (o / a.NP) == (o % a.NP);   // real plane, row == column
        a.B1[i] = a.a[0] * x1 + a.a[1] * x2 + a.a[2] * x3;
        a.B5[i] = a.e[0] * x2 + a.e[1] * x3 + a.e[2] * x6;
        a.B4[i] = a.d[1] * x1 + a.d[2] * x2 + a.d[3] * x3 + a.d[4] * x6 + (diag ? a.d[0] : 0.0);
        a.B3[i] = a.c[1] * x1 + a.c[2] * x2 + a.c[3] * x3 + a.c[4] * x6 + (diag ? a.c[0] : 0.0);
        a.B2[i] = a.b[1] * x1 + a.b[2] * x2 + a.b[3] * x3 + a.b[4] * x6 + (diag ? a.b[0] : 0.0);
    }
}
// Round 5: the norm pass rides in the combination pass.  lg_t18_scale_kernel read two powers of every cell (33.6 GB per
// C5-shard evaluation, 6.4 ms) only to find, at the benchmark's norms, that no cell needs a squaring.  Now the combination
// pass forms B1 .. B5 SPECULATIVELY for s = 0 and takes the column sums of the two powers it reads anyway on its way;
// lg_t18_decide_kernel turns the partial sums into the squaring count of every cell (same bound, same statistics), and a
// second launch of the combination pass redoes the cells with s > 0 from the intact powers (it leaves at once for the
// others: at C5 for all of them).  The partial sums are written per row part and added in a fixed order: the decision of a
// cell does not depend on the order in which workgroups finish.
constexpr int LG_PARTS = 16;      // row parts per cell (16 rows each at NP = 256)
struct LgT18Operands2Args {
    LgT18OperandsArgs o;
    double *colpart;       // [ncell][2][LG_PARTS][NP]: column sums of |re| + |im| of P = A2 and Q = A6 (Hermitian) / A3 (general)
    int q_is_a6;           // 1: Q = A6, 0: Q = A3
    int redo;              // 0: every cell with s = 0 (+ partial sums); 1: only the cells with s_cell > 0, with their scaling
};
__global__ void __launch_bounds__(256) lg_t18_operands2_kernel(LgT18Operands2Args g) {
    const LgT18OperandsArgs &a = g.o;
    const int cell = blockIdx.x / LG_PARTS, part = blockIdx.x - cell * LG_PARTS, NP = a.NP, tid = threadIdx.x;
    int s = 0;
    if (g.redo) {
        s = a.s_cell[cell];
        if (s == 0) return;
    }
    if (tid >= NP) return;
    const size_t pp = a.per_cell / 2, base = (size_t)cell * a.per_cell;
    const int rows = NP / LG_PARTS, i0 = part * rows;
    const double f1 = ldexp(1.0, -s), f2 = f1 * f1, f3 = f2 * f1, f6 = f3 * f3;
    double sp = 0., sq = 0.;
#pragma unroll 4
    for (int i = i0; i < i0 + rows; ++i) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const size_t q = base + (size_t)pl * pp + (size_t)i * NP + tid;
            const double r1 = __builtin_nontemporal_load(a.A + q), r2 = __builtin_nontemporal_load(a.A2 + q);
            const double r3 = __builtin_nontemporal_load(a.A3 + q), r6 = __builtin_nontemporal_load(a.A6 + q);
            sp += fabs(r2);
            sq += fabs(g.q_is_a6 ? r6 : r3);
            const double x1 = f1 * r1, x2 = f2 * r2, x3 = f3 * r3, x6 = f6 * r6;
            const bool diag = pl == 0 && i == tid;
            a.B1[q] = a.a[0] * x1 + a.a[1] * x2 + a.a[2] * x3;
            a.B5[q] = a.e[0] * x2 + a.e[1] * x3 + a.e[2] * x6;
            a.B4[q] = a.d[1] * x1 + a.d[2] * x2 + a.d[3] * x3 + a.d[4] * x6 + (diag ? a.d[0] : 0.0);
            a.B3[q] = a.c[1] * x1 + a.c[2] * x2 + a.c[3] * x3 + a.c[4] * x6 + (diag ? a.c[0] : 0.0);
            a.B2[q] = a.b[1] * x1 + a.b[2] * x2 + a.b[3] * x3 + a.b[4] * x6 + (diag ? a.b[0] : 0.0);
        }
    }
    if (!g.redo) {
        double *cp = g.colpart + ((size_t)cell * 2 * LG_PARTS + part) * NP;
        cp[tid] = sp;
        cp[(size_t)LG_PARTS * NP + tid] = sq;
    }
}
// Round 5 (second half): the launch that writes the last power forms the combinations in its epilogue (lg_gemm_asm `comb`,
// asm/gen_lg.py: every block, with the column sums of its 64 rows in row part bi of the column-sum scratch); the pass above
// is then only launched for the cells that need a scaling (redo = 1) and as the twin (GRAPE_LG_FUSE=0).
// the decision of lg_t18_scale_kernel from the partial column sums (one workgroup per cell, thread j owns column j)
struct LgT18DecideArgs {
    const double *colpart;
    LgT18ScaleArgs s;      // P, Q unused
    int nparts;            // row parts that were written: LG_PARTS (lg_t18_operands2_kernel) or NP / 64 (fused epilogue)
    // round 6 -- the economized derivative series (deriv_econ_kernel): for a Hermitian cell without scaling the smallest
    // segment theta_i with ||A^2|| <= theta_i^2 or ||A^6|| <= theta_i^6 (the spectral radius of a normal matrix is at most
    // ||A^k||^(1/k)) names the degree of its polynomial; cell_deg[cell0 + cell] = that degree, 0: none
    int *cell_deg;         // nullptr: not asked for
    int cell0, econ_n;
    double econ_theta[4];
    int econ_deg[4];
};
__global__ void __launch_bounds__(256) lg_t18_decide_kernel(LgT18DecideArgs g) {
    const LgT18ScaleArgs &a = g.s;
    __shared__ double cs[2][256];
    const int tid = threadIdx.x, NP = a.NP;
    double sp = 0., sq = 0.;
    if (tid < NP) {
        const double *cp = g.colpart + (size_t)blockIdx.x * 2 * LG_PARTS * NP;
        for (int part = 0; part < g.nparts; ++part) {   // fixed order
            sp += cp[(size_t)part * NP + tid];
            sq += cp[(size_t)(LG_PARTS + part) * NP + tid];
        }
    }
    cs[0][tid] = sp; cs[1][tid] = sq;
    __syncthreads();
    if (tid == 0) {
        double np_ = 0., nq = 0.;
        bool bad = false;                        // (fmax drops a NaN: a column sum that is not finite is looked for)
        for (int j = 0; j < NP; ++j) {
            np_ = fmax(np_, cs[0][j]); nq = fmax(nq, cs[1][j]);
            bad = bad || !(cs[0][j] <= 1.7e308) || !(cs[1][j] <= 1.7e308);
        }
        np_ *= 1.0 + 1e-9; nq *= 1.0 + 1e-9;   // rounding of the computed powers
        int s = 0;
        if (!a.norm1) {   // beta <= theta 2^s  <=>  ||A2|| <= (theta 2^s)^2  or  ||A6|| <= (theta 2^s)^6
            double t2 = a.theta * a.theta, t6 = t2 * t2 * t2;
            while (!(np_ <= t2 || nq <= t6) && s < 64) { ++s; t2 *= 4.0; t6 *= 64.0; }
            bad = bad || s >= 64;
        } else {          // alpha <= theta 2^s  <=>  ||A|| <= theta 2^s  or  (||A2|| <= (theta 2^s)^2 and ||A3|| <= (theta 2^s)^3)
            const double n1 = a.norm1[blockIdx.x];
            double t1 = a.theta, t2 = t1 * t1, t3 = t2 * t1;
            while (!(n1 <= t1 || (np_ <= t2 && nq <= t3)) && s < 64) { ++s; t1 *= 2.0; t2 *= 4.0; t3 *= 8.0; }
            bad = bad || s >= 64;
        }
        if (g.cell_deg) {
            int deg = 0;
            if (!a.norm1 && s == 0 && !bad)
                for (int i = 0; i < g.econ_n && !deg; ++i) {
                    const double t2 = g.econ_theta[i] * g.econ_theta[i];
                    if (np_ <= t2 || nq <= t2 * t2 * t2) deg = g.econ_deg[i];
                }
            g.cell_deg[g.cell0 + blockIdx.x] = deg;
        }
        if (bad) { s = 0; atomicOr(&a.flags[0], 64); }   // NaN / overflow in the generator
        a.s_cell[blockIdx.x] = s;
        atomicMax(a.smax, s);
        stat_add(a.stats, 12, a.mfma_per_cell + (unsigned long long)s * a.mfma_per_sq);
        stat_add(a.stats, 13, (unsigned long long)s);
        stat_add(a.stats, 14, 1ull);
    }
}

// Dinv = inverse of the 64x64 block (jb, jb) of Q, one workgroup per cell (fused-kernel solver, P = I)
struct LgInvArgs {
    LgView Q;         // view positioned at block (jb, jb)
    double *Dinv;     // [ncell][2][64*64]
    int *flags;
    double inv_scale2;
    int *cellflag;    // [chunk cells] set when a pivot of the unpivoted elimination is numerically unsafe
};
__global__ void __launch_bounds__(256) lg_inv64_kernel(LgInvArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem_inv[];
    double *pan = smem_inv;                    // 3 panel slots of 2*64*18 doubles
    double *dv = pan + 3 * 2 * 64 * 18;        // 3 x 512 doubles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cell = blockIdx.x;
    const double *q = lg_ptr(a.Q, cell, 0, 0);
    const int col = 16 * wave + (lane & 15), rg = lane >> 4;
    Strip<4> Q, P;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + 4 * r + rg;
            Q.re[t][r] = q[(size_t)row * a.Q.ld + col];
            Q.im[t][r] = q[a.Q.plane + (size_t)row * a.Q.ld + col];
            P.re[t][r] = row == col ? 1.0 : 0.0;
            P.im[t][r] = 0.0;
        }
    double minrel = 1e300;
    block_gj_solve<4>(Q, P, pan, dv, wave, lane, minrel, a.inv_scale2, true);
    double *d = a.Dinv + (size_t)cell * 2 * 4096;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + 4 * r + rg;
            d[row * 64 + col] = P.re[t][r];
            d[4096 + row * 64 + col] = P.im[t][r];
        }
    // |pivot| < 1e-3 b0 (or NaN): the cell is re-solved with partial pivoting by lg_pivoted_kernel
    if (lane == 0 && !(minrel > 1e-6)) a.cellflag[cell] = 1;
}

// Robust fallback of the blocked path: cells flagged by lg_inv64_kernel are re-solved from V and U with
// Gaussian elimination with partial pivoting over the whole NP x NP system (LAPACK gesv semantics of the
// reference), in place in global memory (2 MB per cell, L2 resident), one 1024-thread workgroup per
// flagged cell.  Slow by design: it only runs for cells whose unpivoted block elimination was unsafe.
struct LgPivArgs {
    const double *V, *Uo;   // [ncell][2][NP*NP]
    double *P, *Q;          // [ncell][2][NP*NP]: P <- X = (V-U)^-1 (V+U); Q is scratch
    const int *cellflag;
    int *flags;
    unsigned long long *stats;
    int NP, ncell;
};
__global__ void __launch_bounds__(1024) lg_pivoted_kernel(LgPivArgs a) {
    __shared__ double mre[256], mim[256];
    __shared__ double best[16];
    __shared__ int bestrow[16];
    __shared__ int piv;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NP = a.NP;
    const size_t pp = (size_t)NP * NP;
    for (int cell = blockIdx.x; cell < a.ncell; cell += gridDim.x) {
        if (!a.cellflag[cell]) continue;
        __syncthreads();
        const double *Vr = a.V + (size_t)cell * 2 * pp, *Vi = Vr + pp, *Ur = a.Uo + (size_t)cell * 2 * pp, *Ui = Ur + pp;
        double *Pr = a.P + (size_t)cell * 2 * pp, *Pi = Pr + pp, *Qr = a.Q + (size_t)cell * 2 * pp, *Qi = Qr + pp;
        for (size_t i = tid; i < pp; i += 1024) {
            const double vr = Vr[i], vi = Vi[i], ur = Ur[i], ui = Ui[i];
            Pr[i] = vr + ur; Pi[i] = vi + ui;
            Qr[i] = vr - ur; Qi[i] = vi - ui;
        }
        __syncthreads();
        bool ok = true;
        for (int k = 0; k < NP; ++k) {
            // pivot search down column k (first maximum, like izamax on |re|^2 + |im|^2)
            double b = -1.0; int br = k;
            for (int i = k + tid; i < NP; i += 1024) {
                const double v = Qr[(size_t)i * NP + k] * Qr[(size_t)i * NP + k] + Qi[(size_t)i * NP + k] * Qi[(size_t)i * NP + k];
                if (v > b) { b = v; br = i; }
            }
            for (int off = 32; off >= 1; off >>= 1) {
                const double ob = __shfl_xor(b, off, 64);
                const int orow = __shfl_xor(br, off, 64);
                if (ob > b || (ob == b && orow < br)) { b = ob; br = orow; }
            }
            if (lane == 0) { best[wave] = b; bestrow[wave] = br; }
            __syncthreads();
            if (tid == 0) {
                double bb = best[0]; int rr = bestrow[0];
                for (int w = 1; w < 16; ++w)
                    if (best[w] > bb || (best[w] == bb && bestrow[w] < rr)) { bb = best[w]; rr = bestrow[w]; }
                piv = bb > 0. ? rr : -1;
            }
            __syncthreads();
            const int p = piv;
            if (p < 0) { ok = false; break; }
            if (p != k) {   // swap rows k and p of [Q | P]
                for (int j = tid; j < 2 * NP; j += 1024) {
                    double *re = j < NP ? Qr : Pr, *im = j < NP ? Qi : Pi;
                    const int c = j < NP ? j : j - NP;
                    const double tr_ = re[(size_t)k * NP + c], ti_ = im[(size_t)k * NP + c];
                    re[(size_t)k * NP + c] = re[(size_t)p * NP + c]; im[(size_t)k * NP + c] = im[(size_t)p * NP + c];
                    re[(size_t)p * NP + c] = tr_; im[(size_t)p * NP + c] = ti_;
                }
                __syncthreads();
            }
            {   // multipliers m_i = q_ik / q_kk
                const double pr = Qr[(size_t)k * NP + k], pi = Qi[(size_t)k * NP + k];
                const double inv = 1.0 / (pr * pr + pi * pi);
                for (int i = k + 1 + tid; i < NP; i += 1024) {
                    const double ar = Qr[(size_t)i * NP + k], ai = Qi[(size_t)i * NP + k];
                    mre[i] = (ar * pr + ai * pi) * inv;
                    mim[i] = (ai * pr - ar * pi) * inv;
                }
            }
            __syncthreads();
            // row_i -= m_i row_k over the remaining columns of Q and all columns of P (lanes along columns)
            for (int i = k + 1 + wave; i < NP; i += 16) {
                const double mr = mre[i], mi = mim[i];
                for (int j = k + 1 + lane; j < NP; j += 64) {
                    const double xr = Qr[(size_t)k * NP + j], xi = Qi[(size_t)k * NP + j];
                    Qr[(size_t)i * NP + j] -= mr * xr - mi * xi;
                    Qi[(size_t)i * NP + j] -= mr * xi + mi * xr;
                }
                for (int j = lane; j < NP; j += 64) {
                    const double xr = Pr[(size_t)k * NP + j], xi = Pi[(size_t)k * NP + j];
                    Pr[(size_t)i * NP + j] -= mr * xr - mi * xi;
                    Pi[(size_t)i * NP + j] -= mr * xi + mi * xr;
                }
            }
            __syncthreads();
        }
        if (ok) {
            // back substitution, column oriented: X[k][:] = P[k][:] / q_kk, then P[i][:] -= q_ik X[k][:] for i < k
            for (int k = NP - 1; k >= 0; --k) {
                const double pr = Qr[(size_t)k * NP + k], pi = Qi[(size_t)k * NP + k];
                const double inv = 1.0 / (pr * pr + pi * pi);
                for (int j = tid; j < NP; j += 1024) {
                    const double sr = Pr[(size_t)k * NP + j], si = Pi[(size_t)k * NP + j];
                    Pr[(size_t)k * NP + j] = (sr * pr + si * pi) * inv;
                    Pi[(size_t)k * NP + j] = (si * pr - sr * pi) * inv;
                }
                __syncthreads();
                for (int i = wave; i < k; i += 16) {
                    const double qr = Qr[(size_t)i * NP + k], qi = Qi[(size_t)i * NP + k];
                    for (int j = lane; j < NP; j += 64) {
                        const double xr = Pr[(size_t)k * NP + j], xi = Pi[(size_t)k * NP + j];
                        Pr[(size_t)i * NP + j] -= qr * xr - qi * xi;
                        Pi[(size_t)i * NP + j] -= qr * xi + qi * xr;
                    }
                }
                __syncthreads();
            }
        }
        if (tid == 0) {
            if (!ok) atomicOr(&a.flags[0], 1);   // exactly singular denominator
            stat_add(a.stats, 9, 1ull);
        }
    }
}

// planar chunk result -> U[cell0 + i] (row-major interleaved complex)
// smax_ptr: the store only happens when no cell needed a squaring (otherwise the last squaring launch wrote U)
__global__ void lg_store_u_kernel(const double *X, double2 *U, int NP, size_t ncell_elems, const int *smax_ptr) {
    if (smax_ptr && *smax_ptr != 0) return;
    const size_t pp = (size_t)NP * NP;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < ncell_elems; i += (size_t)gridDim.x * blockDim.x) {
        const size_t cell = i / pp, o = i - cell * pp;
        U[i] = make_double2(X[cell * 2 * pp + o], X[cell * 2 * pp + pp + o]);
    }
}

// the squaring plan of the launch sequence was too short for the counts found on the device: flag the evaluation
// (bit 5) so that the host repeats it with a longer plan
__global__ void lg_plan_check_kernel(int *flags, int cap, const int *smax2) {
    if (smax2 && *smax2 > flags[1]) flags[1] = *smax2;   // (second lane of chunks: the host sizes its next plan from flags[1])
    if (flags[1] > cap) atomicOr(&flags[0], 32);
}

// ---- sweeps for NP in {128, 256}: one 1024-thread workgroup per trajectory ----
template <bool BACKWARD>
__global__ void __launch_bounds__(1024) sweep_lg_kernel(SweepArgs a, int NP) {
    constexpr int NW = 16;
    __shared__ double2 x[2][256];
    __shared__ double2 part[8][256];
    __shared__ double sc[2];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double2 *Uk = a.U + (size_t)(a.cls ? a.cls[k] : k) * a.N_T * NP * NP;
    double2 *st = a.store + (size_t)k * (a.N_T + 1) * NP;
    if (!BACKWARD) {
        if (tid < NP) {
            const double2 v = tid < a.N ? a.psi0[(size_t)k * a.N + tid] : make_double2(0., 0.);
            x[0][tid] = v;
            st[tid] = v;
        }
    } else {
        double2 v = make_double2(0., 0.);
        if (tid < a.N) {
            v = chi_boundary(a, k, tid);
            if (a.xi) {   // chi_k(T) += lambda_b dt/2 xi_k(T)
                const double2 x_ = a.xi[((size_t)k * (a.N_T + 1) + a.N_T) * NP + tid];
                const double c = a.lambda_b * a.wq[a.N_T];
                v.x += c * x_.x; v.y += c * x_.y;
            }
        }
        if (tid < 256) part[0][tid] = make_double2(v.x * v.x + v.y * v.y, 0.);
        __syncthreads();
        if (tid == 0) {
            double n2 = 0.;
            for (int i = 0; i < NP; ++i) n2 += part[0][i].x;
            sc[0] = sqrt(n2);
        }
        __syncthreads();
        const double rho = sc[0];
        if (tid == 0) {
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
    const int CHN = NP / 64;      // column chunks per lane
    const int RW = NP / NW;       // rows per wave (forward)
    int cur = 0;
    for (int step = 0; step < a.N_T; ++step) {
        const int n = BACKWARD ? a.N_T - 1 - step : step;
        const double2 *Un = Uk + (size_t)n * NP * NP;
        if (!BACKWARD) {
            for (int r = 0; r < RW; ++r) {
                const int row = wave * RW + r;
                double pr = 0., pi = 0.;
                for (int cc = 0; cc < CHN; ++cc) {
                    const double2 u = Un[(size_t)row * NP + cc * 64 + lane];
                    const double2 xv = x[cur][cc * 64 + lane];
                    pr += u.x * xv.x - u.y * xv.y;
                    pi += u.x * xv.y + u.y * xv.x;
                }
                pr = wave_sum_dpp(pr);
                pi = wave_sum_dpp(pi);
                if (lane == 0) x[cur ^ 1][row] = make_double2(pr, pi);
            }
            __syncthreads();
            if (tid < NP) st[(size_t)(n + 1) * NP + tid] = x[cur ^ 1][tid];
        } else {
            const int j = tid % NP, q = tid / NP, NQ = 1024 / NP;   // column, row part
            double ar = 0., ai = 0.;
            for (int i = q; i < NP; i += NQ) {
                const double2 u = Un[(size_t)i * NP + j];
                const double2 xi = x[cur][i];
                ar += u.x * xi.x + u.y * xi.y;
                ai += u.x * xi.y - u.y * xi.x;
            }
            part[q][j] = make_double2(ar, ai);
            __syncthreads();
            if (tid < NP) {
                double2 s = part[0][tid];
                for (int qq = 1; qq < NQ; ++qq) { s.x += part[qq][tid].x; s.y += part[qq][tid].y; }
                if (a.xi && n > 0) {   // chi(t_n) += lambda_b Dt_n / rho_k xi_k(t_n)
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
    if (!BACKWARD) {
        if (tid < 256) {
            double pr = 0., pi = 0.;
            if (tid < a.N) {
                const double2 t = a.target[(size_t)k * a.N + tid];
                const double2 p = x[cur][tid];
                pr = t.x * p.x + t.y * p.y;
                pi = t.x * p.y - t.y * p.x;
            }
            part[0][tid] = make_double2(pr, pi);
        }
        __syncthreads();
        if (tid == 0) {
            double sr = 0., si = 0.;
            for (int i = 0; i < NP; ++i) { sr += part[0][i].x; si += part[0][i].y; }
            a.tau[k] = make_double2(sr, si);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Cooperative sweeps for few, large trajectories (K * NP small against the chip: C5 has 8 trajectories
// of N = 256 per GPU, i.e. 8 workgroups on 256 CUs with the kernel above).  S workgroups share one
// trajectory: each owns R = NP / S rows of the state (forward: rows of U_n; backward: columns of U_n,
// i.e. rows of U_n^dagger), reads only its 16 * R * NP bytes of U_n per step and publishes its slice of
// the new state straight into the storage array, which is what the next step of every sibling reads.
// One counter per trajectory orders the steps (agent-scope fences around a relaxed atomic); the U slice
// of the next step is requested before the wait, so the exchange latency overlaps the HBM stream.
// Placement: block b runs on XCD b % 8 (round-robin dispatch), so all siblings of a trajectory are given
// the same b % 8 and meet in one L2.  The grid never exceeds one workgroup per CU: all siblings are
// resident, and the spin is bounded anyway (flag bit 8 -> GRAPE_ERR_HIP) so that the grid always drains.
//   forward : Psi_n     = U_n Psi_{n-1}      (optimize.jl:731-738), tau_k (:753)
//   backward: chi_{n-1} = U_n^dagger chi_n   (optimize.jl:881), boundary :848-868, xi inhomogeneity :897-908
// ---------------------------------------------------------------------------------------
// Slice exchange without whole-cache maintenance: the slices are written and read with agent-scope relaxed
// atomics (write-through / cache-bypassing accesses), a release is then just "my stores are acknowledged"
// (s_waitcnt vmcnt(0)) before the barrier that precedes the counter increment.  Agent-scope FENCES would
// write back and invalidate the whole L2 on every step.
__device__ __forceinline__ void coop_store(double2 *p, double2 v) {
    __hip_atomic_store(&p->x, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&p->y, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double2 coop_load(const double2 *p) {
    double2 v;
    v.x = __hip_atomic_load(&p->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.y = __hip_atomic_load(&p->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}
__device__ __forceinline__ bool coop_wait(unsigned *cnt, unsigned target, int *flags) {
    for (int spin = 0; spin < (1 << 22); ++spin) {
        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    atomicOr(&flags[0], 8);
    return false;
}

// round 6 (SweepArgs::xmode = 1): poll ONE element of the state until it no longer shows the arming pattern (all ones in
// either half: a 16-byte element is published as two 8-byte stores).  Bounded: a sibling that never publishes raises bit 3.
#define COOP_SENTINEL 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ double2 coop_poll(const double2 *p, int *flags, bool &gone) {
    double2 v = make_double2(0., 0.);
    if (gone) return v;
    for (int spin = 0; spin < (1 << 21); ++spin) {
        v = coop_load(p);
        if ((unsigned long long)__double_as_longlong(v.x) != COOP_SENTINEL && (unsigned long long)__double_as_longlong(v.y) != COOP_SENTINEL) return v;
        __builtin_amdgcn_s_sleep(1);
    }
    atomicOr(&flags[0], 8);
    gone = true;
    return make_double2(0., 0.);
}

// a slice element published for siblings on the SAME XCD: past the per-CU vector cache into that XCD's L2 (sc0); the polls
// stay device-scope loads, which that L2 serves while the line is there (grape_cheby.hip.h has the measurement)
__device__ __forceinline__ void coop_store_l2(double2 *p, double2 v) {
    __hip_atomic_store(&p->x, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&p->y, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int CPL, int RPW, int NW, bool BACKWARD>
__global__ void __launch_bounds__(64 * NW) sweep_coop_kernel(SweepArgs a, int S, unsigned *cnt) {
    constexpr int NP = 64 * CPL, T = 64 * NW, R = NW * RPW, E = CPL * RPW, NG = T / R;
    __shared__ double2 x[NP];
    __shared__ double2 part[T > NP ? T : NP];
    __shared__ double sc[2];
    __shared__ int xl;
    bool gone = false;   // (thread 0) a sibling did not arrive within the spin limit
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int k = (slot / S) * 8 + xcd, s = slot % S;
    if (k >= a.K) return;
    if (a.drop_sibling && s == a.drop_sibling - 1) return;   // fault injection (tests): the siblings must time out, not hang
    const int r0 = s * R;   // first row (forward) / column (backward) of this workgroup's slice
    const double2 *Uk = a.U + (size_t)(a.cls ? a.cls[k] : k) * a.N_T * NP * NP;
    double2 *st = a.store + (size_t)k * (a.N_T + 1) * NP;
    unsigned *ck = cnt + k;
    // ---- do all siblings of this trajectory share an XCD (and with it an L2)?  Checked, not assumed. ----
    if (tid == 0) {
        int same = 0;
        if (a.xcc && a.xmode) {
            int *xc = a.xcc + (size_t)k * 32;
            const int mine = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf);   // HW_REG_XCC_ID[3:0]
            __hip_atomic_store(&xc[s], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            same = 1;
            for (int j = 0; j < S; ++j) {
                int v = -1, spin = 0;
                while ((v = __hip_atomic_load(&xc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0) {
                    if (++spin > (1 << 22)) { atomicOr(&a.flags[0], 8); break; }   // a sibling never started
                    __builtin_amdgcn_s_sleep(2);
                }
                same &= (v == mine);
            }
        }
        xl = same;
    }
    __syncthreads();
    const bool l2 = xl != 0;

    // ---- initial state: every sibling forms it redundantly (N elements), sibling 0 stores it ----
    double rho = 1.0;
    if (!BACKWARD) {
        if (tid < NP) {
            const double2 v = tid < a.N ? a.psi0[(size_t)k * a.N + tid] : make_double2(0., 0.);
            x[tid] = v;
            if (s == 0) st[tid] = v;
        }
    } else {
        double2 v = make_double2(0., 0.);
        if (tid < a.N) {
            v = chi_boundary(a, k, tid);
            if (a.xi) {   // chi_k(T) += lambda_b dt/2 xi_k(T)
                const double2 x_ = a.xi[((size_t)k * (a.N_T + 1) + a.N_T) * NP + tid];
                const double c = a.lambda_b * a.wq[a.N_T];
                v.x += c * x_.x; v.y += c * x_.y;
            }
        }
        if (tid < NP) part[tid] = make_double2(v.x * v.x + v.y * v.y, 0.);
        __syncthreads();
        if (tid == 0) {   // same summation order as sweep_lg_kernel
            double n2 = 0.;
            for (int i = 0; i < NP; ++i) n2 += part[i].x;
            sc[0] = sqrt(n2);
        }
        __syncthreads();
        rho = sc[0];
        if (tid == 0 && s == 0) {
            a.rho[k] = rho;
            if (rho < a.chi_min_norm) atomicOr(&a.flags[0], 2);
        }
        if (tid < NP) {
            const double ir = rho > 0. ? 1.0 / rho : 0.;
            v.x *= ir; v.y *= ir;
            x[tid] = v;
            if (s == 0) st[(size_t)a.N_T * NP + tid] = v;
        }
    }
    __syncthreads();

    // thread -> element map (T = 64 NW threads, R = NW * RPW slice rows).  forward: wave w owns rows
    // r0 + w*RPW + r, lane covers columns c*64 + lane; backward: thread (ig = tid / R, jj = tid % R) owns
    // column r0 + jj over rows ig + NG * m
    const int jj = tid % R, ig = tid / R;
    // the U slices of the next TWO steps are in flight (a ring of two register sets with static indices: the step loop is
    // unrolled twice) -- with one the HBM latency of the slice was on the critical path of every step
    double2 u2[2][E];
    auto load_u = [&](double2 (&u)[E], int n) __attribute__((always_inline)) {
        const double2 *Un = Uk + (size_t)n * NP * NP;
        if (!BACKWARD) {
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int c = 0; c < CPL; ++c) u[r * CPL + c] = Un[(size_t)(r0 + wave * RPW + r) * NP + c * 64 + lane];
        } else {
#pragma unroll
            for (int m = 0; m < E; ++m) u[m] = Un[(size_t)(ig + NG * m) * NP + r0 + jj];
        }
    };
    load_u(u2[0], BACKWARD ? a.N_T - 1 : 0);
    if (a.N_T > 1) load_u(u2[1], BACKWARD ? a.N_T - 2 : 1);
    const bool sentinel = a.xmode != 0;
    bool lost = false;   // (any thread, sentinel mode) an element never arrived

    for (int step0 = 0; step0 < a.N_T; step0 += 2) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const int step = step0 + d;
        if (step >= a.N_T) break;
        double2 (&u)[E] = u2[d];
        const int n = BACKWARD ? a.N_T - 1 - step : step;
        const int nout = BACKWARD ? n : n + 1;       // storage row of the new state
        if (step > 0) {
            const int nin = BACKWARD ? n + 1 : n;
            if (sentinel) {
                // the state of the previous step: every element is polled by the thread that needs it in the LDS copy
                if (tid < NP) x[tid] = coop_poll(&st[(size_t)nin * NP + tid], a.flags, lost);
                __syncthreads();
            } else {
                // wait until all S siblings have published step - 1, then fetch the full state
                if (tid == 0 && !gone) gone = !coop_wait(ck, (unsigned)(S * step), a.flags);   // a time-out is final: no further waits
                __syncthreads();
                if (tid < NP) x[tid] = coop_load(&st[(size_t)nin * NP + tid]);
                __syncthreads();
            }
        }
        double2 y = make_double2(0., 0.);
        if (!BACKWARD) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                double pr = 0., pi = 0.;
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const double2 uu = u[r * CPL + c], xv = x[c * 64 + lane];
                    pr += uu.x * xv.x - uu.y * xv.y;
                    pi += uu.x * xv.y + uu.y * xv.x;
                }
                pr = wave_sum_dpp(pr);
                pi = wave_sum_dpp(pi);
                if (lane == 0) {
                    if (l2) coop_store_l2(&st[(size_t)nout * NP + r0 + wave * RPW + r], make_double2(pr, pi));
                    else coop_store(&st[(size_t)nout * NP + r0 + wave * RPW + r], make_double2(pr, pi));
                }
            }
        } else {
            double ar = 0., ai = 0.;
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const double2 uu = u[m], xv = x[ig + NG * m];
                ar += uu.x * xv.x + uu.y * xv.y;
                ai += uu.x * xv.y - uu.y * xv.x;
            }
            part[tid] = make_double2(ar, ai);
            __syncthreads();
            if (tid < R) {
                double2 sum = part[tid];
                for (int q = 1; q < NG; ++q) { sum.x += part[q * R + tid].x; sum.y += part[q * R + tid].y; }
                if (a.xi && n > 0) {   // chi(t_n) += lambda_b Dt_n / rho_k xi_k(t_n)
                    const double2 x_ = a.xi[((size_t)k * (a.N_T + 1) + n) * NP + r0 + tid];
                    const double c = a.lambda_b * a.wq[n] / rho;
                    sum.x += c * x_.x; sum.y += c * x_.y;
                }
                if (l2) coop_store_l2(&st[(size_t)nout * NP + r0 + tid], sum);
                else coop_store(&st[(size_t)nout * NP + r0 + tid], sum);
            }
            (void)y;
        }
        if (!sentinel) {
            __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this thread's slice elements are acknowledged ...
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(ck, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before the count
        } else {
            __syncthreads();                      // (x is rewritten by the next step's poll)
        }
        // request the U slice of the step after next now: it does not depend on the state and streams in during the exchanges
        if (step + 2 < a.N_T) load_u(u2[d], BACKWARD ? n - 2 : n + 2);
      }
    }

    if (!BACKWARD && s == 0) {   // tau_k = <target_k | Psi_k(T)>
        if (!sentinel) {
            if (tid == 0 && !gone) coop_wait(ck, (unsigned)(S * a.N_T), a.flags);
            __syncthreads();
        }
        double pr = 0., pi = 0.;
        if (tid < a.N) {
            const double2 t = a.target[(size_t)k * a.N + tid];
            const double2 p = sentinel ? coop_poll(&st[(size_t)a.N_T * NP + tid], a.flags, lost) : coop_load(&st[(size_t)a.N_T * NP + tid]);
            pr = t.x * p.x + t.y * p.y;
            pi = t.x * p.y - t.y * p.x;
        }
        if (tid < NP) part[tid] = make_double2(pr, pi);
        __syncthreads();
        if (tid == 0) {
            double sr = 0., si = 0.;
            for (int i = 0; i < NP; ++i) { sr += part[i].x; si += part[i].y; }
            a.tau[k] = make_double2(sr, si);
        }
    }
}
