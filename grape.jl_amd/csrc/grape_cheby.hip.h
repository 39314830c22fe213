// grape_cheby.hip.h -- matrix-free polynomial propagator for 64 < N <= 256 (prop_method = GRAPE_PROP_SERIES).
//
// The reference's answer to larger Hilbert spaces is `prop_method = Cheby` (README.md:55, docs/src/tutorial.md:308,
// 432): the state advances by a Chebyshev expansion of exp(-i H_n dt_n) on the VECTOR and no propagator is ever
// formed.  On MI355X that removes the two things that dominate the blocked Pade path of grape_large.hip.h: the O(N^3)
// exponential of every cell (307 of 351 ms per C5-shard evaluation) and the K N_T N^2 16 B of stored propagators
// (16.8 GB per C5 shard).  What is left is a serial chain of matrix-vector products,
//     Hermitian generators (Chebyshev, spectrum of H_n inside [-r_n, r_n], r_n from the norm estimates of grape_create):
//         phi_0 = Psi,  phi_1 = z H_n Psi,  phi_{j+1} = 2 z H_n phi_j + phi_{j-1},  z = -i / r_n  (backward: +i, H_n^dagger)
//         Psi'  = sum_j a_j phi_j,   a_j = (2 - delta_j0) J_j(r_n dt_n)                (Bessel functions, Miller recurrence)
//     other generators (Taylor with sub-steps, the series of grape_series.hip.h with an a-priori term count):
//         t_0 = Psi,  t_{j+1} = (-i dt / (m (j+1))) H_n t_j,  Psi' = sum_j t_j,  m sub-steps
// and the number of terms is known BEFORE the recursion starts (Chebyshev: from the coefficient table; Taylor: from the
// norm bound), so there is no convergence test and no reduction over the vector inside the chain.
//
// A single workgroup cannot hold a 256 x 256 complex generator (1 MB), and the K trajectories of a GPU (8 at C5) would
// leave the chip idle anyway: S = NP / 16 workgroups share one trajectory.  Sibling s owns 16 rows: it forms its
// 16 x NP slice of H_n = H0_k + sum_l eps_nl S_ln H_l in LDS once per time step (64 KB at N = 256), multiplies it with
// the current vector (thread (row, part): NP/16 products, the 16 parts of a row meet in a DPP row reduction) and
// publishes its 16 elements of the new vector.  One exchange per term is what a term costs, so the exchange is built
// for latency: there is NO counter.  The vectors travel through four rotating slots in global memory whose elements
// are armed with a sentinel (a signalling-NaN bit pattern no computation produces); a writer stores its elements with
// agent-scope stores and moves on, a reader polls the very element it needs until the sentinel is gone -- one trip
// through the memory fabric after the data has landed, instead of store-acknowledge + counter increment + counter
// poll + load (5.4 -> 2 us per term).  Re-arming: while it writes exchange e (slot e % 4) a sibling re-arms its rows of
// slot (e + 2) % 4, last used by exchange e - 2, which every sibling has consumed (it published e - 1 after reading
// it); the acknowledgement of those stores is awaited before the data of exchange e + 1 is issued, so whoever sees that
// data polls a slot that is already armed.  Forward and backward sweep run in the same launch on different CUs.
// Coherence level: the siblings of a trajectory are dealt to ONE XCD (blockIdx % 8), whose L2 is where their CUs meet.
// The STORES of the exchange therefore only have to get past the per-CU vector cache (sc0, "work-group scope" in the
// ISA's terms: they land in the XCD's L2 instead of travelling out to the memory fabric); the polls stay device-scope
// loads (sc1), which are served by that L2 while the line is there (measured: 175 -> 141 ms for the C5-shard sweeps;
// polling with sc0 loads behind a buffer_inv sc0 never observed the data and is not used).  The placement is checked,
// not assumed: at start-up every sibling publishes the XCC id of the CU it runs on (device scope), and a group whose
// siblings do not share one XCD uses device-scope stores as well.
//   forward : Psi_n     = exp(-i H_n dt_n) Psi_{n-1}          (optimize.jl:731-738), tau_k (:753)
//   backward: chi_{n-1} = exp(+i H_n^dagger dt_n) chi_n       (optimize.jl:881), boundary :848-868, xi :897-908
// The grid never exceeds one workgroup per CU (the host launches the trajectories in rounds), every spin is bounded.
#pragma once

struct ChebyArgs {
    SweepArgs s;          // boundary data, storage, tau / rho / flags
    const double *H0;     // [K][2][NP*NP] planar row-major: H0f (forward) or H0t (backward: rows of H^T, conjugated here)
    const double *Hc;     // [Kc][L][2][NP*NP]: Hcf or Hct
    const double *eps, *shape, *dts;
    const double *rb;     // [K + Kc*L] 2-norm estimates: r0_k, then r_(kc,l)
    unsigned long long *stats;   // [10] += terms, [11] += (sub-)steps
    double2 *xch;         // [K][4][NP] exchange slots of the recursion vectors, armed with the sentinel at launch
    int *xcc;             // [K][32] XCC id of every sibling's CU (-1 at launch)
    int xmode;            // XCD-local accesses: bit 0 stores, bit 1 loads (0: device scope throughout)
    double tol;           // terms below tol are dropped (1e-17: converged to rounding)
    int L, hc_per_traj, NP, herm, k0, kn;   // this launch covers the trajectories [k0, k0 + kn)
};

#define CHEBY_SENTINEL 0x7FF4DEADBEEFCAFEull   // signalling NaN with a payload no arithmetic produces
#define CHEBY_MAXT 384   // coefficient table (r dt up to ~300 per step; larger steps are cut into sub-steps)
#define CHEBY_RMAX 300.0

// a_j = (2 - delta_j0) J_j(alpha), j = 0..J-1, by Miller's downward recurrence (normalised with J_0 + 2 sum J_2k = 1);
// returns J = number of coefficients kept (the remaining ones are below tol).  One thread.
__device__ inline int cheby_coefficients(double alpha, double tol, double *coef) {
    if (!(alpha > 1e-290)) { coef[0] = 1.0; return 1; }
    int m0 = (int)(alpha * 1.2) + 60;
    if (m0 > CHEBY_MAXT + 40) m0 = CHEBY_MAXT + 40;
    m0 &= ~1;                                  // even start index
    double jp = 0.0, jc = 1e-250, sum = 0.0;   // J_{m0+1}, J_{m0} (unnormalised)
    for (int k = m0; k >= 1; --k) {
        const double jm = (2.0 * k / alpha) * jc - jp;   // J_{k-1}
        jp = jc; jc = jm;
        if (k - 1 < CHEBY_MAXT) coef[k - 1] = jm;
        if (((k - 1) & 1) == 0) sum += (k - 1 == 0 ? 1.0 : 2.0) * jm;
        if (fabs(jc) > 1e200) {   // rescale (everything computed so far is relative)
            const double sc = 1e-200;
            jc *= sc; jp *= sc; sum *= sc;
            for (int i = k - 1; i < CHEBY_MAXT && i <= m0; ++i) coef[i] *= sc;
        }
    }
    const double inv = 1.0 / sum;
    int J = 1;
    const int top = m0 < CHEBY_MAXT ? m0 : CHEBY_MAXT;
    for (int j = 0; j < top; ++j) {
        const double v = coef[j] * inv * (j == 0 ? 1.0 : 2.0);
        coef[j] = v;
        if (fabs(v) > tol) J = j + 1;
    }
    return J;
}

// exchange accesses: L2-coherent inside one XCD (sc0) or device-coherent (sc1)
__device__ __forceinline__ void xch_store(double2 *p, double2 v, bool xcd_local) {
    if (xcd_local) {
        __hip_atomic_store(&p->x, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&p->y, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else coop_store(p, v);
}
__device__ __forceinline__ double2 xch_load(const double2 *p, bool xcd_local) {
    if (!xcd_local) return coop_load(p);
    // the per-CU vector cache may hold the (armed) line from the previous poll: drop it, then read the XCD's L2
    asm volatile("buffer_inv sc0" ::: "memory");
    double2 v;
    v.x = __hip_atomic_load(&p->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    v.y = __hip_atomic_load(&p->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return v;
}

template <bool BACKWARD>
__device__ __forceinline__ void cheby_coop_body(const ChebyArgs &a, const int k, const int s, const int S, double *hs,
                                                double2 *x, double *coef, int *plan) {
    constexpr int T = 256, R = 16, TPR = 16;   // threads, rows per sibling, threads per row
    const int NP = a.NP, LDH = NP + 2, CPT = NP / TPR;   // LDS row stride of the slice, columns per thread
    const SweepArgs &sa = a.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int row = tid / TPR, part = tid % TPR;
    const int N_T = sa.N_T, L = a.L;
    const int r0 = s * R;
    double2 *st = sa.store + (size_t)k * (N_T + 1) * NP;
    double2 *xk = a.xch + (size_t)k * 4 * NP;
    unsigned epoch = 0;   // exchanges completed by this trajectory's group
    __shared__ double sc[2];
    __shared__ int xl;
    // ---- do all siblings of this trajectory share an XCD (and with it an L2)? ----
    if (tid == 0) {
        int *xc = a.xcc + (size_t)k * 32;
        const int mine = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf);   // HW_REG_XCC_ID[3:0]
        __hip_atomic_store(&xc[s], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int same = 1;
        for (int j = 0; j < S; ++j) {
            int v = -1, spin = 0;
            while ((v = __hip_atomic_load(&xc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < 0) {
                if (++spin > (1 << 22)) { atomicOr(&sa.flags[0], 8); break; }   // a sibling never started
                __builtin_amdgcn_s_sleep(2);
            }
            same &= (v == mine);
        }
        xl = same;
    }
    __syncthreads();
    const bool xcd_st = xl != 0 && (a.xmode & 1), xcd_ld = xl != 0 && (a.xmode & 2);
    bool dead = false;   // a sibling did not answer within the spin limit: flag 8 is up, nobody waits any more
    double *hre = hs, *him = hs + R * LDH;
    const double *h0 = a.H0 + (size_t)k * 2 * NP * NP;
    const double *hc = a.Hc + (size_t)(a.hc_per_traj ? k : 0) * L * 2 * NP * NP;
    const double r0k = a.rb[k];
    const double *rl = a.rb + sa.K + (size_t)(a.hc_per_traj ? k : 0) * L;
    constexpr double SG = BACKWARD ? -1.0 : 1.0;   // backward: H^dagger = conj of the transposed planes, and +i dt

    // ---- boundary state: every sibling forms it redundantly, sibling 0 stores it ----
    double rho = 1.0;
    if (!BACKWARD) {
        for (int i = tid; i < NP; i += T) {
            const double2 v = i < sa.N ? sa.psi0[(size_t)k * sa.N + i] : make_double2(0., 0.);
            x[i] = v;
            if (s == 0) st[i] = v;
        }
    } else {
        __shared__ double part2[T];
        double n2 = 0.;
        for (int i = tid; i < NP; i += T) {
            double2 v = make_double2(0., 0.);
            if (i < sa.N) {
                v = chi_boundary(sa, k, i);
                if (sa.xi) {   // chi_k(T) += lambda_b dt/2 xi_k(T)   (optimize.jl:856-866)
                    const double2 x_ = sa.xi[((size_t)k * (N_T + 1) + N_T) * NP + i];
                    const double c = sa.lambda_b * sa.wq[N_T];
                    v.x += c * x_.x; v.y += c * x_.y;
                }
            }
            x[i] = v;
            n2 += v.x * v.x + v.y * v.y;
        }
        part2[tid] = n2;
        __syncthreads();
        if (tid == 0) {
            double t = 0.;
            for (int i = 0; i < T; ++i) t += part2[i];
            sc[0] = sqrt(t);
        }
        __syncthreads();
        rho = sc[0];
        if (tid == 0 && s == 0 && !sa.unit_chi) {
            sa.rho[k] = rho;
            if (rho < sa.chi_min_norm) atomicOr(&sa.flags[0], 2);
        }
        const double ir = rho > 0. ? 1.0 / rho : 0.;
        for (int i = tid; i < NP; i += T) {
            double2 v = x[i];
            v.x *= ir; v.y *= ir;
            x[i] = v;
            if (s == 0) st[(size_t)N_T * NP + i] = v;
        }
    }
    __syncthreads();

    // y_row = sum_j Hs[row][j] v[j] over this thread's columns part, part + 16, ..; the 16 parts of a row are the 16
    // lanes of a DPP row
    auto matvec = [&](double &yr, double &yi) __attribute__((always_inline)) {
        double pr = 0., pi = 0.;
        const double *hr_ = hre + row * LDH + part, *hi_ = him + row * LDH + part;
        for (int c = 0; c < CPT; ++c) {
            const double mr = hr_[c * TPR], mi = hi_[c * TPR];
            const double2 v = x[part + c * TPR];
            pr = fma(mr, v.x, pr); pr = fma(-mi, v.y, pr);
            pi = fma(mr, v.y, pi); pi = fma(mi, v.x, pi);
        }
        yr = group_sum<16>(pr);
        yi = group_sum<16>(pi);
    };
    // publish this sibling's 16 elements of a vector in the current exchange slot (and, plainly, in the storage row
    // `dst`), re-arm the slot after next, then poll the whole vector into x
    auto exchange = [&](double2 *dst, const double vr, const double vi) __attribute__((always_inline)) {
        double2 *buf = xk + (size_t)(epoch & 3) * NP, *rearm = xk + (size_t)((epoch + 2) & 3) * NP;
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the re-arming stores of the previous exchange are acknowledged
        if (part == 0) {
            xch_store(&buf[r0 + row], make_double2(vr, vi), xcd_st);
            const double sn = __longlong_as_double((long long)CHEBY_SENTINEL);
            xch_store(&rearm[r0 + row], make_double2(sn, sn), xcd_st);
            if (dst) dst[r0 + row] = make_double2(vr, vi);
        }
        ++epoch;
        __syncthreads();   // everybody is done reading x (the products of this term)
        for (int i = tid; i < NP; i += T) {
            double2 v;
            int spin = 0;
            for (;;) {
                v = xch_load(&buf[i], xcd_ld);
                if (__double_as_longlong(v.x) != (long long)CHEBY_SENTINEL &&
                    __double_as_longlong(v.y) != (long long)CHEBY_SENTINEL) break;
                if (dead || ++spin > (1 << 19)) {   // a sibling is missing: give up for good (the evaluation fails with flag 8)
                    atomicOr(&sa.flags[0], 8);
                    dead = true;
                    v = make_double2(0., 0.);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            x[i] = v;
        }
        if (!dead && (epoch & 63) == 0) dead = (__hip_atomic_load(&sa.flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 8) != 0;
        dead = __syncthreads_or(dead);
    };

    unsigned long long terms = 0, steps = 0;
    for (int step = 0; step < N_T; ++step) {
        const int n = BACKWARD ? N_T - 1 - step : step;
        const int nout = BACKWARD ? n : n + 1;
        const double dt = a.dts[n];
        // ---- H_n slice -> LDS (rows r0 .. r0+15), spectral bound, coefficient table ----
        double e[8], bound = r0k;
        for (int l = 0; l < L; ++l) {
            e[l] = a.eps[(size_t)l * N_T + n] * (a.shape ? a.shape[(size_t)l * N_T + n] : 1.0);
            bound += fabs(e[l]) * rl[l];
        }
        for (int idx = tid; idx < R * NP; idx += T) {
            const int i = idx / NP, j = idx - i * NP;
            const size_t g = (size_t)(r0 + i) * NP + j;
            double vr = h0[g], vi = h0[(size_t)NP * NP + g];
            for (int l = 0; l < L; ++l) {
                vr = fma(e[l], hc[(size_t)l * 2 * NP * NP + g], vr);
                vi = fma(e[l], hc[(size_t)l * 2 * NP * NP + (size_t)NP * NP + g], vi);
            }
            hre[i * LDH + j] = vr;
            him[i * LDH + j] = SG * vi;
        }
        if (tid == 0) {
            int nsub = 1, J;
            if (a.herm) {
                const double alpha_full = bound * dt;
                nsub = (int)ceil(alpha_full / CHEBY_RMAX);
                if (nsub < 1) nsub = 1;
                J = cheby_coefficients(alpha_full / nsub, a.tol, coef);
            } else {
                // Taylor: sub-steps of norm <= 3, term count from alpha^J / J! < tol
                nsub = (int)ceil(bound * dt / 3.0);
                if (nsub < 1) nsub = 1;
                const double al = bound * dt / nsub;
                double t = 1.0;
                J = 1;
                while (J < CHEBY_MAXT - 1) { t *= al / J; if (t < a.tol && J >= 2) break; ++J; }
                ++J;
            }
            plan[0] = nsub; plan[1] = J;
        }
        __syncthreads();
        const int nsub = plan[0], J = plan[1];
        const double dts_ = dt / nsub;
        for (int sub = 0; sub < nsub; ++sub) {
            // x holds the full current state; this thread's row element of it:
            const double2 x0 = x[r0 + row];
            double accr, acci;
            if (a.herm) {
                // z = -/+ i / r:  z (a + i b) = (+/- b - -/+ ... ) written out below; zr = 0, zi = -SG / bound
                const double zi = bound > 0. ? -SG / bound : 0.;
                double pr_ = x0.x, pi_ = x0.y;           // phi_{j-1}
                accr = coef[0] * x0.x; acci = coef[0] * x0.y;
                double cr_ = 0., ci_ = 0.;               // phi_j
                if (J > 1) {
                    double yr, yi;
                    matvec(yr, yi);
                    cr_ = -zi * yi; ci_ = zi * yr;       // phi_1 = z H Psi, z = i zi
                    accr = fma(coef[1], cr_, accr); acci = fma(coef[1], ci_, acci);
                }
                for (int j = 2; j < J; ++j) {
                    exchange(nullptr, cr_, ci_);   // everybody needs phi_{j-1} in full
                    double yr, yi;
                    matvec(yr, yi);
                    const double nr = fma(-2.0 * zi, yi, pr_), ni = fma(2.0 * zi, yr, pi_);   // 2 z H phi_{j-1} + phi_{j-2}
                    pr_ = cr_; pi_ = ci_;
                    cr_ = nr; ci_ = ni;
                    accr = fma(coef[j], cr_, accr); acci = fma(coef[j], ci_, acci);
                }
            } else {
                double tr = x0.x, ti = x0.y;
                accr = tr; acci = ti;
                for (int j = 1; j < J; ++j) {
                    if (j > 1) exchange(nullptr, tr, ti);
                    double yr, yi;
                    matvec(yr, yi);
                    const double f = -SG * dts_ / (double)j;   // (-/+ i dt / j) (a + i b)
                    tr = -f * yi; ti = f * yr;
                    accr += tr; acci += ti;
                }
            }
            terms += (unsigned long long)J;
            ++steps;
            const bool last = sub == nsub - 1;
            if (last && BACKWARD && sa.xi && n > 0) {   // chi(t_n) += lambda_b Dt_n / rho_k xi_k(t_n)   (optimize.jl:897-908)
                const double2 x_ = sa.xi[((size_t)k * (N_T + 1) + n) * NP + r0 + row];
                const double c = sa.lambda_b * sa.wq[n] / rho;
                accr += c * x_.x; acci += c * x_.y;
            }
            // the new state: the stored one after the last sub-step (every sibling reads it back in full)
            exchange(last ? st + (size_t)nout * NP : nullptr, accr, acci);   // (storage row: for the later kernels)
        }
    }
    if (tid == 0 && s == 0) {
        stat_add(a.stats, 10, terms);
        stat_add(a.stats, 11, steps);
    }
    if (!BACKWARD && s == 0) {   // tau_k = <target_k | Psi_k(T)>   (x holds Psi_k(T))
        __shared__ double2 part3[T];
        double pr = 0., pi = 0.;
        for (int i = tid; i < sa.N; i += T) {
            const double2 t = sa.target[(size_t)k * sa.N + i], p = x[i];
            pr += t.x * p.x + t.y * p.y;
            pi += t.x * p.y - t.y * p.x;
        }
        part3[tid] = make_double2(pr, pi);
        __syncthreads();
        if (tid == 0) {
            double sr = 0., si = 0.;
            for (int i = 0; i < T; ++i) { sr += part3[i].x; si += part3[i].y; }
            sa.tau[k] = make_double2(sr, si);
        }
    }
}

// arms every element of the exchange slots with the sentinel (ordinary stores: the kernel boundary publishes them)
__global__ void cheby_arm_kernel(unsigned long long *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = CHEBY_SENTINEL;
}

// blocks [0, nb): forward sweeps, [nb, 2 nb): backward sweeps (ab.kn == 0: forward only); within a direction block b
// serves trajectory k0 + (slot / S) * 8 + (b % 8), sibling slot % S (b % 8 = XCD: the siblings of a trajectory share an L2)
__global__ void __launch_bounds__(256) cheby_coop_kernel(ChebyArgs af, ChebyArgs ab, int S, int nb) {
    extern __shared__ __attribute__((aligned(16))) double csm[];
    const int NP = af.NP;
    double *hs = csm;                                   // [2][16][NP + 2]
    double2 *x = (double2 *)(hs + 2 * 16 * (NP + 2));   // [NP]
    double *coef = (double *)(x + NP);                  // [CHEBY_MAXT]
    int *plan = (int *)(coef + CHEBY_MAXT);             // [2]
    const bool bw = (int)blockIdx.x >= nb;
    const int b = bw ? blockIdx.x - nb : blockIdx.x;
    const ChebyArgs &a = bw ? ab : af;
    const int xcd = b & 7, slot = b >> 3;
    const int kl = (slot / S) * 8 + xcd, s = slot % S;
    if (kl >= a.kn) return;
    if (a.s.drop_sibling && s == a.s.drop_sibling - 1) return;   // fault injection (tests): the siblings must time out, not hang
    if (bw) cheby_coop_body<true>(ab, ab.k0 + kl, s, S, hs, x, coef, plan);
    else cheby_coop_body<false>(af, af.k0 + kl, s, S, hs, x, coef, plan);
}
