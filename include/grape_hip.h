/*
 * grape_hip.h -- C ABI of the MI355X-native GRAPE gradient evaluator.
 *
 * This is the drop-in boundary for the hot path of JuliaQuantumControl/GRAPE.jl: one call of
 * grape_eval() replaces the body of
 *     evaluate_functional(pulsevals, wrk)        /root/reference/src/optimize.jl:696-768
 *     evaluate_gradient!(G, pulsevals, wrk)      /root/reference/src/optimize.jl:824-1014
 * i.e. what the closure fg!(F, G, pulsevals) (src/optimize.jl:105-111) does when the optimizer
 * backend calls it (ext/GRAPELBFGSBExt.jl:99, ext/GRAPEOptimExt.jl:31), for trajectories whose
 * generator is H_k(t) = H0_k + sum_l eps_l(t) S_l(t) H_l propagated with `prop_method = ExpProp`.
 * grape_create() replaces the data-layout half of the GrapeWrk constructor
 * (/root/reference/src/workspace.jl:147-362): it uploads the static problem and allocates the
 * forward storage (workspace.jl:215), tau_grads (workspace.jl:236-237) and gradient buffers
 * (workspace.jl:196-198) in HBM.
 *
 * Conventions
 *   - plain C, no exceptions, no torch types; every function returns 0 on success or a negative
 *     grape_status; the message of the last failure is available from grape_last_error().  Every entry point is an
 *     exception barrier: a C++ exception raised by the host side (std::bad_alloc of a staging vector, std::system_error
 *     of a shard thread) is caught inside the library, everything the call had allocated is released, and the caller
 *     (a Julia process behind `ccall`) sees GRAPE_ERR_HOST with the exception's message -- never std::terminate.
 *   - complex numbers are interleaved (re, im) doubles == Julia ComplexF64 == C double _Complex.
 *   - matrices are COLUMN-major N x N (Julia Matrix{ComplexF64}); states are length-N vectors.
 *   - pulsevals / G are CONTROL-major: index (l * N_T + n), l < L, n < N_T
 *     (workspace.jl:159-162, optimize.jl:579, 935-936).
 *   - host pointers are owned by the caller and are not retained after the call returns
 *     (they may be Julia-GC managed: `obj.g`, `wrk.pulsevals`).  One in-flight call per handle.
 *   - trajectories may be sharded over processes/GPUs: a handle owns K local trajectories out of
 *     K_total; the two cross-trajectory reductions of the path (sum_k w_k tau_k for J_T_sm/chi_sm,
 *     sum_k of the gradient, optimize.jl:579) are exposed by the split-phase calls below so that
 *     the host can all-reduce them (RCCL) between phases.
 *   - several GPUs behind ONE handle (ABI v4, grape_problem.ndev / .devices): the K trajectories of the
 *     handle are dealt to the devices in contiguous blocks, every host-pointer entry point enqueues the work
 *     of all of them (one host thread per device, asynchronous launches on one stream per device; the
 *     calling thread waits and reads back) and performs the two reductions itself -- RCCL all-reduces on the
 *     shard streams (one communicator rank per device, ncclCommInitAll) when every shard has a device of its
 *     own, otherwise staged through pinned host memory in shard order -- the caller never sees the devices,
 *     exactly like the transparent `@threadsif` loops over k of optimize.jl:720, 876.
 */
#ifndef GRAPE_HIP_H
#define GRAPE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRAPE_HIP_ABI_VERSION 6

typedef struct grape_handle grape_handle;

typedef enum {
    GRAPE_OK = 0,
    GRAPE_ERR_INVALID = -1,      /* bad argument / unsupported size                          */
    GRAPE_ERR_HIP = -2,          /* HIP runtime failure (message has the hipError string)    */
    GRAPE_ERR_CHI_NORM = -3,     /* ||chi_k|| < chi_min_norm          (optimize.jl:1021-1025)*/
    GRAPE_ERR_SINGULAR = -4,     /* Pade denominator numerically singular                    */
    GRAPE_ERR_TAYLOR = -5,       /* Taylor derivative series did not converge (optimize.jl:644-648) */
    GRAPE_ERR_NO_CONTROLS = -6,  /* L == 0                             (workspace.jl:155-157)*/
    GRAPE_ERR_AGAIN = -7,        /* device-pointer API, N > 64 only: the asynchronous launch plan (number of squaring
                                    launches) was too short for this evaluation and has been adapted -- repeat the call.
                                    The host-pointer entry points repeat internally and never return this.          */
    GRAPE_ERR_HOST = -8          /* ABI v6: a host-side C++ exception (out of memory, thread creation) was caught at the
                                    boundary; nothing of the call is left allocated, the handle (if any) stays usable */
} grape_status;

/* J_T of QuantumControl.Functionals (docs/src/tutorial.md:349-356, 402) */
typedef enum {
    GRAPE_J_T_SM = 0,  /* 1 - |sum_k w_k tau_k|^2 / K^2 ; chi_k = w_k (sum_j w_j tau_j) / K^2 * tgt_k */
    GRAPE_J_T_SS = 1,  /* 1 - sum_k w_k |tau_k|^2 / K   ; chi_k = w_k tau_k / K * tgt_k               */
    GRAPE_J_T_RE = 2   /* 1 - Re sum_k w_k tau_k / K    ; chi_k = w_k / (2K) * tgt_k                  */
} grape_functional;

/* gradient_method keyword of the reference (workspace.jl:150, optimize.jl:871-998).
 * GRAPE_GRAD_GRADGEN: the contraction <chi| D exp(-i H dt)[-i dt mu_l] |Psi> to rounding (what the exponential of the gradient
 * generator delivers in the reference), by a two-pass polynomial series on the stored states: the Taylor sum cut at 1e-16,
 * sub-stepped for long steps -- and, for Hermitian cells whose spectrum this evaluation's exponential kernel has certified
 * to lie in a segment of the imaginary axis, the economized polynomial of that segment (same accuracy, fewer orders;
 * DESIGN.md 4.3, tools/econ_coeffs.py; environment GRAPE_DERIV_ECON=0 keeps the Taylor sum everywhere). */
typedef enum {
    GRAPE_GRAD_GRADGEN = 0, /* exact derivative of exp (what the gradient generator yields)  */
    GRAPE_GRAD_TAYLOR = 1   /* Kuprov-Rodgers recursion, taylor_grad_step! (optimize.jl:604) */
} grape_gradient_method;

/* prop_method keyword of the reference (workspace.jl:222-232 -> QuantumPropagators.init_prop) */
typedef enum {
    GRAPE_PROP_EXP = 0,    /* ExpProp: U_n = exp(-i H_n dt_n) materialised on MFMA by an inverse-free polynomial with scaling
                              and squaring: Hermitian generators, 16 < N <= 64 -- degree 16 in four products for the cells
                              whose spectral radius a per-cell bound proves <= 1.36, degree 18 in five products (Chebyshev
                              coefficients, spectral scaling) for the others and for every other size; general matrices
                              -- degree-18 Taylor polynomial.  GRAPE_EXPM_T16=0: five products everywhere;
                              GRAPE_EXPM_T18=0: the order-13 Pade approximant as in Julia's exp! (parity reference).
                              Propagators that do not fit the device make the handle evaluate matrix-free
                              (grape_get_work[12]).                                                                */
    GRAPE_PROP_SERIES = 1  /* matrix-free polynomial propagator on the state vector (the role of the reference's
                              Cheby / Newton methods, README.md:55), no U: N <= 64 power series of exp(-i H_n dt_n) Psi
                              summed to prop_tolerance, O(N^2) per term; 64 < N <= 512 cooperative Chebyshev sweeps
                              (Hermitian generators, guaranteed spectral interval) or Taylor sub-steps              */
} grape_prop_method;

typedef struct {
    int32_t abi_version;     /* GRAPE_HIP_ABI_VERSION                                           */
    int32_t N;               /* Hilbert-space dimension                                         */
    int32_t L;               /* number of controls                                              */
    int32_t K;               /* trajectories owned by this handle (the reference's `N`, :703)   */
    int32_t K_total;         /* trajectories of the whole job (== K when not sharded; 0 => K)   */
    int32_t N_T;             /* time intervals = length(tlist) - 1 (optimize.jl:705)            */
    int32_t functional;      /* grape_functional                                                */
    int32_t gradient_method; /* grape_gradient_method                                           */
    int32_t hc_per_traj;     /* 0: Hc is [L][N*N] shared by all k; 1: Hc is [K][L][N*N]         */
    int32_t device;          /* HIP device ordinal                                              */
    const double *tlist;     /* [N_T+1] time grid (non-uniform allowed)                         */
    const double *H0;        /* [K][N*N] complex: drift of trajectory k                         */
    const double *Hc;        /* control operators mu_l = dH/d eps_l (see hc_per_traj)           */
    const double *shape;     /* NULL or [L][N_T] real: a_l(eps, t_n) = shape[l][n] * eps_{nl}   */
    const double *psi0;      /* [K][N] complex initial states                                   */
    const double *target;    /* [K][N] complex target states.  ABI v6: NULL = trajectories without a target_state
                                (optimize.jl:753: tau_k = NaN) -- legal only with a caller-side J_T / chi pair, i.e.
                                grape_forward + grape_get_final_states + grape_backward_chi / _xi(chi != NULL);
                                grape_eval, grape_backward and the built-in functionals need the targets and refuse */
    const double *weights;   /* NULL (all 1) or [K]                                             */
    double chi_min_norm;     /* <= 0 selects the reference default 1e-100 (optimize.jl:846)     */
    int32_t taylor_max_order;/* <= 0 selects 100   (optimize.jl:915)                            */
    double taylor_tolerance; /* <= 0 selects 1e-16 (optimize.jl:916)                            */
    /* State-dependent running cost of the family g_b(Psi) = <Psi|D|Psi> with xi = -D Psi
     * (optimize.jl:727-750, 764-766, 856-866, 897-908; test/test_state_running_cost.jl:32-40):
     * J gains lambda_b * sum_k trapezoid_n g_b(Psi_k(t_n)), chi the inhomogeneity of
     * docs/src/background.md:771-775.  Dpen == NULL or lambda_b == 0 switches it off.          */
    const double *Dpen;      /* NULL, [N*N] (shared) or [K][N*N] complex Hermitian, column-major */
    int32_t dpen_per_traj;   /* 0: one D for all trajectories; 1: one per trajectory            */
    double lambda_b;
    int32_t prop_method;     /* grape_prop_method (ABI v3)                                      */
    double prop_tolerance;   /* GRAPE_PROP_SERIES: stop at ||term|| < tol ||state||; <= 0 selects 1e-17 */
    /* ABI v4: GPUs behind one handle.  ndev <= 1: the single device `device`.  ndev > 1: trajectory
     * block g (contiguous, sizes differ by at most one) lives on devices[g]; devices == NULL selects
     * device, device+1, ...  An ordinal may repeat (several shards on one GPU).                  */
    int32_t ndev;
    const int32_t *devices;  /* NULL or [ndev] HIP device ordinals                              */
    /* ABI v6: taylor_grad_check_convergence (optimize.jl:917-918, taylor_grad_step! :611, :631-651).  0 = the reference
     * default `true`: gradient_method = GRAPE_GRAD_TAYLOR raises GRAPE_ERR_TAYLOR when a series has not reached
     * taylor_tolerance within taylor_max_order terms.  1 = `check_convergence = false`: the series is cut at
     * taylor_max_order terms and NO error is raised (terms below taylor_tolerance are still skipped: each of them changes
     * the result by less than the tolerance; the series kernels of N > 32 hold at most 64 terms -- a larger
     * taylor_max_order that is actually reached there is reported, not silently cut).  Ignored by GRAPE_GRAD_GRADGEN. */
    int32_t taylor_no_check;
} grape_problem;

/* Replaces GrapeWrk(...) data set-up: /root/reference/src/workspace.jl:147-362 */
int grape_create(grape_handle **out, const grape_problem *problem);
void grape_destroy(grape_handle *h);

/*
 * Replaces fg!(F, G, x): /root/reference/src/optimize.jl:105-111.
 *   G == NULL  -> evaluate_functional only (forward sweep, optimize.jl:696-768)
 *   G != NULL  -> evaluate_gradient!      (optimize.jl:824-1014); G has L*N_T doubles
 *   tau  (nullable) [K] complex  -> wrk.result.tau_vals      (optimize.jl:753)
 *   psiT (nullable) [K][N] complex -> fw_propagators[k].state (optimize.jl:187-189, 752)
 * Single-handle form: valid only when K == K_total.
 */
int grape_eval(grape_handle *h, const double *pulsevals, double *J, double *G, double *tau,
               double *psiT);

/*
 * Split-phase form for trajectory shards (one handle per GPU):
 *   1. grape_forward      : expm of every local cell + forward sweep; returns local tau[K]
 *   2. host all-reduces   : f = sum over ALL trajectories of w_k tau_k   (2 doubles)
 *   3. grape_backward     : chi from (f, local tau), backward sweep, per-cell derivatives and the
 *                           local sum over k; returns the PARTIAL gradient (L*N_T) and the local
 *                           partial sums needed for J ([2]=sum w|tau|^2, [3]=Re sum w tau, [4]=sum J_b)
 *   4. host all-reduces   : G (sum) -- the sum over k of optimize.jl:579
 * grape_eval() is exactly 1 + 3 with f computed locally.
 */
int grape_forward(grape_handle *h, const double *pulsevals, double *tau /* [K] complex */);
int grape_backward(grape_handle *h, const double f_total[2], double *G_partial);

/* The partial sums of this handle's trajectories after grape_forward -- what the host all-reduces in step 2 and what
 * every J_T and the state running cost need (J_parts[1], J_parts[3], optimize.jl:757-766):
 *   sums[0..1] = f = sum_k w_k tau_k, [2] = sum_k w_k |tau_k|^2, [3] = Re sum_k w_k tau_k,
 *   [4] = sum_k J_b,k (trapezoid sum of g_b, optimize.jl:727-750; multiply by lambda_b), [5..7] = 0. */
int grape_get_sums(grape_handle *h, double sums[8]);

/* Final states Psi_k(T) of the last forward sweep ([K][N] complex): `fw_propagators[k].state`, the argument of a
 * user-supplied J_T / chi (optimize.jl:752-760, 849-855). */
int grape_get_final_states(grape_handle *h, double *psiT);

/*
 * Backward half with HOST-SUPPLIED boundary states: chi[k] = chi_k(T) = -d J_T / d <Psi_k(T)| exactly as the user's
 * `chi(Psi, trajectories; tau)` returns them (optimize.jl:845-855; default constructor workspace.jl:306-308) --
 * [K][N] complex, NOT normalised.  The library adds the xi(T) term of the state running cost (:856-866), forms
 * rho_k = ||chi_k||, applies the chi_min_norm guard (:1021-1025), normalises (:867-868) and runs the backward sweep,
 * the per-cell derivatives and the sum over k.  Together with grape_forward + grape_get_final_states this keeps an
 * arbitrary J_T / chi pair on the caller's side: the three built-in functionals are only a fast path.
 * Returns the (partial, if sharded) gradient in G [L*N_T].
 */
int grape_backward_chi(grape_handle *h, const double *chi, double *G);

/* ABI v5.  An ARBITRARY state running cost g_b (optimize.jl:727-750, 856-866, 897-908: the reference calls the user's
 * g_b(state, trajectory, tlist, n) and xi(state, trajectory, tlist, n) = -d g_b / d<Psi| inside its loops; callbacks cannot
 * cross a C ABI, so the data does): after grape_forward the caller reads the stored forward states
 * (grape_get_storage(0): Psi_k(t_n), [K][N_T+1][N]), evaluates g_b and xi on them, and hands back
 *   xi       : [K][N_T+1][N] complex, xi_k(t_n) (entry n = 0 is not used, as in the reference),
 *   lambda_b : the weight of J_b in J.
 * The library adds lambda_b dt/2 xi_k(T) to chi_k(T) (optimize.jl:856-866) and lambda_b Dt_n / rho_k xi_k(t_n) behind
 * every backward step (:897-908, trapezoid weights of :727-750) and returns the gradient of J_T + lambda_b J_b.  chi:
 * [K][N] boundary states of a user-defined J_T as in grape_backward_chi, or NULL for the handle's built-in functional
 * (then f_total = all-reduced sum_k w_k tau_k as in grape_backward).  J_b itself is the caller's trapezoid sum of g_b.
 * (The built-in family g_b = <Psi|D|Psi> of grape_problem.Dpen stays the device-side fast path.) */
int grape_backward_xi(grape_handle *h, const double f_total[2], const double *chi, const double *xi, double lambda_b,
                      double *G);

/* Device-resident variants used by the bench / RCCL path: same semantics, every pointer is a
 * device pointer on the handle's device; work is enqueued on `stream` (a hipStream_t passed as
 * void*) without host synchronisation (single-device handles only).
 *   d_out layout of grape_forward_device (2K + 8 doubles): [0..2K) tau, then the shard sums
 *        [2K] Re f, [2K+1] Im f (f = sum_k w_k tau_k), [2K+2] sum_k w_k |tau_k|^2,
 *        [2K+3] Re sum_k w_k tau_k, [2K+4] sum_k J_b,k (state running cost), [2K+5..2K+7] 0
 *   d_f   : 2 doubles, the all-reduced f;  d_G: L*N_T doubles (partial gradient, overwritten) */
int grape_forward_device(grape_handle *h, const double *d_pulsevals, double *d_out, void *stream);
int grape_backward_device(grape_handle *h, const double *d_f, double *d_G, void *stream);

/* Concurrent sweeps.  With one workgroup per trajectory a sweep occupies K of the 256 CUs, and the backward
 * recursion chi_{n-1} = U_n^dagger chi_n (optimize.jl:881) is linear in chi; so, unless the state running cost adds its
 * inhomogeneity, grape_forward[_device] also runs the backward sweep, from the unit targets, in the same launch, and
 * grape_backward[_device] only applies the boundary coefficient of chi_k(T) = c_k target_k (docs/src/tutorial.md:402)
 * to the overlaps.  Results are identical to rounding.  on = 0 restores the sequential order (a caller that only wants
 * tau from the split-phase API should do that; grape_eval with G == NULL never runs the backward sweep).  Returns 1 if
 * the concurrent path is active for this handle afterwards, 0 if not.  Default: on (env GRAPE_FUSED_SWEEPS=0: off). */
int grape_set_fused_sweeps(grape_handle *h, int on);

/* Synchronise `stream` and translate the device-side error flags of the evaluation in flight
 * (singular Pade denominator, chi norm guard, series non-convergence) into a grape_status. */
int grape_check(grape_handle *h, void *stream);

/* U_kn = exp(-i H_kn dt_n) of the last evaluation as an N x N column-major complex matrix
 * (parity check of the ExpProp step, optimize.jl:732). */
int grape_get_propagator(grape_handle *h, int k, int n, double *out);

/* Optional outputs of the last evaluation (debug / parity): tau_grads[k][l][n] complex
 * (workspace.jl:236-237, value of optimize.jl:894), forward storage [k][n][N] complex
 * (workspace.jl:215; n = 0..N_T) and the backward states chi_k(t_n) [k][n][N]. */
int grape_get_tau_grads(grape_handle *h, double *out /* K*L*N_T complex */);
int grape_get_storage(grape_handle *h, int which /*0 fw, 1 bw*/, double *out /* K*(N_T+1)*N complex */);

/* Per-phase device time in milliseconds, measured with HIP events recorded on the stream the
 * kernels were launched on, AVERAGED over the evaluations since the last grape_reset_timings
 * (at most the 64 most recent): [0] expm kernel, [1] forward sweep, [2] backward sweep,
 * [3] cell derivatives, [4] reduction, [5] whole grape_eval; handles with several devices (ndev > 1) also [6] the host
 * wall time of the enqueue halves per evaluation (every shard is enqueued from its own host thread) and [7] the latency of
 * the RCCL all-reduce of the gradient across the shards (HIP events on the first shard's stream; -1 when the reductions
 * are host-staged: repeated device ordinals, GRAPE_MULTI_RCCL=0, RCCL not loadable).  Synchronises the device.
 * Returns the number of entries written. */
int grape_get_timings(grape_handle *h, double *ms, int n);
int grape_reset_timings(grape_handle *h);
/* Algorithmic work of the last evaluation: [0] cells, [1] sum of squarings s over cells,
 * [2] flop of the expm kernel (SURVEY 8d F_exp), [3] flop of the derivative kernel, [4] derivative series orders,
 * [5] cells solved by the pivoted fallback, [6] propagators exponentiated, [7] terms and [8] (sub-)steps of the
 * matrix-free propagator, [9] flop of the matrix instructions EXECUTED by the inverse-free exponential of Hermitian
 * generators (grape_t18.hip.h), [10] its squarings, [11] its cells, [12] 1 when prop_method = GRAPE_PROP_EXP was asked for
 * but the propagators (KC N_T NP^2 16 bytes) do not fit the device and the handle evaluates matrix-free instead of
 * failing in hipMalloc (same results to rounding; shards of a composite handle: the number of shards in that mode),
 * [13] the cells of [11] that took the four-product degree-16 route (Hermitian generators, 16 < N <= 64, spectral bound
 * within its range), [14] 1 if the four-product route of this handle is the hand-allocated assembly kernel (csrc/asm/gen_t16.py:
 * Hermitian generators, 49 <= N <= 64, control operators shared by the trajectories; GRAPE_EXPM_ASM=0: the C++ kernel), 2 if
 * the exponential of this handle is the assembly cell for GENERAL matrices (csrc/asm/gen_t18g.py: non-Hermitian drift or
 * controls, same sizes; GRAPE_EXPM_ASM18G=0: the C++ kernel), 3 if it is the four-product assembly cell for control operators
 * per trajectory (csrc/asm/gen_t16p.py: Hermitian generators, up to four controls; GRAPE_EXPM_ASM16P=0), 4 the same for general matrices (gen_t18gp.py: one or two controls),
 * [15] the derivative kernel of the ExpProp route: 0 a compiled kernel, 1 deriv3_asm (49 <= N <= 64, Hermitian, L <= 2),
 * 2 deriv3s_asm (3 <= L <= 8, controls streamed through the LDS), 3 deriv3g_asm (general drift / controls), 4 deriv4_asm
 * (64 < N <= 256), [16] 1 if the products of the blocked polynomial route are the assembly kernel lg_gemm_asm,
 * [17] the steps of the two sweeps that the walks of the exponential kernel carried in the last evaluation (the sweep
 * launch did the remaining 2 K N_T - [17]), [18] the block length of the scanned sweeps of N <= 16 (round 6: the time axis is
 * cut into blocks whose propagators are formed first; 0: sequential sweeps; GRAPE_SCAN16=0 / 1 forces)
 * (entries beyond n are not written). */
int grape_get_work(grape_handle *h, double *out, int n);

const char *grape_last_error(grape_handle *h); /* h may be NULL: error of the last failed create */
int grape_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GRAPE_HIP_H */
