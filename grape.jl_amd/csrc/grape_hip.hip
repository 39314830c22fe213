// grape_hip.hip -- host side of the C ABI declared in include/grape_hip.h (gfx950 only).
//
// Owns every device buffer of the hot path (the GrapeWrk data of
// /root/reference/src/workspace.jl:147-362) and sequences the kernels of
// grape_kernels.hip.h on one HIP stream:
//   expm (all cells) -> forward sweep -> tau partial sums | chi boundary + backward sweep ->
//   per-cell derivative overlaps -> sum over trajectories.
// No CPU fallback exists: every entry point fails loudly when HIP does.
#include "../../include/grape_hip.h"
#include "grape_kernels.hip.h"
#include "grape_t18_coeffs.h"
#include "grape_econ_coeffs.h"
#include "grape_large.hip.h"
#include "grape_series.hip.h"
#include "grape_cheby.hip.h"

#include <rccl/rccl.h>
#include <dlfcn.h>
#include <mutex>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_create_error;

// Exception barrier of the C ABI (include/grape_hip.h: "no exceptions").  Every extern "C" entry point is a
// function-try-block that ends in GRAPE_BARRIER(h): whatever the host side throws -- std::bad_alloc of a staging vector,
// std::system_error of a shard thread -- is turned into GRAPE_ERR_HOST + message; local containers and the guards that own
// half-built handles unwind on the way.  (The caller is a Julia process behind `ccall`: an exception that crossed the
// boundary would end it in std::terminate.)
int barrier_fail(std::string *err, const char *what) noexcept {
    try {
        *err = std::string("host-side C++ exception caught at the C boundary: ") + what;
    } catch (...) {   // (not even the message could be allocated: the status alone has to do)
    }
    return -8;   // GRAPE_ERR_HOST
}
#define GRAPE_BARRIER(errptr)                                                                                   \
    catch (const std::bad_alloc &) { return barrier_fail((errptr), "std::bad_alloc (out of host memory)"); }    \
    catch (const std::exception &e) { return barrier_fail((errptr), e.what()); }                                \
    catch (...) { return barrier_fail((errptr), "unknown exception"); }

// fault injection of the test suite (GRAPE_TEST_HOOKS=1 only): GRAPE_TEST_THROW = bad_alloc | runtime at the named
// point of grape_create (GRAPE_TEST_THROW_AT = early: before the handle exists; late: the handle owns device memory)
void test_throw_point(const char *where) {
    const char *hk = getenv("GRAPE_TEST_HOOKS");
    if (!(hk && atoi(hk) == 1)) return;
    const char *t = getenv("GRAPE_TEST_THROW"), *at = getenv("GRAPE_TEST_THROW_AT");
    if (!t || strcmp(at ? at : "early", where)) return;
    if (!strcmp(t, "bad_alloc")) throw std::bad_alloc();
    throw std::runtime_error(t);
}

constexpr int kRing = 64;   // evaluations kept for phase timing
constexpr int kPhases = 6;
struct Phase { hipEvent_t e0, e1; bool used; };

}  // namespace

struct grape_handle {
    grape_problem p{};
    int N = 0, NP = 0, NT = 0, L = 0, K = 0, K_total = 0, N_T = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    // static problem
    double *d_H0f = nullptr, *d_Hcf = nullptr, *d_H0t = nullptr, *d_Hct = nullptr;
    double *d_H0p = nullptr, *d_Hcp = nullptr, *d_vecs = nullptr;
    double *d_H0q = nullptr, *d_Hcq = nullptr, *d_park2 = nullptr;   // two-pass series kernel: untransposed fragments, parking area
    // blocked path, derivative kernel as assembly (asm/gen_d4.py): fragments with three planes (re, im, re + im) of the
    // operators (pass 1) and of their adjoints (pass 2; Hermitian operators: the same arrays)
    double *d_H0q3 = nullptr, *d_Hcq3 = nullptr, *d_H0p3 = nullptr, *d_Hcp3 = nullptr;
    int deriv4_blocks = 0;
    // one wave per batch (grape_deriv3.hip.h; Hermitian operators, 32 < N <= 64, L <= 2; GRAPE_DERIV3=0: off)
    double *d_park3 = nullptr;
    int deriv3_blocks = 0, deriv3_wpt = 0;
    bool deriv3_h0g = false;     // general drift beside Hermitian control operators (all tiles of H0_k in LDS)
    bool deriv3_general = false; // general operators at four tiles per side: the streamed assembly kernel with all tiles
    int deriv2 = 0, deriv2_maxm = 0;
    bool deriv_stream = false;   // GRAPE_DERIV_STREAM=1: matrix-at-a-time products in deriv2_kernel for 3-4 controls as well
    bool deriv_stream_never = false;   // GRAPE_DERIV_STREAM=0: the all-at-once form for more than four controls too (A/B timing)
    int deriv_blocks = 0;
    // blocked path (64 < N <= 256): per-chunk scratch matrices, planar [cell][2][NP*NP]
    bool large = false;
    int chunk = 0;
    double *d_lg[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double *d_dinv = nullptr;
    int *d_scell = nullptr;
    double *d_colpart = nullptr;   // [chunk][2][LG_PARTS][NP] partial column sums of two powers (lg_t18_operands2_kernel)
    bool lg_form2 = false;         // formation with the operators in registers (lg_form2_kernel; GRAPE_LG_FORM2=0: lg_form_kernel)
    double *d_normpart = nullptr, *d_normpart2 = nullptr;   // [chunk][NP / 8][NP] partial column sums of |A| (per lane of chunks)
    bool lg_fuse = true;           // GRAPE_LG_FUSE=0: the combinations in a pass of their own (lg_t18_operands2_kernel) instead of
                                   // the epilogue of the launch that writes the last power
    bool lg_pow = false;           // GRAPE_LG_POW=1 (round 6, measured, off): B4, B3, B2 formed from the powers by the launches that add them
    bool lg_spec = true;           // GRAPE_LG_SPEC=0: the separate norm pass (lg_t18_scale_kernel) in front of the combinations
    // second lane of the polynomial route (round 5, GRAPE_LG_LANES=2; off by default): the chunks of an evaluation are
    // independent, so odd chunks run on a second stream with their own scratch and fill the launch tails of the even ones
    int lg_lanes = 1;
    double *d_lg2[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double *d_dinv2 = nullptr, *d_colpart2 = nullptr;
    int *d_scell2 = nullptr, *d_smax2 = nullptr;
    hipStream_t lg_stream2 = nullptr;
    hipEvent_t lg_ev_fork = nullptr, lg_ev_join = nullptr;
    double *d_dts = nullptr, *d_shape = nullptr, *d_weights = nullptr;
    double2 *d_psi0 = nullptr, *d_target = nullptr;
    // per-evaluation
    double *d_eps = nullptr;
    double2 *d_U = nullptr, *d_fw = nullptr, *d_bw = nullptr, *d_tg = nullptr;
    double *d_ret = nullptr;  // slab: d_out | d_G | d_flags | d_stats (see grape_create)
    double *d_out = nullptr;  // [2K + 8] tau + partial sums (host API)
    double *d_f = nullptr, *d_G = nullptr, *d_rho = nullptr;
    int *d_flags = nullptr;
    int *d_cellflag = nullptr;   // [K*N_T] cells flagged for the pivoted Pade solve
    // generator classes: trajectories with bit-identical H0 (and control operators) share one set of
    // propagators U_cn; KC == K (d_cls == nullptr) for ensembles of distinct generators
    int KC = 0;
    bool herm = false;           // all generators Hermitian: A = -i dt H is skew-Hermitian (expm uses the symmetry)
    bool herm_ctrl = false;      // every control operator is Hermitian (the drift may not be)
    bool t18 = false;            // inverse-free polynomial exponential (grape_t18.hip.h)
    bool t18_small = true;       // ... also for N <= 32 (GRAPE_EXPM_T18_SMALL=0: the Pade kernels there)
    // four-product degree-16 route for Hermitian generators at 16 < N <= 64 (GRAPE_EXPM_T16=0: off).  Cells whose spectral
    // bound is beyond its range are listed by the kernel and redone by a launch of the degree-18 variant behind it (four
    // products lost per listed cell), so the route is only TRIED where it pays -- decided per evaluation ON THE DEVICE from
    // the pulse values alone (t16_plan_kernel: a spectral-radius estimate from the Gram matrices below); no state of the
    // handle enters, the same pulses always take the same route (round-3 advisor finding: the host-side switch of round 3
    // made the bits of an evaluation depend on the evaluations before it).
    bool t16 = true;
    double *d_gram = nullptr;           // [KC][(L + 1)^2] Re tr(O_a^dagger O_b), O = (H0_k, H_1 .. H_L) of every generator class
    int *d_celllist = nullptr;          // [KC * N_T] the listed cells (counter: d_flags[4])
    // the four-product route as hand-allocated assembly (asm/gen_t16.py; GRAPE_EXPM_ASM=0: the C++ kernel): four tiles
    // per side, Hermitian generators, controls shared by the trajectories
    bool asm16 = false;
    bool asm16p = false;         // ... its variant for control operators per trajectory (one or two controls): expm_t16p_asm fetches the
                                 // operators of the trajectory itself (GRAPE_EXPM_ASM16P=0: controls summed per cell, or the compiled kernel)
    double *d_dte = nullptr;     // [N_T][4] dt, e1, e2, - of every time step (expm_t16p_asm, expm_t18gp_asm; [N_T][8] with four slots)
    bool asm18gp = false;        // general matrices, control operators per trajectory, one or two controls: expm_t18gp_asm
    bool asm18g = false;         // general matrices at four tiles per side: the five-product cell as assembly (GRAPE_EXPM_ASM18G=0: compiled)
    // round 5: the assembly kernel's workgroups walk contiguous ranges of cells (d_wgtab) and carry the state of their
    // trajectory along while the cell's result is in registers -- Psi upwards from t = 0, conj(chi~) downwards from t = T
    // (d_xinit: the two start vectors of every trajectory) -- and report how far each end got (d_prog[2][K]); the sweep
    // kernel behind picks up from there.  GRAPE_EXPM_WALK=0: the kernel only exponentiates (A/B timing, parity twin).
    int *d_wgtab = nullptr, *d_prog = nullptr, *d_splan = nullptr;   // d_splan: squarings planned per cell (scaling and squaring around the four products)
    double2 *d_xinit = nullptr;
    int asm_blocks = 0;
    int last_walk_fuse = 0;      // which ends the walks of the LAST evaluation carried (grape_get_work[17] counts their steps from d_prog)
    bool asm_sq = true;          // GRAPE_EXPM_SQ=0: no scaling and squaring around the four products (cells beyond the bound go to
                                 // the five-product launch, as before round 5)
    int asm_walk = 0;            // bits of the walks that may carry a state: 1 ascending (Psi), 2 descending (conj(chi~)); GRAPE_EXPM_WALK
                                 // = 0 / 1 / 2 narrows it (diagnostics, A/B timing)
    // the stream of the last device-pointer call: the getters that read device buffers wait for the device when it was
    // not the handle's own stream (a caller's non-blocking stream is not ordered against a blocking copy)
    bool foreign_stream = false;
    // the assembly route books its credited statistics (Pade order / squarings Julia's exp! would use) on demand, in
    // grape_get_work: outside the certifying window of the operator-norm bound they need the norm of every cell
    bool credit_pending = false;
    // balancing of general (non-Hermitian) generators, the role of gebal in Julia's exp!: ONE diagonal
    // similarity D = diag(2^e) for the whole handle, chosen at grape_create from sum_k |H0_k| + sum_l |H_l| by gebal's
    // scaling loop; the handle then works on D^-1 H D, D^-1 Psi0, D target (every quantity of the path -- tau, J, the
    // gradient -- is invariant) and the entry points that hand states or propagators across the boundary transform
    // them back.  Empty: no balancing (Hermitian operators are balanced: the loop returns the identity).
    std::vector<double> bal;
    std::vector<double> bal_stage_chi, bal_stage_xi;   // D chi, D xi of the caller's arrays (backward_enqueue)
    double *d_Sf = nullptr;             // [N_T][2][NP*NP] summed control operators of every time step (polynomial kernel, L > 2)
    // diagnostic switches, read ONCE in grape_create (never in the evaluation path: getenv is not thread-safe against setenv)
    bool expm_persist = true;    // GRAPE_EXPM_PERSIST=0: one workgroup per cell instead of the persistent Pade kernel
    int expm_lds_pad_kb = 0;     // GRAPE_EXPM_LDS_PAD: extra dynamic LDS of the Pade kernels (fewer cells per CU)
    int cheby_xmode = 1;         // GRAPE_CHEBY_XMODE
    bool test_hooks = false;     // GRAPE_TEST_HOOKS=1 at grape_create: the fault injection of the test suite (GRAPE_TEST_DROP_SIBLING)
                                 // is looked up per evaluation; without it the evaluation path never calls getenv
    bool lg_asm = true;          // GRAPE_LG_ASM=0: the compiled lg_gemm_kernel for the products of the blocked polynomial route
    bool deriv3_asm = true;      // GRAPE_DERIV3_ASM=0: the compiled deriv3_kernel<4, L> instead of deriv3_asm
    // taylor_grad_check_convergence = false (optimize.jl:917-918, :631-651): a series that is cut at taylor_max_order is
    // not an error (grape_problem.taylor_no_check)
    bool taylor_check = true;
    // trajectories without a target_state (optimize.jl:753: tau_k = NaN): only the caller-side J_T / chi route is legal
    bool no_target = false;
    std::vector<int> cls;        // [K] class of trajectory k
    int *d_cls = nullptr, *d_rep = nullptr;
    unsigned *d_coop = nullptr;  // [2][K] step counters of the cooperative sweeps (forward, backward)
    int coop_S = 0;              // workgroups per trajectory in the cooperative sweeps (0: one-workgroup kernel)
    int *d_xcc_sw = nullptr;     // [2][K][32] XCC ids of the siblings of the cooperative sweeps (forward, backward)
    int coop_xmode = 1;          // GRAPE_COOP_XMODE=0: step counters instead of armed storage rows (SweepArgs::xmode)
    int coop_rpw = 0, coop_nw = 0;  // rows per wave and waves of a cooperative workgroup (R = nw * rpw state rows)
    int coop_S_fw = 0, coop_rpw_fw = 0, coop_nw_fw = 0;   // the forward sweep's own split (GRAPE_COOP_S_FW; default: fewer siblings, see grape_create)
    // state running cost (g_b = <Psi|D|Psi>): transposed D, trapezoid weights, xi and g per stored state
    double2 *d_Dt = nullptr, *d_xi = nullptr;
    double *d_wq = nullptr, *d_gb = nullptr;
    bool have_gb = false;
    // caller-supplied inhomogeneity xi_k(t_n) of an arbitrary state running cost (grape_backward_xi): uploaded into d_xi
    const double2 *xi_user = nullptr;   // non-null only while that call's backward phase is being enqueued
    double lambda_user = 0.0;
    unsigned long long *d_stats = nullptr;
    double *h_pin = nullptr;  // pinned staging
    size_t h_pin_doubles = 0;
    Phase ph[kRing][kPhases]{};
    long n_fwd = 0, n_bwd = 0;  // evaluations recorded since the last grape_reset_timings
    std::string err;
    bool have_forward = false, in_eval = false;
    double chi_min_norm = 1e-100, taylor_tol = 1e-16;
    int taylor_max_order = 100;
    // matrix-free polynomial propagator (prop_method = GRAPE_PROP_SERIES): no U is materialised
    // concurrent sweeps: the backward sweep starts from the unit targets in the same launch as the forward sweep
    bool fuse = false;           // capability (no state running cost, N <= 64) and not switched off
    bool fuse_on = true;         // grape_set_fused_sweeps
    bool want_bw = true;         // the evaluation in flight wants a gradient (grape_eval with G == NULL clears it)
    bool bw_done = false;        // the last forward call already ran the (unit) backward sweep
    bool bw_unit = false;        // d_bw holds the unit backward states chi~ (storage getter applies the phase)
    bool z_valid = false;        // d_z belongs to the states in d_bw (grape_backward ran after the fused forward)
    double *d_inv_tnorm = nullptr, *d_ones = nullptr;
    double2 *d_z = nullptr;
    // round 6: the sweeps of N <= 16 as a parallel scan over the time axis (scan16_* kernels, grape_kernels.hip.h): block
    // propagators, boundary states of the coarse sweeps.  On for small ensembles (GRAPE_SCAN16=0: off, =1: always)
    bool scan16 = false;
    int scan_Bk = 0, scan_NB = 0;
    double2 *d_scanF = nullptr, *d_scan_fw = nullptr, *d_scan_bw = nullptr;
    int num_cus = 256;
    bool series = false;
    bool u_fallback = false;     // prop_method = ExpProp was asked for, but the propagators do not fit the device: matrix-free
    double *d_rb = nullptr;      // [K + Kc*L] 2-norm estimates of H0_k and of the control operators
    double series_tol = 1e-17, series_theta = 3.0;
    double *d_n1 = nullptr;      // 1-norms of H0_k and of the control operators (order-13 certificate of the expm kernel)
    double2 *d_gpark = nullptr;  // [K][N_T][maxp][NP] terms of the forward series, consumed by deriv2_kernel
    int *d_morder = nullptr;     // [K][N_T]
    int maxp = 0;
    int *d_batchflag = nullptr;  // [2][K * ceil(N_T / 16)] derivative batches left to deriv_sub_kernel (sub-stepped series) | certified for the economized series
    // Round 6: the economized derivative series (tools/econ_coeffs.py, asm/gen_d3.py).  Batches of 16 cells that the
    // four-product exponential kernel of THIS evaluation has certified (verdict 0 without scaling: spectral radius <= 1.36)
    // take a degree-16 polynomial whose derivative is within 2e-16 of exp's on that segment where the Taylor sum needs
    // 19-21 terms.  Exact-derivative route only (:taylor is the reference's recursion, term by term), default tolerance
    // or looser... a tighter one keeps the Taylor sum.  GRAPE_DERIV_ECON=0: off.
    bool deriv_econ = false;
    const double *d_econ_pairs = nullptr;   // device: (omega, sigma) of the economized polynomials, degree M at 64 (M - 16) doubles
    int *d_celldeg = nullptr;    // blocked path: [KC * N_T] degree named by lg_t18_decide_kernel for every cell
    double sub_theta = 0.0;      // threshold of deriv_substeps (0: off, gradient_method = :taylor mirrors the reference)
    int sq_plan = 2;             // blocked path: squaring launches issued per chunk (adapted by grape_check, see expm_large)
    // matrix-free propagator for 64 < N <= 256 (grape_cheby.hip.h): exchange slots, counters, launch plan
    double2 *d_xch = nullptr;    // [2][K][4][NP]  (forward, backward), armed with the sentinel before every launch
    int *d_xcc = nullptr;        // [2][K][32] XCC ids of the siblings (which XCD does each workgroup run on?)
    int cheby_S = 0, cheby_round = 0, cheby_pair = 0;   // siblings per trajectory, trajectories per launch, both directions in one launch
    double2 *d_chi_in = nullptr; // [K][N] host-supplied boundary states of grape_backward_chi (allocated on first use)
    // several GPUs behind one handle (grape_problem.ndev > 1): this handle owns no device memory, its trajectories
    // are dealt to the child handles in contiguous blocks [shard_lo[g], shard_lo[g+1])
    std::vector<grape_handle *> shards;
    std::vector<int> shard_lo;
    std::vector<int> shard_dev;
    // composite handle: the two cross-shard reductions (8 partial sums between the sweeps, the gradient at the end) as
    // RCCL all-reduces on the shard streams (ncclCommInitAll: one communicator rank per device of this process) when the
    // shard devices are distinct; otherwise, or with GRAPE_MULTI_RCCL=0, or when RCCL cannot be loaded: staged through
    // pinned host memory and added in shard order.  d_red: [8 + L N_T] all-reduced copies per shard.
    bool use_rccl = false;
    void *comm_set = nullptr;      // CommSet (shared by the handles of one device list, see comm_set_acquire)
    std::vector<double *> d_red;
    // The collective path has to earn its place: the first evaluation of a handle ALSO takes the host-staged sums (shard
    // order) and compares -- the 8 partial sums after the forward half, the gradient after the backward half, and that
    // every rank holds the same totals.  A disagreement beyond rounding fails the call loudly (GRAPE_ERR_HIP).
    bool rccl_fwd_checked = false, rccl_bwd_checked = false;
    hipEvent_t ar0 = nullptr, ar1 = nullptr;   // around shard 0's gradient all-reduce
    double allreduce_ms = 0.0;
    long allreduce_calls = 0;
    std::vector<double> h_multi;   // host scratch of the composite: pulses, per-shard gradients
    bool multi_threads = true;     // composite: the enqueue half of every shard from its own host thread (GRAPE_MULTI_THREADS=0:
                                   // one after the other from the calling thread)
    double host_enqueue_ms = 0.0;  // composite: wall time of the enqueue halves since the last grape_reset_timings
    long host_enqueue_calls = 0;
    // Small systems are launch-bound (C2: ~14 launches, copies and fills for 0.25 ms of kernels): grape_eval with a gradient
    // replays the whole evaluation -- H2D of the pulses, every kernel, D2H of the result slab -- as ONE captured HIP graph
    // (N <= 64, one device).  Nothing in that sequence is decided on the host from device data, so the graph of a handle
    // never changes (it is rebuilt when grape_set_fused_sweeps changes the sweeps).  The first two evaluations and every
    // 16th run uncaptured: they warm the lazy parts of the runtime (module loads, LDS limits) and keep the per-phase HIP-event
    // timings alive (events recorded inside a capture cannot be read).  GRAPE_GRAPH=0: off.
    bool graph_ok = false, capturing = false;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    bool graph_fuse_on = true;     // the setting the graph was captured with
    bool graph_bw_unit = false, graph_z_valid = false, graph_credit = false;   // host-side state the captured calls leave behind
    long n_eval = 0;               // grape_eval calls on the single-wait path
};

namespace {

#define HIPCHK(h, expr)                                                                         \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
            return GRAPE_ERR_HIP;                                                               \
        }                                                                                       \
    } while (0)

template <typename T>
hipError_t dmalloc(T **p, size_t n) { return hipMalloc((void **)p, n * sizeof(T)); }

// hipFuncAttributeMaxDynamicSharedMemorySize is raised per kernel and device whenever a launch needs more than has been
// set so far (the size depends on NP and on diagnostic padding: a cache keyed by the device alone would let a second,
// larger handle launch with the first one's limit)
struct LdsLimit {
    size_t set[64] = {0};
    hipError_t ensure(const void *fn, int dev, size_t bytes) {
        size_t &cur = set[((unsigned)dev) % 64u];
        if (bytes <= cur) return hipSuccess;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) cur = bytes;
        return e;
    }
};

size_t expm_lds_bytes(int NT) {
    const int total = NT == 1 ? ExpmLds<1>::TOTAL : NT == 2 ? ExpmLds<2>::TOTAL : NT == 3 ? ExpmLds<3>::TOTAL : ExpmLds<4>::TOTAL;
    return sizeof(double) * (size_t)total;
}

template <int NT>
hipError_t launch_expm(const ExpmArgs &a, bool herm, hipStream_t s, int persistent_blocks = 0, int lds_pad_kb = 0) {
    static LdsLimit lim_fast, lim_piv, lim_herm, lim_pers, lim_persh;
    const size_t lds = expm_lds_bytes(NT) + (size_t)lds_pad_kb * 1024;   // (padding: diagnostic, fewer cells per CU)
    int dev = 0;
    hipGetDevice(&dev);
    {
        hipError_t e = lim_fast.ensure((const void *)expm_pade_kernel<NT, false>, dev, lds);
        if (e == hipSuccess) e = lim_piv.ensure((const void *)expm_pade_kernel<NT, true>, dev, lds);
        if (e == hipSuccess && NT >= 3) e = lim_herm.ensure((const void *)expm_pade_kernel<NT, false, NT >= 3>, dev, lds);
        if (e != hipSuccess) return e;
    }
    hipError_t e = hipMemsetAsync(a.cellflag, 0, (size_t)a.K * a.N_T * sizeof(int), s);
    if (e != hipSuccess) return e;
    // fast pass: unpivoted block Gauss-Jordan, flags the cells it cannot solve safely; one workgroup per cell
    if (NT == 4 && persistent_blocks > 0) {   // one workgroup per CU walks its cells (see expm_persistent)
        hipError_t ep = lim_pers.ensure((const void *)expm_persistent_kernel<NT, false>, dev, lds);
        if (ep == hipSuccess) ep = lim_persh.ensure((const void *)expm_persistent_kernel<NT, NT == 4>, dev, lds);
        if (ep != hipSuccess) return ep;
        if (herm) hipLaunchKernelGGL((expm_persistent_kernel<NT, NT == 4>), dim3(persistent_blocks), dim3(NT * 64), lds, s, a);
        else hipLaunchKernelGGL((expm_persistent_kernel<NT, false>), dim3(persistent_blocks), dim3(NT * 64), lds, s, a);
    } else
    if (herm && NT >= 3)   // Hermitian generators: NT - 1 of NT row tiles per strip from the MFMAs
        hipLaunchKernelGGL((expm_pade_kernel<NT, false, NT >= 3>), dim3(a.K * a.N_T), dim3(NT * 64), lds, s, a);
    else
        hipLaunchKernelGGL((expm_pade_kernel<NT, false>), dim3(a.K * a.N_T), dim3(NT * 64), lds, s, a);
    // pivoted pass over the flagged cells (all other workgroups exit at once)
    hipLaunchKernelGGL((expm_pade_kernel<NT, true>), dim3(std::min(a.K * a.N_T, 1024)), dim3(NT * 64), lds, s, a);
    return hipGetLastError();
}

// Hermitian generators, N in (32, 64]: inverse-free degree-18 polynomial kernel (grape_t18.hip.h, its own translation
// unit grape_t18.hip), persistent grid
extern "C" int grape_t18_launch(int NT, int herm, int t16, const void *args, size_t args_size, void *stream, int blocks);
extern "C" const double *grape_econ_pairs(void);   // grape_t18.hip: the tables of the economized series on the current device
extern "C" int grape_t16p_asm_launch(const void *args, size_t args_size, int *verdict, void *stream, int blocks,
                                     const void *const *walk, int fuse, int K, const double *dte);
extern "C" int grape_t18g_asm_launch(const void *args, size_t args_size, int *verdict, void *stream, int blocks,
                                     const void *const *walk, int fuse, int K, const double *dte);
extern "C" int grape_t16_asm_launch(const void *args, size_t args_size, int *verdict, void *stream, int blocks,
                                    const void *const *walk, int fuse, int K);
extern "C" void grape_t16_walks(int KC, int N_T, int nblk, int *tab);
extern "C" int grape_t16_credit_launch(const void *args, size_t args_size, void *stream);
// deriv3_kernel keeps the upper 16 x 16 tiles (re, im; stride 17) of H0_k and of the L control operators in LDS
// (general drift: all NT x NT tiles of H0_k, three and four tiles per side and at most two controls)
static bool deriv3_fits(int NT, int L, bool h0_general = false) {
    const size_t tile = (size_t)2 * 16 * 17 * sizeof(double), mat = (size_t)(NT * (NT + 1) / 2) * tile;
    if (h0_general) return NT >= 3 && NT <= 4 && L >= 1 && L <= 2 && (size_t)NT * NT * tile + (size_t)L * mat <= 160 * 1024;
    return L >= 1 && L <= 8 && (size_t)(1 + L) * mat <= 160 * 1024;
}
extern "C" int grape_deriv3_launch(int NT, const void *d2args, size_t d2size, const double *H0f, const double *Hcf, int wpt,
                                   int skip_if_flagged, int h0_general, int use_asm, void *stream, int blocks);
extern "C" int grape_lg_asm_launch(const void *k, size_t size, unsigned blocks, void *stream);
extern "C" int grape_deriv4_launch(int NP, const void *d2args, size_t d2size, const double *H0q3, const double *Hcq3, const double *H0p3,
                                   const double *Hcp3, void *stream, int blocks);
#ifdef GRAPE_DIAG
extern "C" void grape_t18_set_stamps(unsigned long long *d_stamps, void *stream);
#endif

template <int CPL>
hipError_t launch_coop(const SweepArgs &a, bool backward, int S, int rpw, int nw, unsigned *cnt, hipStream_t s) {
    hipError_t e = hipMemsetAsync(cnt, 0, (size_t)a.K * sizeof(unsigned), s);
    if (e != hipSuccess) return e;
    if (a.xcc) {
        e = hipMemsetAsync(a.xcc, 0xFF, (size_t)a.K * 32 * sizeof(int), s);
        if (e != hipSuccess) return e;
    }
    if (a.xmode) {   // arm every row of the storage (the boundary row of a trajectory is written by sibling 0 inside the kernel and polled by nobody)
        e = hipMemsetAsync((void *)a.store, 0xFF, (size_t)a.K * (a.N_T + 1) * 64 * CPL * sizeof(double2), s);
        if (e != hipSuccess) return e;
    }
    const dim3 grid(8 * ((a.K + 7) / 8) * S), block(64 * nw);
#define COOP_CASE(NW_, RPW_)                                                                                          \
    if (nw == NW_ && rpw == RPW_) {                                                                                   \
        if (backward) hipLaunchKernelGGL((sweep_coop_kernel<CPL, RPW_, NW_, true>), grid, block, 0, s, a, S, cnt);    \
        else hipLaunchKernelGGL((sweep_coop_kernel<CPL, RPW_, NW_, false>), grid, block, 0, s, a, S, cnt);            \
        return hipGetLastError();                                                                                     \
    }
    COOP_CASE(4, 1) COOP_CASE(8, 1) COOP_CASE(16, 1) COOP_CASE(16, 2) COOP_CASE(16, 4)
#undef COOP_CASE
    return hipErrorInvalidValue;
}

// GRAPE_SWEEP1W=0 (read in grape_create, process-wide: a diagnostic): the 256-thread sweeps at two tiles per side
static bool g_sweep_wg32 = false;
template <int NP>
hipError_t launch_sweep(const SweepArgs &a, bool backward, hipStream_t s);
template <int NP>
hipError_t launch_sweep_pair(const SweepArgs &af, const SweepArgs &ab, hipStream_t s);

// phases 1 and 3 of the scanned sweeps for 17 <= N <= 64 (scan_block_kernel / scan_fill_kernel)
template <int NT>
hipError_t launch_scan_block(const ScanArgs &sa, hipStream_t s) {
    static LdsLimit lim;
    constexpr int NP = 16 * NT;
    const size_t lds = (size_t)2 * 2 * NP * (NP + 2) * sizeof(double);
    int dev = 0;
    hipGetDevice(&dev);
    hipError_t e = lim.ensure((const void *)scan_block_kernel<NT>, dev, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((scan_block_kernel<NT>), dim3((unsigned)(sa.KC * sa.NB)), dim3(64 * NT), lds, s, sa);
    return hipGetLastError();
}

// Parallel scan of the sweeps over the time axis (grape_kernels.hip.h: scan16_* for N <= 16, scan_* for 17 <= N <= 64): block
// propagators (once per set of propagators: `blocks`), the ordinary sweeps over them, then all blocks filled at once.
// af / ab: the arguments of the fine sweeps (nullptr: that direction is not wanted)
hipError_t launch_scan(grape_handle *h, const SweepArgs *af, const SweepArgs *ab, bool blocks, hipStream_t s) {
    const int NB = h->scan_NB, Bk = h->scan_Bk, NP = h->NP;
    const SweepArgs &any = af ? *af : *ab;
    hipError_t e = hipSuccess;
    if (blocks) {
        if (NP == 16) {
            Scan16Args sa{};
            sa.U = any.U; sa.F = h->d_scanF; sa.KC = h->KC; sa.N_T = any.N_T; sa.Bk = Bk; sa.NB = NB;
            hipLaunchKernelGGL(scan16_block_kernel, dim3((unsigned)(h->KC * NB)), dim3(64), 0, s, sa);
            e = hipGetLastError();
        } else {
            ScanArgs sa{};
            sa.U = any.U; sa.F = h->d_scanF; sa.KC = h->KC; sa.N_T = any.N_T; sa.Bk = Bk; sa.NB = NB;
            e = NP == 32 ? launch_scan_block<2>(sa, s) : NP == 48 ? launch_scan_block<3>(sa, s) : launch_scan_block<4>(sa, s);
        }
        if (e != hipSuccess) return e;
    }
    SweepArgs cf{}, cb{};
    if (af) { cf = *af; cf.U = h->d_scanF; cf.N_T = NB; cf.store = h->d_scan_fw; cf.resume = nullptr; }
    if (ab) { cb = *ab; cb.U = h->d_scanF; cb.N_T = NB; cb.store = h->d_scan_bw; cb.resume = nullptr; }
    if (af && ab) e = NP == 16 ? launch_sweep_pair<16>(cf, cb, s) : NP == 32 ? launch_sweep_pair<32>(cf, cb, s)
                      : NP == 48 ? launch_sweep_pair<48>(cf, cb, s) : launch_sweep_pair<64>(cf, cb, s);
    else {
        const SweepArgs &c1 = af ? cf : cb;
        const bool bw = !af;
        e = NP == 16 ? launch_sweep<16>(c1, bw, s) : NP == 32 ? launch_sweep<32>(c1, bw, s)
            : NP == 48 ? launch_sweep<48>(c1, bw, s) : launch_sweep<64>(c1, bw, s);
    }
    if (e != hipSuccess) return e;
    const unsigned nblk = (unsigned)(any.K * NB * ((af && ab) ? 2 : 1));
    const SweepArgs &a1 = af ? *af : *ab, &a2 = ab ? *ab : *af;
    if (NP <= 32) {
        Scan16FillArgs fa{};
        fa.cfw = h->d_scan_fw; fa.cbw = h->d_scan_bw; fa.Bk = Bk; fa.NB = NB; fa.both = (af && ab) ? 1 : 0;
        if (NP == 16) hipLaunchKernelGGL((scan1w_fill_kernel<16>), dim3(nblk), dim3(64), 0, s, a1, a2, fa, af ? 0 : 1);
        else hipLaunchKernelGGL((scan1w_fill_kernel<32>), dim3(nblk), dim3(64), 0, s, a1, a2, fa, af ? 0 : 1);
    } else {
        ScanFillArgs fa{};
        fa.cfw = h->d_scan_fw; fa.cbw = h->d_scan_bw; fa.Bk = Bk; fa.NB = NB;
        if (NP == 48) hipLaunchKernelGGL((scan_fill_kernel<48>), dim3(nblk), dim3(192), 0, s, a1, a2, fa, af ? 0 : 1);
        else hipLaunchKernelGGL((scan_fill_kernel<64>), dim3(nblk), dim3(256), 0, s, a1, a2, fa, af ? 0 : 1);
    }
    return hipGetLastError();
}

template <int NP>
hipError_t launch_sweep(const SweepArgs &a, bool backward, hipStream_t s) {
    if (NP == 16 || (NP == 32 && !g_sweep_wg32)) {   // one wave per trajectory, no barriers in the time loop (NP = 32: round 6; GRAPE_SWEEP1W=0: the workgroup kernel)
        constexpr int P = NP <= 16 ? 16 : 32;
        if (backward) hipLaunchKernelGGL((sweep1w_kernel<P, true>), dim3(a.K), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((sweep1w_kernel<P, false>), dim3(a.K), dim3(64), 0, s, a);
        return hipGetLastError();
    }
    if constexpr (NP == 16) return hipErrorInvalidValue;
    constexpr int NTH = NP == 48 ? 192 : 256;   // NP = 48: three waves of 16 rows
    if (backward) hipLaunchKernelGGL((sweep_kernel<NP, true>), dim3(a.K), dim3(NTH), 0, s, a);
    else hipLaunchKernelGGL((sweep_kernel<NP, false>), dim3(a.K), dim3(NTH), 0, s, a);
    return hipGetLastError();
}

template <int NP>
hipError_t launch_sweep_pair(const SweepArgs &af, const SweepArgs &ab, hipStream_t s) {
    if (NP == 16 || (NP == 32 && !g_sweep_wg32)) hipLaunchKernelGGL((sweep1w_pair_kernel<(NP <= 16 ? 16 : 32)>), dim3(2 * af.K), dim3(64), 0, s, af, ab);
    else if constexpr (NP == 16) return hipErrorInvalidValue;
    else hipLaunchKernelGGL((sweep_pair_kernel<NP>), dim3(2 * af.K), dim3(NP == 48 ? 192 : 256), 0, s, af, ab);
    return hipGetLastError();
}

template <int NP>
hipError_t launch_series_pair(const SeriesArgs &af, const SeriesArgs &ab, bool stream_controls, hipStream_t s) {
    const dim3 grid(2 * af.s.K), block(NP * 16 / SERIES_RPT);
    // more workgroups than CUs: the variant that streams the control operators from L2 needs half the registers, two
    // workgroups share a CU and fill each other's latencies (K = 512 at N = 64: 29 instead of 38 ms); with a CU per
    // workgroup the register-resident operators are faster (K = 128: 10.1 vs 11.6 ms)
    if (stream_controls) hipLaunchKernelGGL((series_pair_kernel<NP, 0>), grid, block, 0, s, af, ab);
    else if (af.L == 1) hipLaunchKernelGGL((series_pair_kernel<NP, 1>), grid, block, 0, s, af, ab);
    else if (af.L == 2) hipLaunchKernelGGL((series_pair_kernel<NP, 2>), grid, block, 0, s, af, ab);
    else hipLaunchKernelGGL((series_pair_kernel<NP, 0>), grid, block, 0, s, af, ab);
    return hipGetLastError();
}

template <int NP>
hipError_t launch_series(const SeriesArgs &a, bool backward, hipStream_t s) {
    // control tiles stay in registers for L <= 2 (instantiated per count), otherwise they stream from L2
    const dim3 grid(a.s.K), block(NP * 16 / SERIES_RPT);
#define SERIES_CASE(LR_)                                                                              \
    do {                                                                                              \
        if (backward) hipLaunchKernelGGL((series_sweep_kernel<NP, true, LR_>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((series_sweep_kernel<NP, false, LR_>), grid, block, 0, s, a);         \
    } while (0)
    if (a.L == 1) SERIES_CASE(1);
    else if (a.L == 2) SERIES_CASE(2);
    else SERIES_CASE(0);
#undef SERIES_CASE
    return hipGetLastError();
}

// RCCL, loaded at run time (no link dependency: a process that never builds a multi-device handle never touches it; a
// process that already carries RCCL -- PyTorch does -- gets that copy)
struct RcclApi {
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
const RcclApi &rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, []() {
        void *lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
        api.CommInitAll = (decltype(api.CommInitAll))dlsym(lib, "ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(lib, "ncclCommDestroy");
        api.AllReduce = (decltype(api.AllReduce))dlsym(lib, "ncclAllReduce");
        api.GroupStart = (decltype(api.GroupStart))dlsym(lib, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(lib, "ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(lib, "ncclGetErrorString");
        api.ok = api.CommInitAll && api.CommDestroy && api.AllReduce && api.GroupStart && api.GroupEnd;
    });
    return api;
}

// One communicator set per device list and process (advisor finding of round 4: ncclCommInitAll per handle cost every
// grape_create of a composite handle a full communicator initialisation).  Handles on the same devices share the set; a
// grouped all-reduce is enqueued under the set's mutex (RCCL does not allow concurrent use of one communicator from
// several host threads, and every rank must see the collectives of two handles in the same order).  A set whose last
// handle is gone stays cached for the next one (GRAPE_MULTI_COMM_CACHE=0: it is destroyed with its last handle).
struct CommSet {
    std::vector<int> devs;
    std::vector<ncclComm_t> comms;
    std::mutex mtx;
    int refs = 0;
};
std::mutex g_comm_mtx;
std::vector<CommSet *> g_comm_sets;
CommSet *comm_set_acquire(const std::vector<int> &devs) {
    const RcclApi &api = rccl_api();
    if (!api.ok) return nullptr;
    std::lock_guard<std::mutex> lock(g_comm_mtx);
    for (CommSet *c : g_comm_sets)
        if (c->devs == devs) { c->refs++; return c; }
    CommSet *c = new CommSet();
    c->devs = devs;
    c->comms.assign(devs.size(), nullptr);
    if (api.CommInitAll(c->comms.data(), (int)devs.size(), devs.data()) != ncclSuccess) {
        for (ncclComm_t q : c->comms)   // (a failed initialisation may have created some ranks: none may leak)
            if (q) api.CommDestroy(q);
        delete c;
        (void)hipGetLastError();
        return nullptr;
    }
    c->refs = 1;
    g_comm_sets.push_back(c);
    return c;
}
void comm_set_release(CommSet *c) {
    if (!c) return;
    std::lock_guard<std::mutex> lock(g_comm_mtx);
    if (--c->refs > 0) return;
    const char *env = getenv("GRAPE_MULTI_COMM_CACHE");
    if (!(env && atoi(env) == 0)) return;   // stays cached
    for (ncclComm_t q : c->comms)
        if (q) rccl_api().CommDestroy(q);
    g_comm_sets.erase(std::remove(g_comm_sets.begin(), g_comm_sets.end(), c), g_comm_sets.end());
    delete c;
}

// The scaling half of LAPACK gebal (2-norm form, factors of two, 5 % rule) on a real non-negative N x N matrix
// (column-major): d with M <- D^-1 M D balanced.  All ones for a symmetric matrix.
std::vector<double> gebal_scaling(int N, std::vector<double> M) {
    std::vector<double> d((size_t)N, 1.0);
    const double radix = 2.0, sfmin1 = 2.2250738585072014e-308 / 2.220446049250313e-16, sfmax1 = 1.0 / sfmin1;
    const double sfmin2 = sfmin1 * radix, sfmax2 = 1.0 / sfmin2;
    for (bool noconv = true; noconv;) {
        noconv = false;
        for (int i = 0; i < N; ++i) {
            double c = 0., r = 0., ca = 0., ra = 0.;
            for (int j = 0; j < N; ++j) {
                const double cji = M[(size_t)i * N + j], rij = M[(size_t)j * N + i];
                c += cji * cji; r += rij * rij;
                ca = std::max(ca, cji); ra = std::max(ra, rij);
            }
            c = std::sqrt(c); r = std::sqrt(r);
            if (c == 0.0 || r == 0.0) continue;
            // (LAPACK's xGEBAL leaves with an error on a NaN; here: no balancing -- the evaluation reports the generator that is
            // not finite.  Without this exit the loop below never converges: every comparison with a NaN is false.  Round 5.)
            if (!(c + r + ca + ra <= 1.7976931348623157e308)) return std::vector<double>((size_t)N, 1.0);
            double g = r / radix, f = 1.0;
            const double s0 = c + r;
            while (c < g && std::max(f, std::max(c, ca)) < sfmax2 && std::min(r, std::min(g, ra)) > sfmin2) {
                f *= radix; c *= radix; ca *= radix; r /= radix; g /= radix; ra /= radix;
            }
            g = c / radix;
            while (g >= r && std::max(r, ra) < sfmax2 && std::min(std::min(f, c), std::min(g, ca)) > sfmin2) {
                f /= radix; c /= radix; g /= radix; ca /= radix; r *= radix; ra *= radix;
            }
            if (c + r >= 0.95 * s0) continue;
            if (f < 1.0 && d[i] < 1.0 && f * d[i] <= sfmin1) continue;
            if (f > 1.0 && d[i] > 1.0 && d[i] >= sfmax1 / f) continue;
            d[i] *= f;
            noconv = true;
            for (int j = 0; j < N; ++j) M[(size_t)j * N + i] /= f;   // row i
            for (int j = 0; j < N; ++j) M[(size_t)i * N + j] *= f;   // column i
        }
    }
    return d;
}

// fn(q) for q in [0, n) over the host cores.  A std::thread that cannot be created (std::system_error) is not an error:
// its share runs on the calling thread; threads that did start are always joined (a joinable std::thread that is
// destroyed ends the process).  An exception that fn throws INSIDE a worker (std::bad_alloc of a scratch vector) must not
// leave the thread function -- that is std::terminate, past every barrier of the C boundary (round-5 advisor finding):
// the worker keeps the first one and the calling thread rethrows it after the join, where GRAPE_BARRIER sees it.
template <class F>
void parallel_for(int n, F fn, unsigned max_threads = 32u) {
    const int nth = (int)std::max(1u, std::min<unsigned>(std::min<unsigned>(std::thread::hardware_concurrency(), max_threads), (unsigned)std::max(n, 1)));
    if (nth <= 1 || n <= 1) { for (int q = 0; q < n; ++q) fn(q); return; }
    std::vector<std::thread> pool;
    pool.reserve((size_t)nth);
    std::vector<char> covered((size_t)nth, 0);
    std::vector<std::exception_ptr> thrown((size_t)nth);
    try {
        for (int t = 0; t < nth; ++t) {
            pool.emplace_back([&, t]() {
                try {
                    if (t == nth - 1) test_throw_point("worker");
                    for (int q = t; q < n; q += nth) fn(q);
                } catch (...) { thrown[(size_t)t] = std::current_exception(); }
            });
            covered[t] = 1;
        }
    } catch (const std::system_error &) {
    }
    for (auto &th : pool) th.join();
    for (auto &ep : thrown)
        if (ep) std::rethrow_exception(ep);
    for (int t = 0; t < nth; ++t)
        if (!covered[t]) for (int q = t; q < n; q += nth) fn(q);
}

// The balancing similarity of a problem (grape_handle::bal): gebal's scaling loop on M = sum_k |H0_k| + sum_l |H_l|
// (element moduli).  Empty: the identity (always for Hermitian operators: M is symmetric) or GRAPE_BALANCE=0.
std::vector<double> balance_of_problem(const grape_problem *p) {
    std::vector<double> bal;
    const char *envb = getenv("GRAPE_BALANCE");
    if (envb && atoi(envb) == 0) return bal;
    const int N = p->N, K = p->K, L = p->L, Kc = p->hc_per_traj ? K : 1;
    const size_t nn = (size_t)N * N;
    std::vector<double> M(nn, 0.0);
    auto add = [&](const double *op) { for (size_t q = 0; q < nn; ++q) M[q] += std::hypot(op[2 * q], op[2 * q + 1]); };
    for (int k = 0; k < K; ++k) add(p->H0 + 2 * (size_t)k * nn);
    for (int q = 0; q < Kc * L; ++q) add(p->Hc + 2 * (size_t)q * nn);
    bal = gebal_scaling(N, M);
    bool ident = true;
    for (double x : bal) ident = ident && x == 1.0;
    if (ident) bal.clear();
    return bal;
}

// 2-norm estimate of an N x N complex column-major matrix: power iteration on M^dagger M (deterministic start),
// inflated by 10 %.  Used only to choose the number of sub-steps of the matrix-free propagator.
double norm2_estimate(const double *M, int N) {
    std::vector<double> v(2 * (size_t)N), w(2 * (size_t)N), y(2 * (size_t)N);
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    for (int j = 0; j < 2 * N; ++j) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        v[j] = (double)((st >> 11) & 0xFFFFF) / 1048576.0 - 0.5;
    }
    double sigma2 = 0.0;
    for (int it = 0; it < 40; ++it) {
        double nv = 0.0;
        for (int j = 0; j < 2 * N; ++j) nv += v[j] * v[j];
        nv = std::sqrt(nv);
        if (!(nv > 0.0)) return 0.0;
        for (int j = 0; j < 2 * N; ++j) v[j] /= nv;
        std::fill(w.begin(), w.end(), 0.0);
        for (int j = 0; j < N; ++j) {   // w = M v   (column j of M times v_j)
            const double vr = v[2 * j], vi = v[2 * j + 1];
            const double *col = M + 2 * (size_t)j * N;
            for (int i = 0; i < N; ++i) {
                w[2 * i] += col[2 * i] * vr - col[2 * i + 1] * vi;
                w[2 * i + 1] += col[2 * i] * vi + col[2 * i + 1] * vr;
            }
        }
        for (int j = 0; j < N; ++j) {   // y = M^dagger w
            const double *col = M + 2 * (size_t)j * N;
            double yr = 0.0, yi = 0.0;
            for (int i = 0; i < N; ++i) {
                yr += col[2 * i] * w[2 * i] + col[2 * i + 1] * w[2 * i + 1];
                yi += col[2 * i] * w[2 * i + 1] - col[2 * i + 1] * w[2 * i];
            }
            y[2 * j] = yr; y[2 * j + 1] = yi;
        }
        double ny = 0.0;
        for (int j = 0; j < 2 * N; ++j) ny += y[j] * y[j];
        sigma2 = std::sqrt(ny);   // ||M^dagger M v|| with ||v|| = 1 -> largest eigenvalue of M^dagger M
        v = y;
    }
    return 1.1 * std::sqrt(sigma2);
}

// RIGOROUS upper bound of the 2-norm of a Hermitian N x N matrix (column-major interleaved complex):
// ||M||_2 = rho(M) and rho(M)^p = rho(M^p) <= ||M^p||_1, so min(||M||_1, ||M^2||_1^(1/2), ||M^4||_1^(1/4)) bounds it;
// for a matrix with semicircle spectrum the last figure is 1.2-1.4 rho at N = 64...256 where ||M||_1 is 4-8 rho.
// The Chebyshev propagator needs a GUARANTEED spectral interval (an eigenvalue outside it makes the three-term recursion
// grow exponentially, silently); the power iteration above approaches the norm from below and guarantees nothing.
// Two N^3 products per operator, once per grape_create.
double herm_norm2_bound(const double *M, int N) {
    const size_t nn = (size_t)N * N;
    auto norm1 = [&](const std::vector<double> &a) {
        double best = 0.0;
        for (int j = 0; j < N; ++j) {
            double cs = 0.0;
            for (int i = 0; i < N; ++i) cs += std::hypot(a[2 * ((size_t)j * N + i)], a[2 * ((size_t)j * N + i) + 1]);
            best = std::max(best, cs);
        }
        return best;
    };
    auto square = [&](const std::vector<double> &a, std::vector<double> &c) {   // c = a a (column j of c = a times column j of a)
        std::fill(c.begin(), c.end(), 0.0);
        for (int j = 0; j < N; ++j)
            for (int k = 0; k < N; ++k) {
                const double br = a[2 * ((size_t)j * N + k)], bi = a[2 * ((size_t)j * N + k) + 1];
                const double *ak = &a[2 * (size_t)k * N];
                double *cj = &c[2 * (size_t)j * N];
                for (int i = 0; i < N; ++i) {
                    cj[2 * i] += ak[2 * i] * br - ak[2 * i + 1] * bi;
                    cj[2 * i + 1] += ak[2 * i] * bi + ak[2 * i + 1] * br;
                }
            }
    };
    std::vector<double> m1(M, M + 2 * nn), m2(2 * nn), m4(2 * nn);
    const double b1 = norm1(m1);
    if (!(b1 > 0.0) || !std::isfinite(b1)) return b1;
    square(m1, m2);
    square(m2, m4);
    const double b2 = std::sqrt(norm1(m2)), b4 = std::sqrt(std::sqrt(norm1(m4)));
    double b = std::min(b1, std::min(b2, b4));
    if (!std::isfinite(b)) b = b1;
    return b * (1.0 + 1e-10 * N);   // rounding of the computed powers
}

#ifndef DERIV32_NTH
#define DERIV32_NTH 128
#endif
#ifndef DERIV16_NTH
#define DERIV16_NTH 64   // N <= 16: one wave per cell chain (4 columns per thread, quad reductions), many chains per CU
#endif
template <int NP>
hipError_t launch_deriv(const DerivArgs &a, int nblocks, hipStream_t s) {
    // 512 threads (8 column chunks per row) at N = 64: half the register tile per thread, so that
    // two blocks (4 waves per SIMD) hide the LDS broadcast latency; 256 threads otherwise.
    constexpr int NTH = NP == 64 ? 512 : (NP == 16 ? DERIV16_NTH : DERIV32_NTH);
    // the series kernel is instantiated per control count: zero-padded controls would cost real FMAs
    switch (a.L) {
        case 1: hipLaunchKernelGGL((deriv_kernel<NP, 1, NTH>), dim3(nblocks), dim3(NTH), 0, s, a); break;
        case 2: hipLaunchKernelGGL((deriv_kernel<NP, 2, NTH>), dim3(nblocks), dim3(NTH), 0, s, a); break;
        case 3: hipLaunchKernelGGL((deriv_kernel<NP, 3, NTH>), dim3(nblocks), dim3(NTH), 0, s, a); break;
        case 4: hipLaunchKernelGGL((deriv_kernel<NP, 4, NTH>), dim3(nblocks), dim3(NTH), 0, s, a); break;
        case 5: case 6: hipLaunchKernelGGL((deriv_kernel<NP, 6, NTH>), dim3(nblocks), dim3(NTH), 0, s, a); break;
        default: hipLaunchKernelGGL((deriv_kernel<NP, 8, NTH>), dim3(nblocks), dim3(NTH), 0, s, a); break;
    }
    return hipGetLastError();
}

template <int NP, int LMAX, bool CACHE, bool VLDS>
hipError_t launch_dm(const DerivMfmaArgs &a, int nblocks, hipStream_t s) {
    constexpr int NW = NP / 16 <= 8 ? NP / 16 : 8;
    const size_t lds = VLDS ? sizeof(double) * 2 * (1 + LMAX) * 2 * NP * 16 : 0;
    if (VLDS) {
        static LdsLimit lim;
        int dev = 0;
        hipGetDevice(&dev);
        hipError_t e = lim.ensure((const void *)deriv_mfma_kernel<NP, LMAX, CACHE, VLDS>, dev, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((deriv_mfma_kernel<NP, LMAX, CACHE, VLDS>), dim3(nblocks), dim3(NW * 64), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_deriv_mfma(int NP, const DerivMfmaArgs &a, int nblocks, hipStream_t s) {
    // the scratch holds (1 + LMAX) vectors per block; LMAX is the instantiated control count
    switch (NP) {
        case 48:
            if (a.L == 1) return launch_dm<48, 1, true, true>(a, nblocks, s);
            if (a.L == 2) return launch_dm<48, 2, true, true>(a, nblocks, s);
            if (a.L <= 4) return launch_dm<48, 4, false, false>(a, nblocks, s);
            return launch_dm<48, 8, false, false>(a, nblocks, s);
        case 64:  // L <= 2: vectors in LDS (2 x (1+L) x 16 KB), operators cached in registers
            if (a.L == 1) return launch_dm<64, 1, true, true>(a, nblocks, s);
            if (a.L == 2) return launch_dm<64, 2, true, true>(a, nblocks, s);
            if (a.L <= 4) return launch_dm<64, 4, false, false>(a, nblocks, s);   // 5 vectors x 2 do not fit in LDS
            return launch_dm<64, 8, false, false>(a, nblocks, s);
        case 128:
            if (a.L > 4) return hipErrorInvalidValue;
            if (a.L == 1) return launch_dm<128, 1, false, false>(a, nblocks, s);
            if (a.L == 2) return launch_dm<128, 2, false, false>(a, nblocks, s);
            return launch_dm<128, 4, false, false>(a, nblocks, s);
        case 256:
            if (a.L > 4) return hipErrorInvalidValue;
            if (a.L == 1) return launch_dm<256, 1, false, false>(a, nblocks, s);
            if (a.L == 2) return launch_dm<256, 2, false, false>(a, nblocks, s);
            return launch_dm<256, 4, false, false>(a, nblocks, s);
        default:
            return hipErrorInvalidValue;
    }
}

template <int NP, int LMAX, bool CACHE, bool STREAM = false>
hipError_t launch_d2(const Deriv2Args &a, int nblocks, hipStream_t s) {
    constexpr int NW = NP / 16 <= 8 ? NP / 16 : 8;
    const size_t lds = sizeof(double) * (NP > 256 ? 1 : 2) * 2 * NP * 16;   // (NP = 512: one vector block, see deriv2_kernel)
    static LdsLimit lim;
    int dev = 0;
    hipGetDevice(&dev);
    if (lds > 48 * 1024) {
        hipError_t e = lim.ensure((const void *)deriv2_kernel<NP, LMAX, CACHE, STREAM>, dev, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((deriv2_kernel<NP, LMAX, CACHE, STREAM>), dim3(nblocks), dim3(NW * 64), lds, s, a);
    return hipGetLastError();
}

// stream_l: the 1 + L products of an order one matrix at a time (deriv2_kernel STREAM_L): always for more than four controls
// (the all-at-once form spills there), for three and four controls when asked for (GRAPE_DERIV_STREAM=1)
hipError_t launch_deriv2(int NP, const Deriv2Args &a, int nblocks, hipStream_t s, bool stream_l = false, bool never = false) {
    switch (NP) {
        case 48:
            if (a.L == 1) return launch_d2<48, 1, true>(a, nblocks, s);
            if (a.L == 2) return launch_d2<48, 2, true>(a, nblocks, s);
            if (a.L <= 4) return stream_l ? launch_d2<48, 4, false, true>(a, nblocks, s) : launch_d2<48, 4, false>(a, nblocks, s);
            return never ? launch_d2<48, 8, false>(a, nblocks, s) : launch_d2<48, 8, false, true>(a, nblocks, s);
        case 64:
            if (a.L == 1) return launch_d2<64, 1, true>(a, nblocks, s);
            if (a.L == 2) return launch_d2<64, 2, true>(a, nblocks, s);
            if (a.L <= 4) return stream_l ? launch_d2<64, 4, false, true>(a, nblocks, s) : launch_d2<64, 4, false>(a, nblocks, s);
            return never ? launch_d2<64, 8, false>(a, nblocks, s) : launch_d2<64, 8, false, true>(a, nblocks, s);
        case 128:
            if (a.L == 1) return launch_d2<128, 1, false>(a, nblocks, s);
            if (a.L == 2) return launch_d2<128, 2, false>(a, nblocks, s);
            if (a.L > 4) return launch_d2<128, 8, false, true>(a, nblocks, s);
            return stream_l ? launch_d2<128, 4, false, true>(a, nblocks, s) : launch_d2<128, 4, false>(a, nblocks, s);
        case 256:
            if (a.L == 1) return launch_d2<256, 1, false>(a, nblocks, s);
            if (a.L == 2) return launch_d2<256, 2, false>(a, nblocks, s);
            if (a.L > 4) return launch_d2<256, 8, false, true>(a, nblocks, s);
            return stream_l ? launch_d2<256, 4, false, true>(a, nblocks, s) : launch_d2<256, 4, false>(a, nblocks, s);
        case 512:   // matrix-free propagator only: the products one matrix at a time (fewest live registers)
            if (a.L <= 2) return launch_d2<512, 2, false, true>(a, nblocks, s);
            if (a.L <= 4) return launch_d2<512, 4, false, true>(a, nblocks, s);
            return launch_d2<512, 8, false, true>(a, nblocks, s);
        default:
            return hipErrorInvalidValue;
    }
}

template <int NP, int LMAX>
hipError_t launch_ds(const DerivSubArgs &a, int nblocks, hipStream_t s) {
    constexpr int NW = NP / 16 <= 8 ? NP / 16 : 8;
    hipLaunchKernelGGL((deriv_sub_kernel<NP, LMAX>), dim3(nblocks), dim3(NW * 64), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_deriv_sub(int NP, const DerivSubArgs &a, int nblocks, hipStream_t s) {
    const int L = a.m.L;
#define DS_CASE(NP_)                                                         \
    case NP_:                                                                \
        if (L == 1) return launch_ds<NP_, 1>(a, nblocks, s);                 \
        if (L == 2) return launch_ds<NP_, 2>(a, nblocks, s);                 \
        if (L <= 4) return launch_ds<NP_, 4>(a, nblocks, s);                 \
        return launch_ds<NP_, 8>(a, nblocks, s);
    switch (NP) {
        DS_CASE(48) DS_CASE(64) DS_CASE(128) DS_CASE(256) DS_CASE(512)
        default: return hipErrorInvalidValue;
    }
#undef DS_CASE
}

// phases 0,1 belong to the forward call, 2,3,4 to the backward call, 5 to grape_eval
long phase_slot(grape_handle *h, int i) { return (i <= 1 ? h->n_fwd : (i <= 4 ? h->n_bwd : h->n_fwd)) % kRing; }
void phase_begin(grape_handle *h, int i, hipStream_t s) {
    if (h->capturing) return;   // (events recorded inside a stream capture cannot be read: the uncaptured evaluations time the phases)
    hipEventRecord(h->ph[phase_slot(h, i)][i].e0, s);
}
void phase_end(grape_handle *h, int i, hipStream_t s) {
    if (h->capturing) return;
    Phase &p = h->ph[phase_slot(h, i)][i];
    hipEventRecord(p.e1, s);
    p.used = true;
}


// ---- blocked Pade-13 for 64 < N <= 256: one launch per product, cells in chunks ----
LgView lg_full(double *p, int NP) { return LgView{p, (size_t)2 * NP * NP, (size_t)NP * NP, NP, 0, 0}; }

// the products of the polynomial route as assembly (asm/gen_lg.py: lg_gemm_asm; GRAPE_LG_ASM=0: the compiled kernel, its
// twin).  The assembly kernel takes whole planar arrays, at most two epilogue terms and no beta / identity / scaling terms;
// the squaring launches of the plan (copy-through of finished cells, device-side count) are covered, the Gauss-Jordan
// launches of the Pade route (block views, skipped rows, scalings) stay with lg_gemm_kernel.
struct LgAsmArgs {
    const double *X, *Y;
    double *C, *C2;
    const double *Add0, *Add1;
    double2 *Uout;
    const int *smax_ptr;
    double coef[2], coef2[2], cI, cI2;
    int NP, NB, ncell, herm, nadd, u_if_smax0, per_cell;
    unsigned magic_pc, magic_nb;   // floor(2^32 / d) + 1 of per_cell and NB: the kernel divides by one multiplication
    int pad;
    const int *s_cell;             // squaring launches: cells with s_cell[cell] <= sq_iter are copied through
    int sq_iter, sq_mode;          // sq_mode: leave at once when sq_iter >= *smax_ptr, write U when sq_iter == *smax_ptr - 1
    // the combinations of the polynomial route from the epilogue of the launch that writes the last power (asm/gen_lg.py, comb)
    int comb_mode, pad2;           // bit 0: on; bit 1: the second column sum is |this product| (A6), else |P3| (A3)
    const double *P1, *P2, *P3;    // A, A2, A3
    double *B1, *B5, *B4, *B3, *B2;
    double *colpart;
    double ca[3], ce[3], cd[5], cc[5], cb[5];
};
static_assert(sizeof(LgAsmArgs) == 416, "argument block of lg_gemm_asm");
// what the launch of the last power needs to form B1 .. B5 (expm_large_t18)
struct LgComb {
    const LgT18OperandsArgs *o;
    double *colpart;
    int q_is_a6;
    int lite;                      // round 6: B4, B3, B2 are left to the launches that add them (LgPow)
};
// round 6: the epilogue terms of a launch as combinations of the powers, formed in its epilogue (asm/gen_lg.py power_adds):
// C += c1 . (1, A, A2, A3, A6); second output C2 = C + c2 . (...).  Cells with s_cell > 0 read the launch's Add arrays.
struct LgPow {
    const double *A, *A2, *A3, *A6;
    const int *s_cell;
    double c1[5], c2[5];
};
bool lg_asm_eligible(const grape_handle *h, int NP, int nc, int per_cell) {
    return h->lg_asm && (NP == 128 || NP == 256) && (long)((nc + 7) / 8) * 8 * per_cell < (1L << 24);
}
bool lg_full_view(const LgView &v, int NP) {
    return v.p && v.rb == 0 && v.cb == 0 && v.ld == NP && v.plane == (size_t)NP * NP && v.cell_stride == (size_t)2 * NP * NP;
}
// true: launched (err holds the status); false: not eligible
bool lg_try_asm(const grape_handle *h, hipStream_t s, const LgGemmArgs &a, hipError_t *err, const LgComb *comb = nullptr,
                const LgPow *pow = nullptr) {
    if (!h->lg_asm) return false;   // (GRAPE_LG_ASM, read once in grape_create: the route of a handle never changes)
    const int NP = a.C.ld, NB = a.nbi;
    if ((NP != 128 && NP != 256) || a.nbj != NB || a.kblocks != NB || NB * 64 != NP) return false;
    if (a.alpha != 1.0 || a.beta != 0.0 || a.cI != 0.0 || a.cI2 != 0.0 || a.scale_s || a.skip_bi != -1 || a.nadd > 2) return false;
    const bool squaring = a.s_cell != nullptr;                  // (launch `sq_iter` of the squaring plan)
    if (squaring && (a.nadd || a.herm || a.C2.p || a.u_if_smax0)) return false;
    if (a.smax_ptr && !a.u_if_smax0 && !squaring) return false;
    if (a.herm && (a.nadd || a.Uout || a.C2.p)) return false;
    if (a.Uout && a.u_np != NP) return false;
    if (!lg_full_view(a.X, NP) || !lg_full_view(a.Y, NP) || !lg_full_view(a.C, NP)) return false;
    if (a.C2.p && !lg_full_view(a.C2, NP)) return false;
    for (int i = 0; i < a.nadd; ++i)
        if (!lg_full_view(a.Add[i], NP)) return false;
    LgAsmArgs k{};
    k.X = a.X.p; k.Y = a.Y.p; k.C = a.C.p; k.C2 = a.C2.p;
    k.Add0 = a.nadd > 0 ? a.Add[0].p : nullptr; k.Add1 = a.nadd > 1 ? a.Add[1].p : nullptr;
    k.Uout = a.Uout; k.smax_ptr = a.smax_ptr;
    for (int i = 0; i < a.nadd; ++i) { k.coef[i] = a.coef[i]; k.coef2[i] = a.coef2[i]; }
    k.NP = NP; k.NB = NB; k.ncell = a.ncell; k.herm = a.herm; k.nadd = a.nadd; k.u_if_smax0 = a.u_if_smax0;
    k.per_cell = a.herm ? NB * (NB + 1) / 2 : NB * NB;
    k.s_cell = a.s_cell; k.sq_iter = a.sq_iter; k.sq_mode = (squaring && a.smax_ptr) ? 1 : 0;
    k.magic_pc = (unsigned)((1ull << 32) / (unsigned)k.per_cell + 1);
    k.magic_nb = (unsigned)((1ull << 32) / (unsigned)NB + 1);
    if (comb) {
        if (a.nadd || a.Uout || a.C2.p || squaring) return false;
        const LgT18OperandsArgs &o = *comb->o;
        k.comb_mode = 1 | (comb->q_is_a6 ? 2 : 0) | (comb->lite ? 8 : 0);
        k.P1 = o.A; k.P2 = o.A2; k.P3 = o.A3; k.B1 = o.B1; k.B5 = o.B5; k.B4 = o.B4; k.B3 = o.B3; k.B2 = o.B2;
        k.colpart = comb->colpart;
        memcpy(k.ca, o.a, sizeof(k.ca)); memcpy(k.ce, o.e, sizeof(k.ce)); memcpy(k.cd, o.d, sizeof(k.cd));
        memcpy(k.cc, o.c, sizeof(k.cc)); memcpy(k.cb, o.b, sizeof(k.cb));
    }
    if (pow) {
        if (comb || !a.nadd || squaring || a.herm) return false;
        k.comb_mode = 4;
        k.P1 = pow->A; k.P2 = pow->A2; k.P3 = pow->A3;
        k.B1 = const_cast<double *>(pow->A6);                       // (slots of the argument block: gen_lg.py power_adds)
        k.B5 = reinterpret_cast<double *>(const_cast<int *>(pow->s_cell));
        memcpy(k.cd, pow->c1, sizeof(k.cd)); memcpy(k.cc, pow->c2, sizeof(k.cc));
    }
    const int groups = (a.ncell + 7) / 8;
    if ((long)groups * 8 * k.per_cell >= (1L << 24)) return false;
    *err = (hipError_t)grape_lg_asm_launch(&k, sizeof(k), (unsigned)(groups * 8 * k.per_cell), (void *)s);
    return true;
}

hipError_t lg_gemm(const grape_handle *h, hipStream_t s, int nc, int nbi, int nbj, LgView X, LgView Y, LgView C, int kblocks, double alpha,
                   double beta, int nadd = 0, const LgView *add = nullptr, const double *coef = nullptr,
                   double cI = 0.0, const int *s_cell = nullptr, int sq_iter = 0, int herm = 0,
                   const int *scale_s = nullptr, int scale_pow = 0, double2 *Uout = nullptr, int u_np = 0,
                   int skip_bi = -1, const int *smax_ptr = nullptr, const LgComb *comb = nullptr) {
    if (nbi <= 0 || nbj <= 0) return hipSuccess;
    LgGemmArgs a{};
    a.X = X; a.Y = Y; a.C = C; a.kblocks = kblocks; a.alpha = alpha; a.beta = beta; a.cI = cI;
    a.nadd = nadd;
    for (int i = 0; i < nadd; ++i) { a.Add[i] = add[i]; a.coef[i] = coef[i]; }
    a.s_cell = s_cell; a.sq_iter = sq_iter; a.herm = (nbi == nbj) ? herm : 0;
    a.scale_s = scale_s; a.scale_pow = scale_pow; a.Uout = Uout; a.u_np = u_np; a.skip_bi = skip_bi;
    a.smax_ptr = smax_ptr;
    a.nbi = nbi; a.nbj = nbj; a.ncell = nc;
    const int groups = (nc + 7) / 8;   // cells are dealt to the 8 XCDs in groups
    const int per_cell = a.herm ? nbi * (nbi + 1) / 2 : nbi * nbj;
    hipError_t easm;
    if (lg_try_asm(h, s, a, &easm, comb)) return easm;
    if (comb) return hipErrorInvalidValue;   // (only the assembly kernel has that epilogue: the caller asks lg_asm_eligible first)
    hipLaunchKernelGGL(lg_gemm_kernel, dim3(groups * 8 * per_cell), dim3(256), 0, s, a);
    return hipGetLastError();
}

// the polynomial route's products: up to four epilogue terms with powers of the per-cell scaling, optional second output
hipError_t lg_gemm_poly(const grape_handle *h, hipStream_t s, int nc, int NB, LgView X, LgView Y, LgView C, int herm, int nadd, const LgView *add,
                        const double *coef, const int *add_pow, double cI, const int *scale_s,
                        const LgView *C2 = nullptr, const double *coef2 = nullptr, double cI2 = 0.0,
                        double2 *Uout = nullptr, int u_np = 0, const int *smax_ptr = nullptr, const LgPow *pow = nullptr) {
    LgGemmArgs a{};
    a.X = X; a.Y = Y; a.C = C; a.kblocks = NB; a.alpha = 1.0; a.beta = 0.0; a.cI = cI;
    a.nadd = nadd;
    for (int i = 0; i < nadd; ++i) {
        a.Add[i] = add[i]; a.coef[i] = coef[i]; a.add_pow[i] = add_pow[i];
        a.coef2[i] = coef2 ? coef2[i] : 0.0;
    }
    if (C2) { a.C2 = *C2; a.cI2 = cI2; }
    a.herm = herm; a.scale_s = scale_s; a.scale_pow = 0; a.skip_bi = -1;
    if (Uout) { a.Uout = Uout; a.u_np = u_np; a.smax_ptr = smax_ptr; a.u_if_smax0 = 1; }
    a.nbi = NB; a.nbj = NB; a.ncell = nc;
    const int groups = (nc + 7) / 8;
    const int per_cell = a.herm ? NB * (NB + 1) / 2 : NB * NB;
    hipError_t easm;
    if (lg_try_asm(h, s, a, &easm, nullptr, pow)) return easm;
    if (pow) return hipErrorInvalidValue;   // (only the assembly kernel forms its terms from the powers: the caller asks lg_asm_eligible first)
    hipLaunchKernelGGL(lg_gemm_kernel, dim3(groups * 8 * per_cell), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t lg_lincomb(hipStream_t s, double *out, size_t n, int nin, const double *const *in, const double *coef) {
    LgLincombArgs a{};
    a.out = out; a.n = n; a.nin = nin;
    for (int i = 0; i < nin; ++i) { a.in[i] = in[i]; a.coef[i] = coef[i]; }
    hipLaunchKernelGGL(lg_lincomb_kernel, dim3(2048), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t lg_lincomb2(hipStream_t s, double *out0, double *out1, size_t n, int nin, const double *const *in,
                       const double *c0, const double *c1) {
    LgLincomb2Args a{};
    a.out0 = out0; a.out1 = out1; a.n = n; a.nin = nin;
    for (int i = 0; i < nin; ++i) { a.in[i] = in[i]; a.c0[i] = c0[i]; a.c1[i] = c1[i]; }
    hipLaunchKernelGGL(lg_lincomb2_kernel, dim3(2048), dim3(256), 0, s, a);
    return hipGetLastError();
}

#define LGCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return _e; } while (0)

// Blocked path, polynomial route (default; GRAPE_EXPM_T18=0 selects the order-13 Pade route below): five products and no
// solve.  Hermitian generators: Chebyshev coefficient set, three of the five products compute the upper block triangle
// only, scaling from the spectral bound (at C5 no squaring at all where ||A||_1 = 8.3 costs the Pade route one).
// General matrices: Taylor coefficient set, scaling from alpha = min(||A||_1, max(d2, d3)).
hipError_t expm_large_t18(grape_handle *h, hipStream_t s) {
    const int NP = h->NP, NB = NP / 64;
    const size_t pp = (size_t)NP * NP;
    const long ncell = (long)h->KC * h->N_T;
    // nine chunk buffers: the four powers, then the five combinations; the powers are dead once those are formed and
    // their buffers take A9, B3 + A9 and the result
    const bool hm = h->herm;
    hipStream_t const s_main = s;
    // executed matrix instructions per cell (all waves of all workgroups; a 64-block product of one output block is 4 waves
    // x 192 instructions): full products NB^3 blocks, triangular ones NB^2 (NB + 1) / 2
    const unsigned long long blk = 4ull * 192ull;
    const unsigned long long gen = blk * NB * NB * NB, tri = blk * NB * NB * (NB + 1) / 2;
    if (h->d_Sf) {   // S_n = sum_l eps_ln shape_ln H_l for every time step (lg_form_kernel then reads H0_k and S_n)
        CtrlSumArgs ca{};
        ca.Hcf = h->d_Hcf; ca.eps = h->d_eps; ca.shape = h->d_shape; ca.Sf = h->d_Sf;
        ca.L = h->L; ca.N_T = h->N_T; ca.pp2 = 2 * NP * NP;
        hipLaunchKernelGGL(ctrl_sum_kernel, dim3(h->N_T, std::max(1, ca.pp2 / 2 / 2048)), dim3(256), 0, s, ca);
        LGCHK(hipGetLastError());
    }
    if (h->lg_lanes == 2) {   // the second lane starts behind what this stream has enqueued so far (pulses, S_n, cleared flags)
        LGCHK(hipMemsetAsync(h->d_smax2, 0, sizeof(int), s_main));
        LGCHK(hipEventRecord(h->lg_ev_fork, s_main));
        LGCHK(hipStreamWaitEvent(h->lg_stream2, h->lg_ev_fork, 0));
    }
    long ichunk = 0;
    for (long c0_ = 0; c0_ < ncell; c0_ += h->chunk, ++ichunk) {
        const int nc = (int)std::min<long>(h->chunk, ncell - c0_);
        const bool lane2 = h->lg_lanes == 2 && (ichunk & 1);
        double *const *lg = lane2 ? h->d_lg2 : h->d_lg;
        double *A = lg[0], *A2 = lg[1], *A3 = lg[2], *A6 = lg[3], *B1 = lg[4], *B5 = lg[5], *B4 = lg[6], *B3 = lg[7], *B2 = lg[8];
        const int nc_ = (int)std::min<long>(h->chunk, ncell - c0_);
        // round 6: B4, B3, B2 are formed by the launches that add them, from the powers (the launch of A6 then writes three
        // arrays instead of six: it is HBM-bound, the two general products have bandwidth to spare).  The powers have to
        // survive until the last product: A9 and B3 + A9 take the buffers of B4 and B3 (a cell that needed a scaling reads
        // its B4 / B3 block and then overwrites it, in the same workgroup).  Measured at the C5 shard (GRAPE_LG_POW=1, same
        // box): the launch of A6 1.33 -> 1.13 ms per chunk, the two general products 1.09 -> 1.22 ms each -- the epilogue's eight
        // serial load groups cost what the HBM-bound launch saves (phase A 124.5 / 126.4 -> 126.4 / 126.3 ms).  Off by default.
        const bool power = h->lg_pow && h->lg_spec && h->lg_fuse && lg_asm_eligible(h, NP, nc_, hm ? NB * (NB + 1) / 2 : NB * NB) &&
                           lg_asm_eligible(h, NP, nc_, NB * NB);
        double *A9 = power ? B4 : A, *Lm = power ? B3 : A2, *T = A3;
        int *const d_scell = lane2 ? h->d_scell2 : h->d_scell;
        double *const d_dinv = lane2 ? h->d_dinv2 : h->d_dinv, *const d_colpart = lane2 ? h->d_colpart2 : h->d_colpart;
        int *const d_smax = lane2 ? h->d_smax2 : h->d_flags + 1;
        s = lane2 ? h->lg_stream2 : s_main;
        const size_t nel = (size_t)nc * 2 * pp;
        LgFormArgs fa{};
        fa.H0f = h->d_H0f; fa.Hcf = h->d_Hcf; fa.eps = h->d_eps; fa.shape = h->d_shape; fa.dts = h->d_dts;
        fa.A = A; fa.s_cell = d_scell; fa.stats = h->d_stats; fa.flags = h->d_flags;
        fa.NP = NP; fa.L = h->L; fa.N_T = h->N_T; fa.hc_per_traj = h->p.hc_per_traj; fa.cell0 = (int)c0_; fa.rep = h->d_rep;
        fa.norm1 = d_dinv;   // (the inverse slots of the Pade route are idle here: ||A||_1 per cell)
        fa.Sf = h->d_Sf;
        if (h->lg_form2) {
            double *np_ = lane2 ? h->d_normpart2 : h->d_normpart;
            const dim3 grid(NP / LG_FORM_ROWS, (nc + 15) / 16);
            if (h->L <= 2) hipLaunchKernelGGL(lg_form2_kernel<2>, grid, dim3(256), 0, s, fa, nc, np_);
            else hipLaunchKernelGGL(lg_form2_kernel<4>, grid, dim3(256), 0, s, fa, nc, np_);
            LGCHK(hipGetLastError());
            hipLaunchKernelGGL(lg_norm1_kernel, dim3(nc), dim3(256), 0, s, fa, (const double *)np_);
        } else {
            hipLaunchKernelGGL(lg_form_kernel, dim3(nc), dim3(1024), 0, s, fa);
        }
        LGCHK(hipGetLastError());
        const LgView vA = lg_full(A, NP), vA2 = lg_full(A2, NP), vA3 = lg_full(A3, NP), vA6 = lg_full(A6, NP),
                     vB1 = lg_full(B1, NP), vB5 = lg_full(B5, NP), vA9 = lg_full(A9, NP), vL = lg_full(Lm, NP), vT = lg_full(T, NP);
        LGCHK(lg_gemm(h, s, nc, NB, NB, vA, vA, vA2, NB, 1.0, 0.0, 0, nullptr, nullptr, 0.0, nullptr, 0, hm ? 1 : 0));     // A2 = A A
        LGCHK(lg_gemm(h, s, nc, NB, NB, vA2, vA, vA3, NB, 1.0, 0.0, 0, nullptr, nullptr, 0.0, nullptr, 0, hm ? -1 : 0));   // A3 = A2 A
        LgT18ScaleArgs sa{};
        sa.P = A2; sa.Q = hm ? A6 : A3; sa.qpow = hm ? 6 : 3; sa.norm1 = hm ? nullptr : d_dinv;
        sa.s_cell = d_scell; sa.flags = h->d_flags; sa.smax = d_smax; sa.stats = h->d_stats; sa.NP = NP;
        sa.theta = hm ? T18_THETA : T18T_THETA;
        sa.mfma_per_cell = (hm ? 3 * tri : 3 * gen) + 2 * gen; sa.mfma_per_sq = gen;
        LgT18OperandsArgs oa{};
        oa.A = A; oa.A2 = A2; oa.A3 = A3; oa.A6 = A6; oa.B1 = B1; oa.B5 = B5; oa.B4 = B4; oa.B3 = B3; oa.B2 = B2;
        oa.s_cell = d_scell; oa.NP = NP; oa.per_cell = 2 * pp; oa.n = nel;
        if (hm) {
            const double a_[3] = {T18_A1, T18_A2, T18_A3}, e_[3] = {T18_E2, T18_E3, T18_E6}, b_[5] = {T18_B0, T18_B1, T18_B2, T18_B3, T18_B6};
            const double c_[5] = {T18_C0, T18_C1, T18_C2, T18_C3, T18_C6}, d_[5] = {T18_D0, T18_D1, T18_D2, T18_D3, T18_D6};
            memcpy(oa.a, a_, sizeof(a_)); memcpy(oa.e, e_, sizeof(e_)); memcpy(oa.b, b_, sizeof(b_));
            memcpy(oa.c, c_, sizeof(c_)); memcpy(oa.d, d_, sizeof(d_));
        } else {
            const double a_[3] = {T18T_A1, T18T_A2, T18T_A3}, e_[3] = {T18T_E2, T18T_E3, T18T_E6}, b_[5] = {T18T_B0, T18T_B1, T18T_B2, T18T_B3, T18T_B6};
            const double c_[5] = {T18T_C0, T18T_C1, T18T_C2, T18T_C3, T18T_C6}, d_[5] = {T18T_D0, T18T_D1, T18T_D2, T18T_D3, T18T_D6};
            memcpy(oa.a, a_, sizeof(a_)); memcpy(oa.e, e_, sizeof(e_)); memcpy(oa.b, b_, sizeof(b_));
            memcpy(oa.c, c_, sizeof(c_)); memcpy(oa.d, d_, sizeof(d_));
        }
        // round 5: the launch of A6 forms the combinations in its epilogue, for s = 0, with the column sums of the decision
        // (Hermitian generators: a workgroup of the upper block triangle also forms those of the mirrored block)
        const bool fused = h->lg_spec && h->lg_fuse && lg_asm_eligible(h, NP, nc, hm ? NB * (NB + 1) / 2 : NB * NB);
        {
            const LgComb cb{&oa, d_colpart, hm ? 1 : 0, (fused && power) ? 1 : 0};
            LGCHK(lg_gemm(h, s, nc, NB, NB, vA3, vA3, vA6, NB, 1.0, 0.0, 0, nullptr, nullptr, 0.0, nullptr, 0, hm ? 1 : 0,    // A6 = A3 A3
                          nullptr, 0, nullptr, 0, -1, nullptr, fused ? &cb : nullptr));
        }
        if (!h->lg_spec) {
            hipLaunchKernelGGL(lg_t18_scale_kernel, dim3(nc), dim3(256), 0, s, sa);
            LGCHK(hipGetLastError());
        }
        if (h->lg_spec) {
            // combinations for s = 0 with the column sums of A2 and A6 / A3 on the way, the decision, and the combinations once
            // more for the cells that need a scaling (none at the benchmark's norms: that launch leaves at once)
            LgT18Operands2Args o2{};
            o2.o = oa; o2.colpart = d_colpart; o2.q_is_a6 = hm ? 1 : 0; o2.redo = 0;
            if (!fused) {
                hipLaunchKernelGGL(lg_t18_operands2_kernel, dim3((unsigned)nc * LG_PARTS), dim3(256), 0, s, o2);
                LGCHK(hipGetLastError());
            }
            LgT18DecideArgs da{};
            da.colpart = d_colpart; da.s = sa; da.nparts = fused ? NB : LG_PARTS;
            if (h->deriv_econ && h->d_celldeg && hm) {
                da.cell_deg = h->d_celldeg; da.cell0 = (int)c0_; da.econ_n = 0;
                for (int i = 0; i < ECON_NSETS && da.econ_n < 4; ++i)
                    if (ECON_THETAS[i] <= T18_THETA) { da.econ_theta[da.econ_n] = ECON_THETAS[i]; da.econ_deg[da.econ_n++] = ECON_DEG[i]; }
            }
            hipLaunchKernelGGL(lg_t18_decide_kernel, dim3(nc), dim3(256), 0, s, da);
            LGCHK(hipGetLastError());
            o2.redo = 1;
            hipLaunchKernelGGL(lg_t18_operands2_kernel, dim3((unsigned)nc * LG_PARTS), dim3(256), 0, s, o2);
            LGCHK(hipGetLastError());
        } else {
            hipLaunchKernelGGL(lg_t18_operands_kernel, dim3(2048), dim3(256), 0, s, oa);
            LGCHK(hipGetLastError());
        }
        const LgView vB4 = lg_full(B4, NP), vB3 = lg_full(B3, NP), vB2 = lg_full(B2, NP);
        const int *smax_ptr = d_smax;
        {   // A9 = B1 B5 + B4 and, from the same launch, B3 + A9 (the left operand of the last product)
            const LgView add[2] = {vB4, vB3};
            const double c1[2] = {1.0, 0.0}, c2[2] = {0.0, 1.0};
            const int pw0[2] = {0, 0};
            LgPow pw{A, A2, A3, A6, d_scell, {0}, {0}};
            memcpy(pw.c1, oa.d, sizeof(pw.c1)); memcpy(pw.c2, oa.c, sizeof(pw.c2));
            LGCHK(lg_gemm_poly(h, s, nc, NB, vB1, vB5, vA9, 0, 2, add, c1, pw0, 0.0, nullptr, &vL, c2, 0.0, nullptr, 0, nullptr,
                               (fused && power) ? &pw : nullptr));
        }
        {   // p = B2 + (B3 + A9) A9: straight into U_kn unless a cell of this evaluation needs a squaring
            const LgView add[1] = {vB2};
            const double c1[1] = {1.0};
            const int pw0[1] = {0};
            LgPow pw{A, A2, A3, A6, d_scell, {0}, {0}};
            memcpy(pw.c1, oa.b, sizeof(pw.c1));
            LGCHK(lg_gemm_poly(h, s, nc, NB, vL, vA9, vT, 0, 1, add, c1, pw0, 0.0, nullptr, nullptr, nullptr, 0.0,
                               h->d_U + (size_t)c0_ * pp, NP, smax_ptr, (fused && power) ? &pw : nullptr));
        }
        // squarings by the launch plan (see expm_large): every launch exits at once when it is not needed, the last
        // needed one writes U_kn for all cells of the chunk (cells that are done are copied through)
        double *X = T, *Y = B1;
        for (int it = 0; it < h->sq_plan; ++it) {
            LGCHK(lg_gemm(h, s, nc, NB, NB, lg_full(X, NP), lg_full(X, NP), lg_full(Y, NP), NB, 1.0, 0.0, 0, nullptr, nullptr,
                          0.0, d_scell, it, 0, nullptr, 0, h->d_U + (size_t)c0_ * pp, NP, -1, smax_ptr));
            std::swap(X, Y);
        }
    }
    s = s_main;
    if (h->lg_lanes == 2) {
        LGCHK(hipEventRecord(h->lg_ev_join, h->lg_stream2));
        LGCHK(hipStreamWaitEvent(s, h->lg_ev_join, 0));
    }
    hipLaunchKernelGGL(lg_plan_check_kernel, dim3(1), dim3(1), 0, s, h->d_flags, h->sq_plan, h->lg_lanes == 2 ? h->d_smax2 : nullptr);
    LGCHK(hipGetLastError());
    return hipSuccess;
}

hipError_t expm_large(grape_handle *h, hipStream_t s) {
    const int NP = h->NP, NB = NP / 64;
    const size_t pp = (size_t)NP * NP;
    const long ncell = (long)h->KC * h->N_T;
    static LdsLimit lim_inv;
    const size_t inv_lds = sizeof(double) * (3 * 2 * 64 * 18 + 1536);
    LGCHK(lim_inv.ensure((const void *)lg_inv64_kernel, h->device, inv_lds));
    double *A = h->d_lg[0], *A2 = h->d_lg[1], *A4 = h->d_lg[2], *A6 = h->d_lg[3], *W = h->d_lg[4], *Z = h->d_lg[5],
           *T = h->d_lg[6], *V = h->d_lg[7], *Uo = h->d_lg[8];
    for (long c0 = 0; c0 < ncell; c0 += h->chunk) {
        const int nc = (int)std::min<long>(h->chunk, ncell - c0);
        const size_t nel = (size_t)nc * 2 * pp;
        LGCHK(hipMemsetAsync(h->d_scell + h->chunk, 0, sizeof(int), s));
        LGCHK(hipMemsetAsync(h->d_cellflag, 0, (size_t)nc * sizeof(int), s));
        LgFormArgs fa{};
        fa.H0f = h->d_H0f; fa.Hcf = h->d_Hcf; fa.eps = h->d_eps; fa.shape = h->d_shape; fa.dts = h->d_dts;
        fa.A = A; fa.s_cell = h->d_scell; fa.stats = h->d_stats; fa.flags = h->d_flags;
        fa.NP = NP; fa.L = h->L; fa.N_T = h->N_T; fa.hc_per_traj = h->p.hc_per_traj; fa.cell0 = (int)c0; fa.rep = h->d_rep;
        hipLaunchKernelGGL(lg_form_kernel, dim3(nc), dim3(1024), 0, s, fa);
        LGCHK(hipGetLastError());
        const LgView vA = lg_full(A, NP), vA2 = lg_full(A2, NP), vA4 = lg_full(A4, NP), vA6 = lg_full(A6, NP),
                     vW = lg_full(W, NP), vZ = lg_full(Z, NP), vT = lg_full(T, NP), vV = lg_full(V, NP), vU = lg_full(Uo, NP);
        // Hermitian generators: A2, A4, A6, T, V are Hermitian and U = A T skew-Hermitian -> upper block triangle only
        const int hm = h->herm ? 1 : 0;
        // A is stored unscaled (lg_form_kernel): A2 = (A / 2^s)^2 and U = (A / 2^s) T take the power of two as a factor
        LGCHK(lg_gemm(h, s, nc, NB, NB, vA, vA, vA2, NB, 1.0, 0.0, 0, nullptr, nullptr, 0.0, nullptr, 0, hm, h->d_scell, 2));
        LGCHK(lg_gemm(h, s, nc, NB, NB, vA2, vA2, vA4, NB, 1.0, 0.0, 0, nullptr, nullptr, 0.0, nullptr, 0, hm));
        LGCHK(lg_gemm(h, s, nc, NB, NB, vA2, vA4, vA6, NB, 1.0, 0.0, 0, nullptr, nullptr, 0.0, nullptr, 0, hm));
        {
            const double *in[3] = {A6, A4, A2};
            const double cw[3] = {B13_13, B13_11, B13_9}, cz[3] = {B13_12, B13_10, B13_8};
            LGCHK(lg_lincomb2(s, W, Z, nel, 3, in, cw, cz));
            const LgView add[3] = {vA6, vA4, vA2};
            const double ct[3] = {B13_7, B13_5, B13_3}, cv[3] = {B13_6, B13_4, B13_2};
            LGCHK(lg_gemm(h, s, nc, NB, NB, vA6, vW, vT, NB, 1.0, 0.0, 3, add, ct, B13_1, nullptr, 0, hm));   // T = A6 W1 + T0
            LGCHK(lg_gemm(h, s, nc, NB, NB, vA6, vZ, vV, NB, 1.0, 0.0, 3, add, cv, B13_0, nullptr, 0, hm));   // V = A6 Z1 + V0
        }
        LGCHK(lg_gemm(h, s, nc, NB, NB, vA, vT, vU, NB, 1.0, 0.0, 0, nullptr, nullptr, 0.0, nullptr, 0, -hm, h->d_scell, 1));   // U = A T
        {
            const double *in[2] = {V, Uo};
            const double cp[2] = {1.0, 1.0}, cq[2] = {1.0, -1.0};
            LGCHK(lg_lincomb2(s, W, Z, nel, 2, in, cp, cq));   // P = V + U (buffer W), Q = V - U (buffer Z)
        }
        // block Gauss-Jordan on 64-blocks: Q X = P
        const LgView vD{h->d_dinv, (size_t)2 * 4096, (size_t)4096, 64, 0, 0};
        for (int jb = 0; jb < NB; ++jb) {
            LgInvArgs ia{};
            ia.Q = vZ; ia.Q.rb = jb; ia.Q.cb = jb; ia.Dinv = h->d_dinv; ia.flags = h->d_flags;
            ia.inv_scale2 = 1.0 / (B13_0 * B13_0); ia.cellflag = h->d_cellflag;
            hipLaunchKernelGGL(lg_inv64_kernel, dim3(nc), dim3(256), inv_lds, s, ia);
            LGCHK(hipGetLastError());
            LgView qrow = vZ; qrow.rb = jb; qrow.cb = jb + 1;
            LgView prow = vW; prow.rb = jb; prow.cb = 0;
            LGCHK(lg_gemm(h, s, nc, 1, NB - 1 - jb, vD, qrow, qrow, 1, 1.0, 0.0));   // Q[jb][jb+1..] = Dinv Q[jb][..]
            LGCHK(lg_gemm(h, s, nc, 1, NB, vD, prow, prow, 1, 1.0, 0.0));             // P[jb][:]      = Dinv P[jb][:]
            {   // trailing update of every block row tr != jb in one launch per matrix (the launch skips row jb):
                // Q[tr][jb+1..] -= Q[tr][jb] Q[jb][jb+1..],  P[tr][:] -= Q[tr][jb] P[jb][:]
                LgView x = vZ; x.rb = 0; x.cb = jb;
                LgView cq = vZ; cq.rb = 0; cq.cb = jb + 1;
                LgView cp = vW; cp.rb = 0; cp.cb = 0;
                LGCHK(lg_gemm(h, s, nc, NB, NB - 1 - jb, x, qrow, cq, 1, -1.0, 1.0, 0, nullptr, nullptr, 0.0, nullptr, 0, 0,
                              nullptr, 0, nullptr, 0, jb));
                LGCHK(lg_gemm(h, s, nc, NB, NB, x, prow, cp, 1, -1.0, 1.0, 0, nullptr, nullptr, 0.0, nullptr, 0, 0,
                              nullptr, 0, nullptr, 0, jb));
            }
        }
        {   // cells whose unpivoted elimination was unsafe: partial pivoting from the intact V, U
            LgPivArgs pa{};
            pa.V = V; pa.Uo = Uo; pa.P = W; pa.Q = Z; pa.cellflag = h->d_cellflag; pa.flags = h->d_flags;
            pa.stats = h->d_stats; pa.NP = NP; pa.ncell = nc;
            hipLaunchKernelGGL(lg_pivoted_kernel, dim3(std::min(nc, 256)), dim3(1024), 0, s, pa);
            LGCHK(hipGetLastError());
        }
        // squarings (per-cell count; cells that are done are copied through).  The largest count is only known on the
        // device (flags[1]); instead of reading it back -- a host synchronisation per chunk -- the host issues a PLAN of
        // h->sq_plan squaring launches: every launch compares its index with the device-side count, exits at once when
        // it is not needed, and the last needed one writes U_kn (interleaved complex) itself.  A plan that turns out
        // too short flags the evaluation (bit 5): grape_check adapts the plan and the call is repeated.
        const int *smax_ptr = h->d_flags + 1;
        double *X = W, *Y = T;
        for (int it = 0; it < h->sq_plan; ++it) {
            LGCHK(lg_gemm(h, s, nc, NB, NB, lg_full(X, NP), lg_full(X, NP), lg_full(Y, NP), NB, 1.0, 0.0, 0, nullptr, nullptr,
                          0.0, h->d_scell, it, 0, nullptr, 0, h->d_U + (size_t)c0 * pp, NP, -1, smax_ptr));
            std::swap(X, Y);
        }
        hipLaunchKernelGGL(lg_store_u_kernel, dim3(2048), dim3(256), 0, s, (const double *)W,
                           h->d_U + (size_t)c0 * pp, NP, (size_t)nc * pp, smax_ptr);
        LGCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(lg_plan_check_kernel, dim3(1), dim3(1), 0, s, h->d_flags, h->sq_plan, (const int *)nullptr);
    LGCHK(hipGetLastError());
    return hipSuccess;
}

SeriesArgs series_args(grape_handle *h, const SweepArgs &sa, bool backward) {
    SeriesArgs ra{};
    ra.s = sa;
    ra.H0 = backward ? h->d_H0t : h->d_H0f;
    ra.Hc = backward ? h->d_Hct : h->d_Hcf;
    ra.eps = h->d_eps; ra.shape = h->d_shape; ra.dts = h->d_dts; ra.rb = h->d_rb; ra.stats = h->d_stats;
    ra.tol = h->series_tol; ra.theta = h->series_theta;
    ra.L = h->L; ra.hc_per_traj = h->p.hc_per_traj; ra.max_order = 200;
    if (!backward) { ra.park = h->d_gpark; ra.morder = h->d_morder; ra.maxp = h->maxp; }
    return ra;
}

// Cooperative polynomial sweeps for 64 < N <= 256 (grape_cheby.hip.h): the trajectories go through the kernel in rounds
// of h->cheby_round; `sb` != nullptr adds the backward sweeps of the same trajectories to every launch.
hipError_t launch_cheby(grape_handle *h, const SweepArgs *sf, const SweepArgs *sb, hipStream_t s) {
    const int NP = h->NP, S = h->cheby_S, K = h->K;
    const size_t lds = sizeof(double) * ((size_t)2 * 16 * (NP + 2) + 2 * (size_t)NP + CHEBY_MAXT + 2);
    static LdsLimit lim;
    {
        hipError_t e = lim.ensure((const void *)cheby_coop_kernel, h->device, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(cheby_arm_kernel, dim3(256), dim3(256), 0, s, (unsigned long long *)h->d_xch, (size_t)2 * K * 4 * NP * 2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(h->d_xcc, 0xFF, (size_t)2 * K * 32 * sizeof(int), s);
    if (e != hipSuccess) return e;
    auto fill = [&](ChebyArgs &c, const SweepArgs &sa, bool backward) {
        c.s = sa;
        c.H0 = backward ? h->d_H0t : h->d_H0f;
        c.Hc = backward ? h->d_Hct : h->d_Hcf;
        c.eps = h->d_eps; c.shape = h->d_shape; c.dts = h->d_dts; c.rb = h->d_rb; c.stats = h->d_stats;
        c.xch = h->d_xch + (backward ? (size_t)K * 4 * NP : 0);
        c.xcc = h->d_xcc + (backward ? (size_t)K * 32 : 0);
        // XCD-local stores (sc0: they land in the XCD's L2, where the siblings' device-scope polls find them) are the
        // default -- C5 shard sweeps 175 -> 141 ms; XCD-local LOADS (buffer_inv sc0 + sc0 load) never saw the data
        // on gfx950 and are not used
        c.xmode = h->cheby_xmode;
        c.tol = h->series_tol;
        c.L = h->L; c.hc_per_traj = h->p.hc_per_traj; c.NP = NP; c.herm = h->herm ? 1 : 0;
    };
    for (int k0 = 0; k0 < K; k0 += h->cheby_round) {
        const int kn = std::min(h->cheby_round, K - k0);
        ChebyArgs cf{}, cb{};
        if (sf) { fill(cf, *sf, false); cf.k0 = k0; cf.kn = kn; }
        if (sb) { fill(cb, *sb, true); cb.k0 = k0; cb.kn = kn; }
        const int nb = 8 * ((kn + 7) / 8) * S;
        if (sf && sb) hipLaunchKernelGGL(cheby_coop_kernel, dim3(2 * nb), dim3(256), lds, s, cf, cb, S, nb);
        else if (sf) { cb = cf; cb.kn = 0; hipLaunchKernelGGL(cheby_coop_kernel, dim3(nb), dim3(256), lds, s, cf, cb, S, nb); }
        else { cf = cb; hipLaunchKernelGGL(cheby_coop_kernel, dim3(nb), dim3(256), lds, s, cf, cb, S, 0); }   // every block is a backward block
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

int status_from_flags(grape_handle *h, int flags) {
    // first: an evaluation whose propagators were never finished raises every other flag as a consequence
    if (flags & 32) {
        h->err = "the squaring plan of the blocked exponential was too short for this evaluation; the plan has been "
                 "adapted: repeat the call (the host-pointer entry points do so themselves)";
        return GRAPE_ERR_AGAIN;
    }
    if (flags & 1) { h->err = "Pade denominator numerically singular in at least one cell"; return GRAPE_ERR_SINGULAR; }
    if (flags & 64) { h->err = "exponential: a generator H_kn dt is not finite (NaN or overflow)"; return GRAPE_ERR_SINGULAR; }
    if (flags & 2) {
        h->err = "The chi state of at least one trajectory has norm < chi_min_norm (optimize.jl:1021-1025)";
        return GRAPE_ERR_CHI_NORM;
    }
    if (flags & 8) { h->err = "cooperative sweep: a sibling workgroup did not arrive (spin limit reached)"; return GRAPE_ERR_HIP; }
    if (flags & 16) {
        h->err = "matrix-free propagator: the series of exp(-i H dt) did not converge within the order limit";
        return GRAPE_ERR_TAYLOR;
    }
    if (flags & 4) {
        // taylor_grad_check_convergence = false (optimize.jl:631-651): the series is cut at max_order and that is not an
        // error.  The series kernels of N > 32 hold 64 terms: a larger taylor_max_order that was actually needed there is
        // still reported -- the reference would have summed the further terms
        const bool cut_at_the_limit = h->NP < 48 || h->taylor_max_order <= 64;
        // (taylor_grad_step! raises only `if check_convergence && max_order > 1`, optimize.jl:644: a series of ONE term is
        // returned as it is -- round-5 advisor finding)
        const bool one_term = h->p.gradient_method == GRAPE_GRAD_TAYLOR && h->taylor_max_order <= 1;
        if ((h->taylor_check || !cut_at_the_limit) && !one_term) {
            h->err = h->taylor_check ? "taylor_grad_step! did not converge within max_order iterations (optimize.jl:644-648)"
                                     : "taylor_grad_check_convergence = false with taylor_max_order > 64: a series of this evaluation was "
                                       "still unconverged at the 64 terms the series kernels of N > 32 hold";
            return GRAPE_ERR_TAYLOR;
        }
    }
    return GRAPE_OK;
}

// several GPUs behind one handle (defined at the end of this file)
int multi_create(grape_handle **out, const grape_problem *p);
// owns a handle under construction: whichever way grape_create / multi_create is left without handing it over -- error
// return or C++ exception on its way to the barrier -- the handle and everything it holds on the devices are released
struct HandleGuard {
    grape_handle *h;
    explicit HandleGuard(grape_handle *h_) : h(h_) {}
    HandleGuard(const HandleGuard &) = delete;
    HandleGuard &operator=(const HandleGuard &) = delete;
    ~HandleGuard() { if (h) { grape_destroy(h); (void)hipGetLastError(); } }
    grape_handle *release() { grape_handle *r = h; h = nullptr; return r; }
};
// composite handles balance the WHOLE problem once (multi_create) and hand the similarity to their shards: every shard
// then works in the same frame (stopping tolerances and the chi-norm guard see the same numbers on every device, and a
// sharded run selects the kernels a single handle selects).  nullptr: grape_create chooses from its own trajectories.
thread_local const std::vector<double> *tl_forced_bal = nullptr;
int multi_fail(grape_handle *h, grape_handle *c, int rc);
int backward_device_impl(grape_handle *h, const double *d_f, double *d_G, hipStream_t s, const double2 *d_chi);

// J from the (all-)reduced sums [Re f, Im f, sum w|tau|^2, Re sum w tau, sum_k J_b,k]: J_parts[1] + J_parts[3],
// optimize.jl:757-766
double functional_from_sums(const grape_handle *h, const double *sums) {
    const double Kt = (double)h->K_total;
    double J;
    if (h->p.functional == GRAPE_J_T_SM) J = 1.0 - (sums[0] * sums[0] + sums[1] * sums[1]) / (Kt * Kt);
    else if (h->p.functional == GRAPE_J_T_SS) J = 1.0 - sums[2] / Kt;
    else J = 1.0 - sums[3] / Kt;
    if (h->p.Dpen && h->p.lambda_b != 0.0) J += h->p.lambda_b * sums[4];
    return J;
}

}  // namespace

extern "C" {

int grape_abi_version(void) { return GRAPE_HIP_ABI_VERSION; }

const char *grape_last_error(grape_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void grape_destroy(grape_handle *h) {
    if (!h) return;
    if (!h->shards.empty() || h->comm_set || !h->d_red.empty()) {
        // (collectives of this handle may still be in flight on the shard streams: wait before the buffers go)
        for (grape_handle *c : h->shards)
            if (hipSetDevice(c->device) == hipSuccess) (void)hipStreamSynchronize(c->stream);
        for (size_t g = 0; g < h->d_red.size(); ++g)
            if (h->d_red[g]) {
                if (g < h->shard_dev.size()) (void)hipSetDevice(h->shard_dev[g]);
                hipFree(h->d_red[g]);
            }
        comm_set_release((CommSet *)h->comm_set);
        if (h->ar0) hipEventDestroy(h->ar0);
        if (h->ar1) hipEventDestroy(h->ar1);
        for (grape_handle *c : h->shards) grape_destroy(c);
        delete h;
        return;
    }
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (double *b : h->d_lg)
        if (b) hipFree(b);
    for (double *b : h->d_lg2)
        if (b) hipFree(b);
    if (h->lg_stream2) { hipStreamSynchronize(h->lg_stream2); hipStreamDestroy(h->lg_stream2); }
    if (h->lg_ev_fork) hipEventDestroy(h->lg_ev_fork);
    if (h->lg_ev_join) hipEventDestroy(h->lg_ev_join);
    void *bufs[] = {h->d_celldeg, h->d_xcc_sw, h->d_scanF, h->d_scan_fw, h->d_scan_bw, h->d_dte, h->d_normpart, h->d_normpart2, h->d_dinv2, h->d_scell2, h->d_colpart2, h->d_smax2, h->d_xch, h->d_xcc, h->d_batchflag, h->d_chi_in, h->d_n1, h->d_gpark, h->d_morder, h->d_inv_tnorm, h->d_ones, h->d_z, h->d_rb, h->d_cls, h->d_rep, h->d_coop, h->d_Dt, h->d_xi, h->d_wq, h->d_gb, h->d_cellflag, h->d_celllist, h->d_gram, h->d_Sf, h->d_dinv, h->d_scell, h->d_colpart, h->d_wgtab, h->d_prog, h->d_splan, h->d_xinit, h->d_H0p, h->d_Hcp, h->d_vecs, h->d_H0q, h->d_Hcq, h->d_H0q3, h->d_Hcq3, h->d_park2, h->d_park3, h->d_H0f, h->d_Hcf, h->d_H0t, h->d_Hct, h->d_dts, h->d_shape, h->d_weights, h->d_psi0,
                    h->d_target, h->d_eps, h->d_U, h->d_fw, h->d_bw, h->d_tg, h->d_ret, h->d_f,
                    h->d_rho};
    for (void *b : bufs)
        if (b) hipFree(b);
    if (h->d_H0p3 && h->d_H0p3 != h->d_H0q3) hipFree(h->d_H0p3);   // (Hermitian operators: the adjoint arrays ARE the plain ones)
    if (h->d_Hcp3 && h->d_Hcp3 != h->d_Hcq3) hipFree(h->d_Hcp3);
    if (h->h_pin) hipHostFree(h->h_pin);
    for (auto &ring : h->ph)
        for (auto &p : ring) {
            if (p.e0) hipEventDestroy(p.e0);
            if (p.e1) hipEventDestroy(p.e1);
        }
    if (h->graph_exec) hipGraphExecDestroy(h->graph_exec);
    if (h->graph) hipGraphDestroy(h->graph);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int grape_create(grape_handle **out, const grape_problem *p) try {
    if (!out || !p) { g_create_error = "null argument"; return GRAPE_ERR_INVALID; }
    *out = nullptr;
    if (p->abi_version != GRAPE_HIP_ABI_VERSION) { g_create_error = "abi_version mismatch"; return GRAPE_ERR_INVALID; }
    if (p->L <= 0) { g_create_error = "no controls in trajectories (workspace.jl:155-157)"; return GRAPE_ERR_NO_CONTROLS; }
    if (p->N <= 0 || p->K <= 0 || p->N_T <= 0 || !p->tlist || !p->H0 || !p->Hc || !p->psi0) {
        g_create_error = "invalid problem dimensions or null array";
        return GRAPE_ERR_INVALID;
    }
    test_throw_point("early");
    if (p->N > 512 || (p->N > 256 && p->prop_method != GRAPE_PROP_SERIES)) {
        g_create_error = "N > 512 is not supported by this build, and 256 < N <= 512 only with prop_method = GRAPE_PROP_SERIES "
                         "(the matrix-free polynomial propagator the reference recommends beyond small systems, README.md:55; "
                         "materialised propagators: fused kernels N <= 64, blocked path N <= 256)";
        return GRAPE_ERR_INVALID;
    }
    if (p->L > 8) { g_create_error = "L > 8 is not supported by this build"; return GRAPE_ERR_INVALID; }
    if (p->functional < 0 || p->functional > 2) { g_create_error = "unknown functional"; return GRAPE_ERR_INVALID; }
    if (p->prop_method != GRAPE_PROP_EXP && p->prop_method != GRAPE_PROP_SERIES) {
        g_create_error = "unknown prop_method";
        return GRAPE_ERR_INVALID;
    }
    for (int n = 0; n < p->N_T; ++n)
        if (!(p->tlist[n + 1] > p->tlist[n])) { g_create_error = "tlist must be strictly increasing"; return GRAPE_ERR_INVALID; }
    if (p->ndev < 0 || p->ndev > 64) { g_create_error = "ndev out of range (0..64)"; return GRAPE_ERR_INVALID; }
    if (p->ndev > 1) return multi_create(out, p);
    if (p->ndev == 1 && getenv("GRAPE_MULTI_RCCL") && atoi(getenv("GRAPE_MULTI_RCCL")) == 1)
        return multi_create(out, p);   // one shard behind a one-rank communicator: the collective code path on a single GPU

    // ---- trajectories without a target_state (optimize.jl:753): the sweeps run against zero targets, tau is NaN ----
    std::vector<double> zero_target;
    grape_problem pnt = *p;
    const bool no_target = p->target == nullptr;
    if (no_target) {
        zero_target.assign((size_t)2 * p->K * p->N, 0.0);
        pnt.target = zero_target.data();
        p = &pnt;
    }

    // ---- balancing (see grape_handle::bal): the problem the handle is built from is D^-1 H D, D^-1 Psi0, D target, D Dpen D ----
    std::vector<double> bal, b_H0, b_Hc, b_psi0, b_target, b_Dpen;
    grape_problem pbal = *p;
    {
        const int N = p->N, K = p->K, L = p->L, Kc = p->hc_per_traj ? K : 1;
        const size_t nn = (size_t)N * N;
        bal = tl_forced_bal ? *tl_forced_bal : balance_of_problem(p);
        if (!bal.empty()) {
            auto similar = [&](const double *src, size_t count, std::vector<double> &dst, bool congruence) {
                dst.assign(src, src + 2 * count * nn);
                for (size_t m = 0; m < count; ++m)
                    for (int j = 0; j < N; ++j)
                        for (int i = 0; i < N; ++i) {   // element (row i, column j), column-major
                            const double f = congruence ? bal[i] * bal[j] : bal[j] / bal[i];
                            dst[2 * (m * nn + (size_t)j * N + i)] *= f;
                            dst[2 * (m * nn + (size_t)j * N + i) + 1] *= f;
                        }
            };
            similar(p->H0, (size_t)K, b_H0, false);
            similar(p->Hc, (size_t)Kc * L, b_Hc, false);
            b_psi0.assign(p->psi0, p->psi0 + 2 * (size_t)K * N);
            b_target.assign(p->target, p->target + 2 * (size_t)K * N);
            for (int k = 0; k < K; ++k)
                for (int i = 0; i < N; ++i)
                    for (int c = 0; c < 2; ++c) {
                        b_psi0[2 * ((size_t)k * N + i) + c] /= bal[i];
                        b_target[2 * ((size_t)k * N + i) + c] *= bal[i];
                    }
            pbal.H0 = b_H0.data(); pbal.Hc = b_Hc.data(); pbal.psi0 = b_psi0.data(); pbal.target = b_target.data();
            if (p->Dpen) {
                similar(p->Dpen, p->dpen_per_traj ? (size_t)K : 1, b_Dpen, true);
                pbal.Dpen = b_Dpen.data();
            }
            p = &pbal;
        }
    }

    grape_handle *h = new grape_handle();
    HandleGuard guard(h);
    h->p = *p;
    h->bal = bal;
    h->no_target = no_target;
    // (taylor_grad_check_convergence belongs to gradient_method = :taylor, optimize.jl:917-918; :gradgen shares the series
    // kernels and their non-convergence flag, and a series of ITS 200 terms that has not converged is always an error --
    // round-5 advisor finding: the switch used to silence it)
    h->taylor_check = p->gradient_method != GRAPE_GRAD_TAYLOR || p->taylor_no_check == 0;
    h->N = p->N; h->L = p->L; h->K = p->K; h->N_T = p->N_T;
    h->K_total = p->K_total > 0 ? p->K_total : p->K;
    h->NT = (p->N + 15) / 16; h->NP = 16 * h->NT;
    // 33 <= N <= 48 runs on three 16-wide tiles (NP = 48); the matrix-free series kernels are built for 16/32/64 only
    if (h->NP == 48 && (p->prop_method == GRAPE_PROP_SERIES || (getenv("GRAPE_NO_NT3") && atoi(getenv("GRAPE_NO_NT3"))))) {
        h->NT = 4; h->NP = 64;
    }
    if (p->N > 64) { h->large = true; h->NP = p->N <= 128 ? 128 : (p->N <= 256 ? 256 : 512); h->NT = h->NP / 16; }
    h->device = p->device;
    if (p->chi_min_norm > 0) h->chi_min_norm = p->chi_min_norm;
    if (p->taylor_tolerance > 0) h->taylor_tol = p->taylor_tolerance;
    if (p->taylor_max_order > 0) h->taylor_max_order = p->taylor_max_order;
    // the gradient-generator route sums the same series until it has converged to rounding: a term below 1e-16 of the state
    // norm.  (The tail behind it is geometric with ratio rho / m -- 0.07 at the headline shape -- so what is cut is below
    // 1e-17; until round 5 the bound was 1e-17 itself: one more order per cell for a gradient that differs by 4e-16 RELATIVE,
    // the rounding of the sums the terms are added to.  tools/tol_ab.py)
    if (p->gradient_method == GRAPE_GRAD_GRADGEN) {
        h->taylor_max_order = 200; h->taylor_tol = 1e-16;
        if (const char *envt = getenv("GRAPE_GRADGEN_TOL")) h->taylor_tol = atof(envt);   // (tools/tol_ab.py)
    }
    h->series = p->prop_method == GRAPE_PROP_SERIES;
    if (p->prop_tolerance > 0) h->series_tol = p->prop_tolerance;

    auto fail = [&](int code) { g_create_error = h->err; return code; };   // (the guard releases the handle)
#define CCHK(expr)                                                                              \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) { h->err = std::string(#expr) + ": " + hipGetErrorString(_e); return fail(GRAPE_ERR_HIP); } \
    } while (0)

    CCHK(hipSetDevice(h->device));
    if (const char *envcu = getenv("GRAPE_STREAM_CUS")) {
        // diagnostic (tools/cu_curve.sh): the handle's stream on the first n CUs of the mask (the bits are dealt round-robin to
        // the XCDs) -- how the phases scale with the CUs they get, i.e. what running two of them side by side could gain
        const int ncu = std::max(8, std::min(256, atoi(envcu)));
        uint32_t mask[8] = {0};
        for (int i = 0; i < ncu; ++i) mask[i >> 5] |= 1u << (i & 31);
        CCHK(hipExtStreamCreateWithCUMask(&h->stream, 8, mask));
    } else
    CCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && ncu > 0)
            h->num_cus = ncu;
    }
    {   // generator classes: bit-identical (H0_k, control operators of k) -> one set of propagators
        const size_t nb0 = (size_t)p->N * p->N * 16, nbc = p->hc_per_traj ? (size_t)p->L * p->N * p->N * 16 : 0;
        auto hash = [](const unsigned char *b, size_t n, unsigned long long hsh) {
            for (size_t i = 0; i < n; ++i) { hsh ^= b[i]; hsh *= 1099511628211ull; }
            return hsh;
        };
        std::vector<unsigned long long> hv(p->K);
        std::vector<int> rep;
        h->cls.assign(p->K, 0);
        const char *env = getenv("GRAPE_NO_DEDUP");
        for (int k = 0; k < p->K; ++k) {
            const unsigned char *b0 = (const unsigned char *)p->H0 + (size_t)k * nb0;
            const unsigned char *bc = nbc ? (const unsigned char *)p->Hc + (size_t)k * nbc : nullptr;
            hv[k] = hash(b0, nb0, 14695981039346656037ull);
            if (bc) hv[k] = hash(bc, nbc, hv[k]);
            int found = -1;
            if (!(env && atoi(env)))
                for (size_t c = 0; c < rep.size() && found < 0; ++c) {
                    const int r = rep[c];
                    if (hv[r] != hv[k]) continue;
                    if (memcmp(b0, (const unsigned char *)p->H0 + (size_t)r * nb0, nb0)) continue;
                    if (bc && memcmp(bc, (const unsigned char *)p->Hc + (size_t)r * nbc, nbc)) continue;
                    found = (int)c;
                }
            if (found < 0) { found = (int)rep.size(); rep.push_back(k); }
            h->cls[k] = found;
        }
        h->KC = (int)rep.size();
        {   // Hermitian generators?  (exact conjugate symmetry of every H0_k and H_l, as built by the caller)
            bool herm = true;
            auto is_herm = [&](const double *m) {
                for (int i = 0; i < p->N && herm; ++i)
                    for (int j = i; j < p->N; ++j) {
                        const double ar = m[2 * ((size_t)j * p->N + i)], ai = m[2 * ((size_t)j * p->N + i) + 1];
                        const double br = m[2 * ((size_t)i * p->N + j)], bi = m[2 * ((size_t)i * p->N + j) + 1];
                        if (ar != br || ai != -bi) { herm = false; break; }
                    }
            };
            const int nhc = (p->hc_per_traj ? p->K : 1) * p->L;
            for (int q = 0; q < nhc && herm; ++q) is_herm(p->Hc + (size_t)q * 2 * p->N * p->N);
            h->herm_ctrl = herm;   // (the control operators alone: a general drift beside Hermitian controls has its own derivative kernel)
            for (int k = 0; k < p->K && herm; ++k) is_herm(p->H0 + (size_t)k * 2 * p->N * p->N);
            const char *envh = getenv("GRAPE_NO_HERM");
            h->herm = herm && !(envh && atoi(envh));
            // GRAPE_EXPM_T18=0: Hermitian generators through the order-13 Pade kernel as well (parity reference, A/B timing)
            const char *envt = getenv("GRAPE_EXPM_T18");
            // (every generator for N > 32: Chebyshev coefficient set and spectral scaling for Hermitian generators, Taylor set
            // and norm-based scaling otherwise; N <= 32 keeps the Pade kernels: two or more cells per CU, latency-bound)
            h->t18 = !(envt && !atoi(envt));
            const char *envs2 = getenv("GRAPE_EXPM_T18_SMALL");
            h->t18_small = !(envs2 && !atoi(envs2));
            const char *env16 = getenv("GRAPE_EXPM_T16");
            h->t16 = !(env16 && !atoi(env16));
            const char *enva = getenv("GRAPE_EXPM_ASM");
            h->asm16 = !(enva && !atoi(enva));
            const char *envg = getenv("GRAPE_EXPM_ASM18G");
            h->asm18g = !(envg && !atoi(envg));
        }
        {
            const char *envp = getenv("GRAPE_EXPM_PERSIST"), *envl = getenv("GRAPE_EXPM_LDS_PAD"), *envx = getenv("GRAPE_CHEBY_XMODE");
            const char *envk = getenv("GRAPE_TEST_HOOKS");
            h->expm_persist = envp ? atoi(envp) != 0 : true;
            h->expm_lds_pad_kb = envl ? std::max(0, atoi(envl)) : 0;
            h->cheby_xmode = envx ? atoi(envx) & 1 : 1;
            h->test_hooks = envk && atoi(envk) == 1;
            const char *envgr = getenv("GRAPE_GRAPH");
            h->graph_ok = !(envgr && atoi(envgr) == 0) && !h->test_hooks;
            const char *envlg = getenv("GRAPE_LG_ASM"), *envd3 = getenv("GRAPE_DERIV3_ASM");
            h->lg_asm = !(envlg && atoi(envlg) == 0);
            h->deriv3_asm = !(envd3 && atoi(envd3) == 0);
            const char *envs = getenv("GRAPE_DERIV_STREAM");
            h->deriv_stream = envs && atoi(envs) == 1;
            h->deriv_stream_never = envs && atoi(envs) == 0;
        }
        if (h->KC < p->K) {
            if (hipSetDevice(h->device) != hipSuccess || hipMalloc((void **)&h->d_cls, p->K * sizeof(int)) != hipSuccess ||
                hipMalloc((void **)&h->d_rep, h->KC * sizeof(int)) != hipSuccess ||
                hipMemcpy(h->d_cls, h->cls.data(), p->K * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(h->d_rep, rep.data(), h->KC * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
                h->err = "generator class tables: HIP allocation/copy failed";
                return fail(GRAPE_ERR_HIP);
            }
        }
    }

    for (auto &ring : h->ph)
        for (auto &ph : ring) { CCHK(hipEventCreate(&ph.e0)); CCHK(hipEventCreate(&ph.e1)); ph.used = false; }

    if (!h->series) {
        // The reference's memory grows as K N (N_T + 1) (stored states, src/workspace.jl:215); the materialised propagators
        // of this path take KC N_T NP^2 16 bytes on top (8.4 GB at C3, 134 GB at C5).  When they do not fit the device,
        // the evaluation does NOT fail in hipMalloc: the handle switches to the matrix-free propagator (power series /
        // Chebyshev sweeps summed to rounding: the same exp(-i H dt) Psi to 1e-15, no U; without the parked terms if
        // those do not fit either) and says so in grape_get_work[12].  GRAPE_U_BUDGET_GB overrides the free-memory figure.
        size_t free_b = 0, total_b = 0;
        const char *envb = getenv("GRAPE_U_BUDGET_GB");
        if (envb) free_b = (size_t)(atof(envb) * 1073741824.0);
        else if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)-1;
        const double u_bytes = (double)h->KC * h->N_T * (double)h->NP * h->NP * 16.0;
        // everything else the ExpProp path allocates is at most a few GB: stored states, operator copies, chunk scratch
        const double other = 2.0 * h->K * (h->N_T + 1.0) * h->NP * 16.0 + 6.0 * h->K * (double)h->NP * h->NP * 16.0 + (h->large ? 7e9 : 1e8);
        if (u_bytes + other > 0.92 * (double)free_b) {
            h->series = true;
            h->u_fallback = true;
            if (h->NP == 48) { h->NT = 4; h->NP = 64; }   // (the matrix-free kernels are built for 16 / 32 / 64)
            if (p->gradient_method == GRAPE_GRAD_GRADGEN) h->series_tol = std::min(h->series_tol, 1e-17);
        }
    }

    const int N = h->N, NP = h->NP, L = h->L, K = h->K, N_T = h->N_T;
    const size_t pp = (size_t)NP * NP, nn = (size_t)N * N;
    const int Kc = p->hc_per_traj ? K : 1;
    // ---- host-side layout conversion: column-major interleaved -> planar row-major (and transpose) ----
    std::vector<double> f((size_t)K * 2 * pp, 0.0), t((size_t)K * 2 * pp, 0.0);
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < N; ++i) {
                const double re = p->H0[2 * ((size_t)k * nn + (size_t)j * N + i)];
                const double im = p->H0[2 * ((size_t)k * nn + (size_t)j * N + i) + 1];
                f[(size_t)k * 2 * pp + (size_t)i * NP + j] = re;
                f[(size_t)k * 2 * pp + pp + (size_t)i * NP + j] = im;
                t[(size_t)k * 2 * pp + (size_t)j * NP + i] = re;
                t[(size_t)k * 2 * pp + pp + (size_t)j * NP + i] = im;
            }
    CCHK(dmalloc(&h->d_H0f, f.size())); CCHK(dmalloc(&h->d_H0t, t.size()));
    CCHK(hipMemcpy(h->d_H0f, f.data(), f.size() * 8, hipMemcpyHostToDevice));
    CCHK(hipMemcpy(h->d_H0t, t.data(), t.size() * 8, hipMemcpyHostToDevice));
    f.assign((size_t)Kc * L * 2 * pp, 0.0); t.assign((size_t)Kc * L * 2 * pp, 0.0);
    for (int kl = 0; kl < Kc * L; ++kl)
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < N; ++i) {
                const double re = p->Hc[2 * ((size_t)kl * nn + (size_t)j * N + i)];
                const double im = p->Hc[2 * ((size_t)kl * nn + (size_t)j * N + i) + 1];
                f[(size_t)kl * 2 * pp + (size_t)i * NP + j] = re;
                f[(size_t)kl * 2 * pp + pp + (size_t)i * NP + j] = im;
                t[(size_t)kl * 2 * pp + (size_t)j * NP + i] = re;
                t[(size_t)kl * 2 * pp + pp + (size_t)j * NP + i] = im;
            }
    CCHK(dmalloc(&h->d_Hcf, f.size())); CCHK(dmalloc(&h->d_Hct, t.size()));
    CCHK(hipMemcpy(h->d_Hcf, f.data(), f.size() * 8, hipMemcpyHostToDevice));
    CCHK(hipMemcpy(h->d_Hct, t.data(), t.size() * 8, hipMemcpyHostToDevice));

    if (NP >= 48) {
        // fragment-packed H^dagger for the MFMA series kernel: [mat][rt][ks][plane][lane],
        // value = conj(H)[col][row] at row = 16 rt + (lane & 15), col = 4 ks + (lane >> 4)
        const int RT = NP / 16, KS = NP / 4;
        auto pack = [&](const double *src, int nmat, std::vector<double> &dst, bool dagger = true) {
            dst.assign((size_t)nmat * RT * KS * 128, 0.0);
            for (int mtx = 0; mtx < nmat; ++mtx)
                for (int rt = 0; rt < RT; ++rt)
                    for (int ks = 0; ks < KS; ++ks)
                        for (int ln = 0; ln < 64; ++ln) {
                            const int row = 16 * rt + (ln & 15), col = 4 * ks + (ln >> 4);
                            if (row >= N || col >= N) continue;
                            // H^dagger[row][col] = conj(H[col][row]); H column-major: H[i][j] at j*N + i
                            const size_t so = dagger ? 2 * ((size_t)mtx * nn + (size_t)row * N + col)
                                                     : 2 * ((size_t)mtx * nn + (size_t)col * N + row);   // H[row][col]
                            const size_t o = (((size_t)mtx * RT + rt) * KS + ks) * 128 + ln;
                            dst[o] = src[so];
                            dst[o + 64] = dagger ? -src[so + 1] : src[so + 1];
                        }
        };
        std::vector<double> pk;
        pack(p->H0, K, pk);
        CCHK(dmalloc(&h->d_H0p, pk.size()));
        CCHK(hipMemcpy(h->d_H0p, pk.data(), pk.size() * 8, hipMemcpyHostToDevice));
        pack(p->Hc, Kc * L, pk);
        CCHK(dmalloc(&h->d_Hcp, pk.size()));
        CCHK(hipMemcpy(h->d_Hcp, pk.data(), pk.size() * 8, hipMemcpyHostToDevice));
        const long nbatch = (long)K * ((N_T + 15) / 16);
        // one workgroup per CU walks its batches for the fused sizes (measured at C3: 6.75 ms with 256 workgroups, 6.83
        // with 1024, 6.93 with 4096; the parking area of the two-pass kernel shrinks with the grid)
        h->deriv_blocks = (int)std::min<long>(nbatch, h->large ? 1024 : std::max(h->num_cus, 64));
        if (const char *envb = getenv("GRAPE_DERIV_BLOCKS")) h->deriv_blocks = (int)std::min<long>(nbatch, std::max(1, atoi(envb)));
        {   // two-pass series kernel (deriv2_kernel): untransposed fragments and the u_a parking area
            const char *env = getenv("GRAPE_DERIV2");
            // (more than four controls beyond N = 64 and the 512-wide padding exist in the two-pass kernel only)
            h->deriv2 = !(env && atoi(env) == 0) || (h->large && (L > 4 || NP > 256));
            if (h->deriv2) {
                pack(p->H0, K, pk, false);
                CCHK(dmalloc(&h->d_H0q, pk.size()));
                CCHK(hipMemcpy(h->d_H0q, pk.data(), pk.size() * 8, hipMemcpyHostToDevice));
                pack(p->Hc, Kc * L, pk, false);
                CCHK(dmalloc(&h->d_Hcq, pk.size()));
                CCHK(hipMemcpy(h->d_Hcq, pk.data(), pk.size() * 8, hipMemcpyHostToDevice));
                h->deriv2_maxm = 64;
                // blocked path: the assembly derivative kernel (asm/gen_d4.py; GRAPE_DERIV4=0: deriv2_kernel, its twin) walks the
                // batches with one workgroup per CU and parks maxm + 1 terms per workgroup
                const char *env4 = getenv("GRAPE_DERIV4");
                if (h->large && !h->series && (NP == 128 || NP == 256) && !(env4 && atoi(env4) == 0)) {
                    h->deriv4_blocks = (int)std::min<long>(nbatch, h->num_cus);
                    if (const char *envb4 = getenv("GRAPE_DERIV4_BLOCKS")) h->deriv4_blocks = (int)std::min<long>(nbatch, std::max(1, atoi(envb4)));
                    const int RT3 = NP / 16, KS3 = NP / 4;
                    auto pack3 = [&](const double *src, int nmat, std::vector<double> &dst, bool dagger) {
                        // [mat][rt][ks][64 lanes x (re, im) | 64 lanes x (re + im)]: element (row 16 rt + (lane & 15), column
                        // 4 ks + (lane >> 4)) of the matrix, or of its conjugate transpose (H column-major: H[i][j] at j N + i)
                        dst.assign((size_t)nmat * RT3 * KS3 * 192, 0.0);
                        for (int mtx = 0; mtx < nmat; ++mtx)
                            for (int rt = 0; rt < RT3; ++rt)
                                for (int ks = 0; ks < KS3; ++ks)
                                    for (int ln = 0; ln < 64; ++ln) {
                                        const int row = 16 * rt + (ln & 15), col = 4 * ks + (ln >> 4);
                                        if (row >= N || col >= N) continue;
                                        const size_t so = dagger ? 2 * ((size_t)mtx * nn + (size_t)row * N + col)
                                                                 : 2 * ((size_t)mtx * nn + (size_t)col * N + row);
                                        const double re = src[so], im = dagger ? -src[so + 1] : src[so + 1];
                                        const size_t o = (((size_t)mtx * RT3 + rt) * KS3 + ks) * 192 + ln;
                                        dst[o + ln] = re; dst[o + ln + 1] = im; dst[o + 128] = re + im;   // (o = base + ln: pairs at 2 ln, sums at 128 + ln)
                                    }
                    };
                    std::vector<double> pk3;
                    pack3(p->H0, K, pk3, false);
                    CCHK(dmalloc(&h->d_H0q3, pk3.size()));
                    CCHK(hipMemcpy(h->d_H0q3, pk3.data(), pk3.size() * 8, hipMemcpyHostToDevice));
                    pack3(p->Hc, Kc * L, pk3, false);
                    CCHK(dmalloc(&h->d_Hcq3, pk3.size()));
                    CCHK(hipMemcpy(h->d_Hcq3, pk3.data(), pk3.size() * 8, hipMemcpyHostToDevice));
                    if (h->herm) {   // H^dagger = H: one set of arrays
                        h->d_H0p3 = h->d_H0q3; h->d_Hcp3 = h->d_Hcq3;
                    } else {
                        pack3(p->H0, K, pk3, true);
                        CCHK(dmalloc(&h->d_H0p3, pk3.size()));
                        CCHK(hipMemcpy(h->d_H0p3, pk3.data(), pk3.size() * 8, hipMemcpyHostToDevice));
                        pack3(p->Hc, Kc * L, pk3, true);
                        CCHK(dmalloc(&h->d_Hcp3, pk3.size()));
                        CCHK(hipMemcpy(h->d_Hcp3, pk3.data(), pk3.size() * 8, hipMemcpyHostToDevice));
                    }
                }
                CCHK(dmalloc(&h->d_park2, std::max((size_t)h->deriv_blocks * h->deriv2_maxm, (size_t)h->deriv4_blocks * (h->deriv2_maxm + 1)) * 2 * NP * 16));
                const char *env3 = getenv("GRAPE_DERIV3");
                // round 6: FEW batches (few trajectories): one wave per batch leaves the chip idle for a whole batch latency
                // (0.41 / 0.65 ms at NP = 48 / 64) while a workgroup per batch (deriv2_kernel) takes 0.17 / 0.21 ms per round of
                // #CUs batches -- measured crossovers ~650 / ~800 batches (K = 4, 1000 steps: 0.41 -> 0.18, 0.65 -> 0.23 ms)
                const long nbatch = (long)K * ((N_T + 15) / 16);
                const double t_wave = (NP == 48 ? 0.41 : 0.65) * std::ceil((double)nbatch / (4.0 * h->num_cus));
                const double t_wg = (NP == 48 ? 0.17 : 0.21) * std::ceil((double)nbatch / (double)h->num_cus);
                const bool few = !(env3 && atoi(env3) != 0) && NP <= 64 && t_wg < t_wave;
                const char *envnh = getenv("GRAPE_NO_HERM");
                h->deriv3_h0g = !h->herm && h->herm_ctrl && !(envnh && atoi(envnh));
                // four tiles per side and more than two controls: the operators do not fit the LDS; the assembly kernel
                // streams them through it (asm/gen_d3s.py; GRAPE_DERIV3S=0: deriv2_kernel's STREAM_L form, the twin)
                const char *env3s = getenv("GRAPE_DERIV3S");
                const bool streamed = h->herm && h->NT == 4 && L > 2 && L <= 8 && !(env3s && atoi(env3s) == 0);
                // four tiles per side, general drift and / or general control operators (up to seven controls): all tiles of
                // every operator stream through the LDS and pass 2 applies the adjoint (asm/gen_d3s.py GenD3G;
                // GRAPE_DERIV3G=0: deriv3_kernel's H0G form where it applies, else deriv2_kernel)
                const char *env3g = getenv("GRAPE_DERIV3G");
                h->deriv3_general = !h->herm && h->NT == 4 && L <= 7 && !(env3g && atoi(env3g) == 0);
                if ((h->herm || h->deriv3_h0g || h->deriv3_general) && !h->large && !h->series
                    && (deriv3_fits(h->NT, L, h->deriv3_h0g) || streamed || h->deriv3_general) && !(env3 && atoi(env3) == 0) && !few) {
                    // workgroups per trajectory: as many as it takes to put a workgroup on every CU, at most one per four batches
                    const int bpk = (N_T + 15) / 16;
                    h->deriv3_wpt = (int)std::max<long>(1, std::min<long>((bpk + 3) / 4, h->num_cus / std::max(1, K)));
                    h->deriv3_blocks = (int)std::min<long>(h->num_cus, (long)K * h->deriv3_wpt);
                    CCHK(dmalloc(&h->d_park3, (size_t)h->deriv3_blocks * 4 * (h->deriv2_maxm + 1) * 2 * NP * 16));   // (+ 1: the assembly kernel parks every order it forms)
                }
            }
        }
        CCHK(dmalloc(&h->d_vecs, (size_t)h->deriv_blocks * 2 * (1 + 8) * 2 * NP * 16));
    }
    if (NP < 48 && h->herm && deriv3_fits(h->NT, L)) {   // one wave per batch also at one and two tiles per side (grape_deriv3.hip.h); the matrix-free
                                          // mode as well: at these sizes the derivative kernel never used the parked forward terms
        const char *env3 = getenv("GRAPE_DERIV3");
        // (few batches at two tiles per side: 0.20 ms per batch latency against 0.14 ms per round of #CUs batches, see above)
        const long nbatch = (long)K * ((N_T + 15) / 16);
        // (one tile per side: deriv_kernel<16> 0.010 ms per ~100 batches against 0.033 ms of batch latency -- the README problem, 32
        // batches: derivatives 0.033 -> 0.010 ms, evaluation 0.102 -> 0.081 ms; C2, 1024 batches: 0.065 against 0.106)
        const bool few = !(env3 && atoi(env3) != 0) &&
                         (NP == 32 ? 0.14 * std::ceil((double)nbatch / (double)h->num_cus) < 0.20 * std::ceil((double)nbatch / (4.0 * h->num_cus))
                                   : NP == 16 && nbatch <= 256);
        if (!(env3 && atoi(env3) == 0) && !few) {
            const int bpk = (N_T + 15) / 16;
            h->deriv2_maxm = 64;
            h->deriv3_wpt = (int)std::max<long>(1, std::min<long>((bpk + 3) / 4, h->num_cus / std::max(1, K)));
            h->deriv3_blocks = (int)std::min<long>(h->num_cus, (long)K * h->deriv3_wpt);
            CCHK(dmalloc(&h->d_park3, (size_t)h->deriv3_blocks * 4 * (h->deriv2_maxm + 1) * 2 * NP * 16));   // (+ 1: the assembly kernel parks every order it forms)
        }
    }
    if (h->large && !h->series) {
        const long ncell = (long)h->KC * N_T;
        // chunk scratch: 6 GB (a launch of 635 cells at N = 256 has tails of ~1 % of its length).  GRAPE_LG_LANES=2: two lanes of
        // chunks of 1 GB on two streams, each covering the other's launch tails -- phase A of the C5 shard 143.4 -> 138.6 ms before
        // the combinations moved into the epilogue of the products, 128.8 -> 127.5 ms since; off by default (1 % for a second
        // scratch set, and kernel statistics in which the launches of the two lanes overlap)
        const char *envl = getenv("GRAPE_LG_LANES"), *envsp0 = getenv("GRAPE_LG_SPEC");
        const bool want_lanes = h->t18 && !(envsp0 && atoi(envsp0) == 0) && envl && atoi(envl) >= 2;
        double scratch_bytes = want_lanes ? 1.0e9 : 6.0e9;
        if (const char *envg = getenv("GRAPE_LG_SCRATCH_GB")) scratch_bytes = std::max(0.1, atof(envg)) * 1e9;   // (experiments: launch tails against scratch)
        const long cap = std::max<long>(1, (long)(scratch_bytes / (9.0 * 2.0 * pp * 8.0)));
        h->chunk = (int)std::min<long>(ncell, std::min<long>(cap, 16384));
        if (const char *envc = getenv("GRAPE_LG_CHUNK")) h->chunk = (int)std::max<long>(1, std::min<long>(h->chunk, atol(envc)));   // (experiments: working set against the Infinity Cache)
        for (auto &b : h->d_lg) CCHK(dmalloc(&b, (size_t)h->chunk * 2 * pp));
        CCHK(dmalloc(&h->d_dinv, (size_t)h->chunk * 2 * 4096));
        CCHK(dmalloc(&h->d_scell, (size_t)h->chunk + 1));
        {
            const char *envsp = getenv("GRAPE_LG_SPEC"), *envsn = getenv("GRAPE_LG_SN");
            h->lg_spec = h->t18 && !(envsp && atoi(envsp) == 0);
            if (const char *envf = getenv("GRAPE_LG_FUSE")) h->lg_fuse = atoi(envf) != 0;
            if (const char *envp = getenv("GRAPE_LG_POW")) h->lg_pow = atoi(envp) != 0;
            if (h->lg_spec) CCHK(dmalloc(&h->d_colpart, (size_t)h->chunk * 2 * LG_PARTS * NP));
            // summed controls of every time step for the generator formation (polynomial route, shared control operators):
            // N_T 2 NP^2 doubles -- 2.1 GB at C5 -- when that is a small part of what the propagators take anyway
            {
                const char *envf2 = getenv("GRAPE_LG_FORM2");
                h->lg_form2 = h->t18 && !p->hc_per_traj && L <= 4 && NP % LG_FORM_ROWS == 0 && !(envf2 && atoi(envf2) == 0);
                if (h->lg_form2) CCHK(dmalloc(&h->d_normpart, (size_t)h->chunk * (NP / LG_FORM_ROWS) * NP));
            }
            const double sn_bytes = (double)N_T * 2.0 * (double)pp * 8.0;
            if (!h->lg_form2 && h->t18 && !p->hc_per_traj && !(envsn && atoi(envsn) == 0) && sn_bytes <= 0.25 * (double)h->KC * N_T * (double)pp * 16.0 + 1e9)
                CCHK(dmalloc(&h->d_Sf, (size_t)N_T * 2 * pp));
            // second lane: only when there is more than one chunk to overlap and the second scratch set is small against the
            // device (the propagators themselves take K N_T pp 16 bytes)
            size_t free_b = 0, total_b = 0;
            CCHK(hipMemGetInfo(&free_b, &total_b));
            const double lane_bytes = 9.0 * (double)h->chunk * 2.0 * (double)pp * 8.0;
            if (want_lanes && h->lg_spec && ncell > h->chunk &&
                lane_bytes + (double)h->KC * N_T * (double)pp * 16.0 + 4e9 < (double)free_b) {
                h->lg_lanes = 2;
                for (auto &b : h->d_lg2) CCHK(dmalloc(&b, (size_t)h->chunk * 2 * pp));
                CCHK(dmalloc(&h->d_dinv2, (size_t)h->chunk * 2 * 4096));
                CCHK(dmalloc(&h->d_scell2, (size_t)h->chunk + 1));
                CCHK(dmalloc(&h->d_colpart2, (size_t)h->chunk * 2 * LG_PARTS * NP));
                CCHK(dmalloc(&h->d_smax2, 1));
                if (h->lg_form2) CCHK(dmalloc(&h->d_normpart2, (size_t)h->chunk * (NP / LG_FORM_ROWS) * NP));
                if (!h->lg_stream2) {
                    CCHK(hipStreamCreateWithFlags(&h->lg_stream2, hipStreamNonBlocking));
                    CCHK(hipEventCreateWithFlags(&h->lg_ev_fork, hipEventDisableTiming));
                    CCHK(hipEventCreateWithFlags(&h->lg_ev_join, hipEventDisableTiming));
                }
            }
        }
    }

    std::vector<double> dts(N_T);
    for (int n = 0; n < N_T; ++n) dts[n] = p->tlist[n + 1] - p->tlist[n];
    CCHK(dmalloc(&h->d_dts, (size_t)N_T));
    CCHK(hipMemcpy(h->d_dts, dts.data(), (size_t)N_T * 8, hipMemcpyHostToDevice));
    if (p->shape) {
        CCHK(dmalloc(&h->d_shape, (size_t)L * N_T));
        CCHK(hipMemcpy(h->d_shape, p->shape, (size_t)L * N_T * 8, hipMemcpyHostToDevice));
    }
    if (p->weights) {
        CCHK(dmalloc(&h->d_weights, (size_t)K));
        CCHK(hipMemcpy(h->d_weights, p->weights, (size_t)K * 8, hipMemcpyHostToDevice));
    }
    CCHK(dmalloc(&h->d_psi0, (size_t)K * N)); CCHK(dmalloc(&h->d_target, (size_t)K * N));
    CCHK(hipMemcpy(h->d_psi0, p->psi0, (size_t)K * N * 16, hipMemcpyHostToDevice));
    CCHK(hipMemcpy(h->d_target, p->target, (size_t)K * N * 16, hipMemcpyHostToDevice));
    {   // concurrent sweeps (see SweepArgs::unit_chi)
        const char *env = getenv("GRAPE_FUSED_SWEEPS");
        const bool have_gb = p->Dpen && p->lambda_b != 0.0;
        h->fuse = (!h->large || h->series) && !have_gb && !(env && atoi(env) == 0);
        std::vector<double> itn(K), ones(K, 1.0);
        for (int k = 0; k < K; ++k) {
            double n2 = 0.0;
            for (int i = 0; i < 2 * N; ++i) n2 += p->target[(size_t)k * 2 * N + i] * p->target[(size_t)k * 2 * N + i];
            itn[k] = n2 > 0.0 ? 1.0 / std::sqrt(n2) : 0.0;
            if (!(n2 > 0.0)) h->fuse = false;   // a zero target has no direction: the sequential path reports the chi norm
        }
        CCHK(dmalloc(&h->d_inv_tnorm, (size_t)K)); CCHK(dmalloc(&h->d_ones, (size_t)K)); CCHK(dmalloc(&h->d_z, (size_t)K));
        CCHK(hipMemcpy(h->d_inv_tnorm, itn.data(), (size_t)K * 8, hipMemcpyHostToDevice));
        CCHK(hipMemcpy(h->d_ones, ones.data(), (size_t)K * 8, hipMemcpyHostToDevice));
    }

    // ---- per-evaluation buffers ----
    CCHK(dmalloc(&h->d_eps, (size_t)L * N_T));
    if (!h->series) CCHK(dmalloc(&h->d_U, (size_t)h->KC * N_T * pp));
    if (!h->series && !h->large) {
        // 1-norms (max column sum; the input is column-major) for the order-13 certificate of expm_single
        const char *env = getenv("GRAPE_NORM_BOUND");
        if (!(env && atoi(env) == 0)) {
            std::vector<double> n1((size_t)K + (size_t)Kc * L);
            auto norm1 = [&](const double *m) {
                double best = 0.0;
                for (int j = 0; j < N; ++j) {
                    double cs = 0.0;
                    for (int i = 0; i < N; ++i) cs += std::hypot(m[2 * ((size_t)j * N + i)], m[2 * ((size_t)j * N + i) + 1]);
                    best = std::max(best, cs);
                }
                return best;
            };
            for (int k = 0; k < K; ++k) n1[k] = norm1(p->H0 + 2 * (size_t)k * nn);
            for (int kl = 0; kl < Kc * L; ++kl) n1[K + kl] = norm1(p->Hc + 2 * (size_t)kl * nn);
            CCHK(dmalloc(&h->d_n1, n1.size()));
            CCHK(hipMemcpy(h->d_n1, n1.data(), n1.size() * 8, hipMemcpyHostToDevice));
        }
    }
    {   // 2-norm estimates of the operators (power iteration): sub-steps of the matrix-free propagator and of the
        // derivative series
        std::vector<double> rb((size_t)K + (size_t)Kc * L);
        // Hermitian generators on the Chebyshev propagator (matrix-free, N > 64): guaranteed bounds instead of estimates
        const bool rigorous = h->herm && h->series && h->large;
        auto bound_of = [&](const double *m) { return rigorous ? herm_norm2_bound(m, N) : norm2_estimate(m, N); };
        const int nops = K + Kc * L;
        auto op_ptr = [&](int q) { return q < K ? p->H0 + 2 * (size_t)q * nn : p->Hc + 2 * (size_t)(q - K) * nn; };
        if (rigorous && nops > 1) {   // two N^3 products per operator: over the host cores
            parallel_for(nops, [&](int q) { rb[q] = bound_of(op_ptr(q)); });
        } else {
            for (int q = 0; q < nops; ++q) rb[q] = bound_of(op_ptr(q));
        }
        CCHK(dmalloc(&h->d_rb, rb.size()));
        CCHK(hipMemcpy(h->d_rb, rb.data(), rb.size() * 8, hipMemcpyHostToDevice));
        // the exact-derivative route sub-steps its series for ||H|| dt > theta; :taylor is the reference's plain recursion
        const char *envd = getenv("GRAPE_DERIV_THETA");
        h->sub_theta = p->gradient_method == GRAPE_GRAD_GRADGEN ? (envd ? atof(envd) : 4.0) : 0.0;
        // (behind the batch flags, round 6: the flags of the economized derivative series, deriv_econ_kernel)
        CCHK(dmalloc(&h->d_batchflag, (size_t)2 * K * ((N_T + 15) / 16)));
        CCHK(hipMemset(h->d_batchflag, 0, (size_t)2 * K * ((N_T + 15) / 16) * sizeof(int)));
    }
    if (h->series) {
        const char *env = getenv("GRAPE_SERIES_THETA");
        if (env && atof(env) > 0) h->series_theta = atof(env);
        // parked terms for the two-pass derivative kernel (N > 32, deriv2 on, series at least as tight as the
        // derivative series, and the area fits a modest share of HBM)
        const char *envp = getenv("GRAPE_SERIES_PARK");
        const size_t bytes = (size_t)K * N_T * 32 * NP * 16;
        size_t free_p = 0, total_p = 0;
        if (const char *envb = getenv("GRAPE_U_BUDGET_GB")) free_p = (size_t)(atof(envb) * 1073741824.0);
        else if (hipMemGetInfo(&free_p, &total_p) != hipSuccess) free_p = (size_t)-1;
        if (NP >= 48 && !h->large && h->deriv2 && h->series_tol <= h->taylor_tol && bytes <= ((size_t)24 << 30) &&
            (double)bytes <= 0.5 * (double)free_p && !(envp && atoi(envp) == 0)) {
            h->maxp = 32;
            CCHK(dmalloc(&h->d_gpark, (size_t)K * N_T * h->maxp * NP));
            CCHK(dmalloc(&h->d_morder, (size_t)K * N_T));
            CCHK(hipMemset(h->d_morder, 0xFF, (size_t)K * N_T * sizeof(int)));
        }
    }
    CCHK(dmalloc(&h->d_fw, (size_t)K * (N_T + 1) * NP));
    CCHK(dmalloc(&h->d_bw, (size_t)K * (N_T + 1) * NP));
    {   // parallel scan of the sweeps over the time axis (N <= 64): a latency chain of N_T steps becomes Bk + NB + Bk steps.  It
        // pays while the chains alone leave the chip idle: phase 1 does NP times the flops of the sweep it replaces, so its
        // throughput term grows with the number of trajectories while the sequential sweeps do not.  Measured crossovers at
        // 1000 steps (forward + backward phase, ms; sequential 1.08 / 1.8 / 1.8 at NP = 32 / 48 / 64): NP = 32: K = 4 / 16 / 64 /
        // 128 -> 0.14 / 0.27 / 0.93 / 1.76; NP = 48: 0.24 / 0.57 / 1.93 / -; NP = 64: 0.32 / 0.92 / 3.44 / -; NP = 16 (500 steps,
        // sequential 0.11): K = 32 / 64 / 128 -> 0.049 / 0.090 / 0.145.  Both sides scale with N_T: the rule is a bound on K.
        const char *envs = getenv("GRAPE_SCAN16");
        if (const char *envs2 = getenv("GRAPE_SCAN")) envs = envs2;
        { const char *env1w = getenv("GRAPE_SWEEP1W"); g_sweep_wg32 = env1w && atoi(env1w) == 0; }
        const bool forced = envs && atoi(envs) != 0;
        const int kmax = NP == 16 ? 96 : NP == 32 ? 64 : NP == 48 ? 48 : 24;
        if (NP <= 64 && !h->series && !h->large && !(envs && atoi(envs) == 0) && N_T >= 8 && (forced || (h->KC <= kmax && N_T >= 64))) {
            // block length: Bk matrix steps (t1 each, a block of phase 1 holds a CU -- NP = 16: a wave -- for its whole chain)
            // + NB + Bk vector steps (t2 each)
            const double t2 = NP == 16 ? 0.22 : NP == 32 ? 1.1 : 1.8, t1 = NP == 16 ? 0.45 : NP == 32 ? 3.3 : NP == 48 ? 7.3 : 13.5;
            const double slots = NP == 16 ? 16.0 * h->num_cus : (double)h->num_cus;
            int best = 4;
            double best_t = 1e300;
            for (int Bk = 4; Bk <= 64 && Bk < std::max(N_T, 5); Bk += (Bk < 16 ? 2 : 4)) {
                const int NBq = (N_T + Bk - 1) / Bk;
                const double t = std::ceil((double)h->KC * NBq / slots) * Bk * t1 + (NBq + Bk) * t2;
                if (t < best_t) { best_t = t; best = Bk; }
            }
            if (const char *envb = getenv("GRAPE_SCAN16_BK")) best = std::max(2, atoi(envb));
            h->scan_Bk = std::min(best, N_T);
            h->scan_NB = (N_T + h->scan_Bk - 1) / h->scan_Bk;
            CCHK(dmalloc(&h->d_scanF, (size_t)h->KC * h->scan_NB * NP * NP));
            CCHK(dmalloc(&h->d_scan_fw, (size_t)K * (h->scan_NB + 1) * NP));
            CCHK(dmalloc(&h->d_scan_bw, (size_t)K * (h->scan_NB + 1) * NP));
            h->scan16 = true;
        }
    }
    CCHK(dmalloc(&h->d_tg, (size_t)K * L * N_T));
    // ONE slab for everything an evaluation hands back or resets: [tau + sums (2K + 8) | G (L N_T) | flags (8 ints) | statistics]
    // -- the single-wait grape_eval reads the first three with one copy and resets the last two with one memset (every copy or
    // memset is a launch of its own: 5 us each on a 340-us evaluation at C2)
    CCHK(dmalloc(&h->d_ret, (size_t)2 * K + 8 + (size_t)L * N_T + 4 + (size_t)GRAPE_STAT_SHARDS * GRAPE_STAT_SLOTS));
    h->d_out = h->d_ret;
    h->d_G = h->d_ret + (size_t)2 * K + 8;
    h->d_flags = (int *)(h->d_G + (size_t)L * N_T);
    h->d_stats = (unsigned long long *)(h->d_G + (size_t)L * N_T + 4);
    CCHK(dmalloc(&h->d_f, 2)); CCHK(dmalloc(&h->d_rho, (size_t)K));
    CCHK(dmalloc(&h->d_cellflag, (size_t)K * N_T));
    // more than two controls shared by all trajectories: the cell fetches H0_k and ONE summed operator S_n (ctrl_sum_kernel)
    // control operators per trajectory (the ensemble of robustness problems): the summed controls are an array per CELL of the
    // generator classes -- as large as the propagators themselves -- when that fits; the assembly cells then apply as they are
    bool sf_per_cell_ok = false;
    const char *envp16 = getenv("GRAPE_EXPM_ASM16P");
    // (Hermitian generators with up to four controls, general ones with up to two: the assembly cell fetches the operators of its
    // trajectory itself)
    const bool p_direct = p->hc_per_traj && (h->herm ? L <= 4 : L <= 2) && !(envp16 && atoi(envp16) == 0);
    if (p->hc_per_traj && !p_direct && h->t18 && !h->large && !h->series && h->NT == 4) {
        size_t free_b = 0, total_b = 0;
        CCHK(hipMemGetInfo(&free_b, &total_b));
        const char *envs = getenv("GRAPE_SF_PER_CELL");
        sf_per_cell_ok = !(envs && atoi(envs) == 0) && 2.0 * (double)h->KC * N_T * NP * NP * 16.0 + 8e9 < (double)free_b;
    }
    const bool sf_shape_ok = !p->hc_per_traj || p_direct || sf_per_cell_ok;
    h->asm16 = h->asm16 && h->t16 && h->t18 && h->herm && !h->large && !h->series && h->NT == 4 && sf_shape_ok && (long)K * N_T < (1L << 28);
    h->asm18g = h->asm18g && h->t18 && !h->herm && !h->large && !h->series && h->NT == 4 && sf_shape_ok && (long)K * N_T < (1L << 28);
    h->asm16p = h->asm16 && p_direct;
    h->asm18gp = h->asm18g && p_direct;
    {
        const char *enve = getenv("GRAPE_DERIV_ECON");
        const bool lg_ok = h->large && h->t18 && h->herm && h->lg_spec && !h->series;
        // (the compiled four-product kernel of three and four tiles per side writes the same verdicts: expm_t18_kernel<.., T16>)
        const bool t16c_ok = h->t16 && h->t18 && h->herm && !h->large && !h->series && (h->NT >= 3 || (h->NT == 2 && h->t18_small));
        h->deriv_econ = (h->asm16 || lg_ok || t16c_ok) && p->gradient_method == GRAPE_GRAD_GRADGEN && h->taylor_tol >= 1e-16 && h->d_batchflag &&
                        !(enve && atoi(enve) == 0);
        if (h->deriv_econ) h->d_econ_pairs = grape_econ_pairs();   // (the compiled derivative kernels read the tables through a pointer)
        if (h->deriv_econ && !h->d_econ_pairs) h->deriv_econ = false;
        if (h->deriv_econ && lg_ok) {
            CCHK(dmalloc(&h->d_celldeg, (size_t)h->KC * N_T));
            CCHK(hipMemset(h->d_celldeg, 0, (size_t)h->KC * N_T * sizeof(int)));
        }
    }
    if (h->asm16p || h->asm18gp) CCHK(dmalloc(&h->d_dte, (size_t)(L <= 2 ? 4 : 8) * N_T));
    if (h->t18 && !h->large && !h->series && ((L > 2 && !p->hc_per_traj) || (h->asm16 && !h->asm16p) || (h->asm18g && !h->asm18gp)) && (h->NT >= 3 || h->t18_small))
        CCHK(dmalloc(&h->d_Sf, (size_t)(p->hc_per_traj ? (size_t)h->KC * N_T : (size_t)N_T) * 2 * NP * NP));
    if (h->asm16 || h->asm18g) {
        // one workgroup per CU (512 registers, 139 KB of LDS), never more workgroups than cells
        const long ncell = (long)h->KC * N_T;
        h->asm_blocks = (int)std::max<long>(1, std::min<long>(h->num_cus, ncell));
        std::vector<int> tab((size_t)4 * h->asm_blocks);
        grape_t16_walks(h->KC, N_T, h->asm_blocks, tab.data());
        CCHK(dmalloc(&h->d_wgtab, tab.size()));
        CCHK(dmalloc(&h->d_splan, (size_t)ncell));
        CCHK(hipMemset(h->d_splan, 0, (size_t)ncell * sizeof(int)));
        CCHK(hipMemcpy(h->d_wgtab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
        const char *envw = getenv("GRAPE_EXPM_WALK"), *envq = getenv("GRAPE_EXPM_SQ");
        h->asm_sq = !(envq && atoi(envq) == 0);
        // (generator classes: one propagator serves several trajectories -- nothing to carry along)
        h->asm_walk = h->KC == K ? (envw ? atoi(envw) & 3 : 3) : 0;
        if (h->scan16) h->asm_walk = 0;   // (few trajectories: the sweeps are scanned, there is nothing for a walk to carry)
        if (h->asm_walk) {
            std::vector<double> xi((size_t)2 * K * 64 * 2, 0.0);
            for (int k = 0; k < K; ++k) {
                double n2 = 0.0;
                for (int i = 0; i < 2 * N; ++i) n2 += p->target[(size_t)k * 2 * N + i] * p->target[(size_t)k * 2 * N + i];
                const double itn = n2 > 0.0 ? 1.0 / std::sqrt(n2) : 0.0;
                for (int i = 0; i < N; ++i) {
                    xi[2 * ((size_t)k * 64 + i)] = p->psi0[2 * ((size_t)k * N + i)];
                    xi[2 * ((size_t)k * 64 + i) + 1] = p->psi0[2 * ((size_t)k * N + i) + 1];
                    xi[2 * ((size_t)(K + k) * 64 + i)] = p->target[2 * ((size_t)k * N + i)] * itn;
                    xi[2 * ((size_t)(K + k) * 64 + i) + 1] = -p->target[2 * ((size_t)k * N + i) + 1] * itn;   // conj
                }
            }
            CCHK(dmalloc(&h->d_xinit, (size_t)2 * K * 64));
            CCHK(hipMemcpy(h->d_xinit, xi.data(), xi.size() * 8, hipMemcpyHostToDevice));
            CCHK(dmalloc(&h->d_prog, (size_t)2 * K));
            CCHK(hipMemset(h->d_prog, 0, (size_t)2 * K * sizeof(int)));
        }
    }
    // (N <= 16 stays with five products: one tile per side is latency-bound -- the kernel gains nothing from the shorter
    // polynomial and the second launch costs 10 us of an evaluation of 0.4 ms; measured at C2: 0.079 -> 0.098 ms)
    if (h->t16 && h->t18 && h->herm && !h->large && !h->series && (h->NT >= 3 || (h->NT == 2 && h->t18_small))) {
        CCHK(dmalloc(&h->d_celllist, (size_t)K * N_T));
        // Gram matrices of the operators of every generator class (plan of the four-product route, t16_plan_kernel)
        // ... and behind each matrix the SHAPE FACTORS kappa_a of its operators (round-4 advisor finding: the estimate
        // R_est = 2 ||H dt||_F / sqrt(N) assumes a semicircle spectrum and is off by up to sqrt(N) / 2 either way).  What the
        // kernel tests is the Schatten-8 norm m8^(1/8) = (sum lam^8)^(1/8); for a semicircle of radius R it is
        // 3.5^(1/8) R.  kappa_a = ||O_a||_S8 / (3.5^(1/8) * 2 ||O_a||_F / sqrt(N)) is 1 for a semicircle, up to N^(3/8) / 2.34
        // for a rank-one operator and 0.72 for a two-point spectrum; the plan multiplies its estimate by the
        // Frobenius-weighted mean of the factors of the operators present in the cell.  ||O||_S8^8 = ||O^4||_F^2 for a
        // Hermitian O: two N^3 products per operator, once per grape_create.
        const int M = L + 1, KCn = h->KC;
        const int GS = M * M + M;                 // doubles per class: Gram matrix | shape factors
        std::vector<double> gram((size_t)KCn * GS);
        std::vector<int> repk(KCn, -1);
        auto shape_factor = [&](const double *op, double fro2) {   // op: N x N column-major complex, Hermitian
            if (!(fro2 > 0.0)) return 1.0;
            std::vector<double> a2((size_t)2 * N * N), a4((size_t)2 * N * N);
            auto square = [&](const double *a, std::vector<double> &c) {
                std::fill(c.begin(), c.end(), 0.0);
                for (int j = 0; j < N; ++j)
                    for (int k = 0; k < N; ++k) {
                        const double br = a[2 * ((size_t)j * N + k)], bi = a[2 * ((size_t)j * N + k) + 1];
                        const double *ak = a + 2 * (size_t)k * N;
                        double *cj = &c[2 * (size_t)j * N];
                        for (int i = 0; i < N; ++i) {
                            cj[2 * i] += ak[2 * i] * br - ak[2 * i + 1] * bi;
                            cj[2 * i + 1] += ak[2 * i] * bi + ak[2 * i + 1] * br;
                        }
                    }
            };
            square(op, a2);
            square(a2.data(), a4);
            double m8 = 0.0;
            for (double x : a4) m8 += x * x;
            const double s8 = std::pow(m8, 0.125), semi = std::pow(3.5, 0.125) * 2.0 * std::sqrt(fro2 / (double)N);
            const double kap = s8 / semi;
            return std::isfinite(kap) && kap > 0.0 ? kap : 1.0;
        };
        std::vector<double> kap_shared((size_t)L, -1.0);   // shared control operators: once, not once per class
        for (int k = K - 1; k >= 0; --k) repk[h->cls.empty() ? k : h->cls[k]] = k;   // first trajectory of the class
        const size_t nn2 = (size_t)2 * N * N;
        for (int kc = 0; kc < KCn; ++kc) {
            const int k = repk[kc] < 0 ? kc : repk[kc];
            auto op = [&](int a) { return a == 0 ? p->H0 + (size_t)k * nn2 : p->Hc + ((size_t)(p->hc_per_traj ? k : 0) * L + (a - 1)) * nn2; };
            for (int a = 0; a < M; ++a)
                for (int b = a; b < M; ++b) {
                    const double *x = op(a), *y = op(b);
                    double sum = 0.;
                    for (size_t i = 0; i < nn2; ++i) sum += x[i] * y[i];
                    gram[(size_t)kc * GS + (size_t)a * M + b] = gram[(size_t)kc * GS + (size_t)b * M + a] = sum;
                }
        }
        if (!p->hc_per_traj)
            for (int l = 0; l < L; ++l) kap_shared[l] = shape_factor(p->Hc + (size_t)l * nn2, gram[(size_t)(1 + l) * M + (1 + l)]);
        parallel_for(KCn, [&](int kc) {   // (two N^3 products per operator)
            const int k = repk[kc] < 0 ? kc : repk[kc];
            for (int a = 0; a < M; ++a) {
                const double fro2 = gram[(size_t)kc * GS + (size_t)a * M + a];
                const double *o = a == 0 ? p->H0 + (size_t)k * nn2 : p->Hc + ((size_t)(p->hc_per_traj ? k : 0) * L + (a - 1)) * nn2;
                gram[(size_t)kc * GS + (size_t)M * M + a] = (a > 0 && !p->hc_per_traj) ? kap_shared[a - 1] : shape_factor(o, fro2);
            }
        });
        CCHK(dmalloc(&h->d_gram, gram.size()));
        CCHK(hipMemcpy(h->d_gram, gram.data(), gram.size() * 8, hipMemcpyHostToDevice));
    }
    if (h->large && h->series) {
        // cooperative polynomial sweeps (grape_cheby.hip.h): S = NP / 16 siblings per trajectory on one XCD, at most one
        // workgroup per CU; the trajectories go through the kernel in rounds of `cheby_round`
        const int per_xcd = std::max(1, h->num_cus / 8), S = h->NP / 16;
        if (per_xcd < S) {
            h->err = "prop_method = GRAPE_PROP_SERIES with N > 64 needs NP / 16 co-resident workgroups per XCD";
            return fail(GRAPE_ERR_INVALID);
        }
        h->cheby_S = S;
        h->cheby_pair = h->fuse && per_xcd >= 2 * S;
        if (!h->cheby_pair) h->fuse = false;   // the two directions do not fit side by side: sequential sweeps
        h->cheby_round = 8 * std::max(1, per_xcd / ((h->cheby_pair ? 2 : 1) * S));
        CCHK(dmalloc(&h->d_xch, (size_t)2 * K * 4 * NP));
        CCHK(dmalloc(&h->d_xcc, (size_t)2 * K * 32));
    }
    if (h->large && !h->series) {
        // cooperative sweeps when the trajectories alone cannot fill the chip: S siblings per trajectory,
        // all siblings of a trajectory on one XCD, at most one workgroup per CU (see sweep_coop_kernel).
        // A workgroup of NW waves owns R = NP / S = NW * RPW state rows, R in {4, 8, 16, 32, 64}.
        hipDeviceProp_t prop;
        CCHK(hipGetDeviceProperties(&prop, h->device));
        const int per_xcd = std::max(1, prop.multiProcessorCount / 8), groups = (K + 7) / 8;
        int S = 1;
        while (S * 2 * groups <= per_xcd && h->NP / (S * 2) >= 4) S *= 2;
        const int R = h->NP / S;
        const char *env = getenv("GRAPE_SWEEP_COOP");
        if (S >= 2 && R <= 64 && !(env && atoi(env) == 0)) {
            h->coop_S = S;
            if (const char *envx = getenv("GRAPE_COOP_XMODE")) h->coop_xmode = atoi(envx) != 0;
            h->coop_nw = R >= 16 ? 16 : R;
            h->coop_rpw = R / h->coop_nw;
            // the two directions have different optima (round 5, C5 shard, forward / backward ms by siblings: 32: 8.5 / 9.6,
            // 16: 7.3 / 11.1, 8: 10.2 / 11.0, 4: 15.3 / -): a step is a latency chain whose length grows with the siblings that
            // have to meet; the forward slice is whole rows (a wave sum per row), the backward one columns (tools/coop_s.sh)
            // round 6 (armed storage rows instead of step counters, stores into the XCD's L2): forward 32 / 16 / 8 siblings
            // 3.5 / 4.3 / 6.9 ms, backward 32 / 16 / 8: 5.4 / 5.7 / 7.1 ms (round 5: 7.1 + 9.6 ms)
            int Sf = h->coop_xmode ? S : (S == 32 ? 16 : S), Sb = S;
            if (const char *e_ = getenv("GRAPE_COOP_S_FW")) Sf = atoi(e_);
            if (const char *e_ = getenv("GRAPE_COOP_S_BW")) Sb = atoi(e_);
            auto valid = [&](int s_) { return s_ >= 2 && s_ <= S && (s_ & (s_ - 1)) == 0 && h->NP / s_ <= 64; };
            if (!valid(Sf)) Sf = S;
            if (!valid(Sb)) Sb = S;
            h->coop_S = Sb; h->coop_nw = h->NP / Sb >= 16 ? 16 : h->NP / Sb; h->coop_rpw = (h->NP / Sb) / h->coop_nw;
            h->coop_S_fw = Sf; h->coop_nw_fw = h->NP / Sf >= 16 ? 16 : h->NP / Sf; h->coop_rpw_fw = (h->NP / Sf) / h->coop_nw_fw;
            CCHK(dmalloc(&h->d_coop, (size_t)2 * K));
            CCHK(dmalloc(&h->d_xcc_sw, (size_t)2 * K * 32));
        }
    }
    CCHK(hipMemset(h->d_flags, 0, 8 * sizeof(int)));
    CCHK(hipMemset(h->d_stats, 0, (size_t)GRAPE_STAT_SHARDS * GRAPE_STAT_SLOTS * sizeof(unsigned long long)));
    CCHK(hipMemset(h->d_fw, 0, (size_t)K * (N_T + 1) * NP * 16));
    CCHK(hipMemset(h->d_bw, 0, (size_t)K * (N_T + 1) * NP * 16));
    CCHK(hipMemset(h->d_tg, 0, (size_t)K * L * N_T * 16));
    h->h_pin_doubles = (size_t)2 * L * N_T + 2 * K + 16;   // pulses | forward outputs (2K + 8) | gradient (single-wait grape_eval)
    if (p->Dpen && p->lambda_b != 0.0) {
        h->have_gb = true;
        const int Kd = p->dpen_per_traj ? K : 1;
        std::vector<double> dt_((size_t)Kd * pp * 2, 0.0);   // Dt[j][i] = D[i][j]; D column-major: D[i][j] at j*N + i
        for (int kd = 0; kd < Kd; ++kd)
            for (int j = 0; j < N; ++j)
                for (int i = 0; i < N; ++i) {
                    dt_[2 * ((size_t)kd * pp + (size_t)j * NP + i)] = p->Dpen[2 * ((size_t)kd * nn + (size_t)j * N + i)];
                    dt_[2 * ((size_t)kd * pp + (size_t)j * NP + i) + 1] = p->Dpen[2 * ((size_t)kd * nn + (size_t)j * N + i) + 1];
                }
        CCHK(dmalloc(&h->d_Dt, (size_t)Kd * pp));
        CCHK(hipMemcpy(h->d_Dt, dt_.data(), dt_.size() * 8, hipMemcpyHostToDevice));
        std::vector<double> wq(N_T + 1);
        for (int m = 0; m <= N_T; ++m)   // trapezoid weights of optimize.jl:727-750
            wq[m] = m == 0 ? (p->tlist[1] - p->tlist[0]) / 2.0
                           : (m < N_T ? 0.5 * (p->tlist[m + 1] - p->tlist[m - 1]) : (p->tlist[N_T] - p->tlist[N_T - 1]) / 2.0);
        CCHK(dmalloc(&h->d_wq, (size_t)N_T + 1));
        CCHK(hipMemcpy(h->d_wq, wq.data(), wq.size() * 8, hipMemcpyHostToDevice));
        CCHK(dmalloc(&h->d_xi, (size_t)K * (N_T + 1) * NP));
        CCHK(dmalloc(&h->d_gb, (size_t)K * (N_T + 1)));
    }
    CCHK(hipHostMalloc((void **)&h->h_pin, h->h_pin_doubles * 8, hipHostMallocDefault));
    // Everything above went through the NULL stream (hipMemset of device memory returns before the fill has run; a copy from
    // pageable memory returns once the data is staged), the evaluations run on the handle's own NON-BLOCKING stream, which
    // the NULL stream does not order: without this wait the fill of the stored-state arrays (1 GB at C4) can still be in
    // flight when the first evaluation starts and zero rows it has already written.  (Found in round 5: the walks of the
    // assembly kernel write fw / bw within microseconds of the first launch -- before, the first writer came 12 ms in.)
    CCHK(hipDeviceSynchronize());
#undef CCHK
    test_throw_point("late");
    *out = guard.release();
    return GRAPE_OK;
}
GRAPE_BARRIER(&g_create_error)

int grape_forward_device(grape_handle *h, const double *d_pulsevals, double *d_out, void *stream_) try {
    if (!h || !d_pulsevals || !d_out) return GRAPE_ERR_INVALID;
    if (!h->shards.empty()) { h->err = "device-pointer entry points need a single-device handle (ndev <= 1)"; return GRAPE_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream_;
    h->foreign_stream = s != h->stream;
    HIPCHK(h, hipSetDevice(h->device));
    // HIP's last-error is sticky per host thread: a failure elsewhere in the process (another handle's failed
    // allocation, a bad device ordinal) must not be reported by the launch checks of this evaluation
    (void)hipGetLastError();
    HIPCHK(h, hipMemsetAsync(h->d_flags, 0, 8 * sizeof(int) + (size_t)GRAPE_STAT_SHARDS * GRAPE_STAT_SLOTS * sizeof(unsigned long long), s));   // flags | statistics
    if (d_pulsevals != h->d_eps)
        HIPCHK(h, hipMemcpyAsync(h->d_eps, d_pulsevals, (size_t)h->L * h->N_T * 8, hipMemcpyDeviceToDevice, s));
    // ---- phase 0: expm of every cell ----
    ExpmArgs ea{};
    ea.H0f = h->d_H0f; ea.Hcf = h->d_Hcf; ea.eps = h->d_eps; ea.shape = h->d_shape; ea.dts = h->d_dts;
    ea.U = h->d_U; ea.flags = h->d_flags; ea.stats = h->d_stats; ea.cellflag = h->d_cellflag;
    ea.K = h->KC; ea.rep = h->d_rep; ea.L = h->L; ea.N_T = h->N_T; ea.hc_per_traj = h->p.hc_per_traj;
    ea.n1 = h->d_n1; ea.n1_k = h->K;
#ifdef GRAPE_DIAG
    ea.ablate = getenv("GRAPE_DIAG_ABLATE") ? atoi(getenv("GRAPE_DIAG_ABLATE")) : 0;
    static unsigned long long *d_stamps = nullptr;
    if (getenv("GRAPE_DIAG_STAMPS")) {
        if (!d_stamps) hipMalloc((void **)&d_stamps, (size_t)h->K * h->N_T * 32 * 8);
        hipMemsetAsync(d_stamps, 0, (size_t)h->K * h->N_T * 32 * 8, s);
    }
    ea.stamps = d_stamps;
    hipMemcpyToSymbolAsync(HIP_SYMBOL(g_diag_slot_base), &d_stamps, sizeof(d_stamps), 0, hipMemcpyHostToDevice, s);
    grape_t18_set_stamps(d_stamps, (void *)s);
#endif
    hipError_t e = hipSuccess;
    int walk_fuse = 0;   // bits of the states the exponential kernel of THIS evaluation carried along (see grape_handle::d_prog)
    if (!h->series) {
        phase_begin(h, 0, s);
        if (h->large) {
            e = h->t18 ? expm_large_t18(h, s) : expm_large(h, s);
        } else {
            const long ncell = (long)ea.K * ea.N_T;
            // workgroups per CU of the polynomial kernel: one at three and four tiles per side (512 registers, 137 / 84 KB of
            // LDS), several of the small ones (NT = 1: 139 registers and 16 KB, NT = 2: 256 registers) -- those are latency-bound
            const int per_cu = h->NT == 1 ? 10 : h->NT == 2 ? 4 : 1;
            const int t18_blocks = 8 * (int)std::max<long>(1, std::min<long>((long)(h->num_cus / 8) * per_cu, (ncell + 7) / 8));
            if (h->t18 && (h->NT >= 3 || h->t18_small)) {
                if (h->d_Sf) {   // S_n = sum_l eps_ln shape_ln H_l for every time step (50 us at C3 with six controls)
                    CtrlSumArgs ca{};
                    ca.Hcf = h->d_Hcf; ca.eps = h->d_eps; ca.shape = h->d_shape; ca.Sf = h->d_Sf;
                    ca.L = h->L; ca.N_T = h->N_T; ca.pp2 = 2 * h->NP * h->NP;
                    ca.per_traj = h->p.hc_per_traj ? 1 : 0; ca.rep = h->d_rep;
                    hipLaunchKernelGGL(ctrl_sum_kernel, dim3((unsigned)((ca.per_traj ? h->KC : 1) * h->N_T), std::max(1, ca.pp2 / 2 / 2048)), dim3(256), 0, s, ca);
                    HIPCHK(h, hipGetLastError());
                    ea.Sf = h->d_Sf;
                }
                const bool t16 = h->d_celllist != nullptr;
                ea.cell_list = h->d_celllist; ea.listed = 0;
                if (t16) {   // the plan of this evaluation: flags[6] = cells predicted beyond the range of the four-product route
                    T16PlanArgs pa{};
                    pa.gram = h->d_gram; pa.eps = h->d_eps; pa.shape = h->d_shape; pa.dts = h->d_dts; pa.flags = h->d_flags;
                    pa.KC = h->KC; pa.L = h->L; pa.N_T = h->N_T; pa.N = h->N;
                    pa.splan = (h->asm16 && h->asm_sq) ? h->d_splan : nullptr;
                    hipLaunchKernelGGL(t16_plan_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, s, pa);
                    HIPCHK(h, hipGetLastError());
                }
                // (three tiles per side, four-product variant: 256 registers and 72 KB -- two workgroups per CU)
                const int blocks16 = (t16 && h->NT == 3) ? 8 * (int)std::max<long>(1, std::min<long>((long)(h->num_cus / 8) * 2, (ncell + 7) / 8)) : t18_blocks;
                h->credit_pending = false;
                if (h->asm18g) {   // general matrices: the assembly cell decides its squarings itself, nothing is handed over
                    walk_fuse = h->asm_walk & (1 | ((h->fuse && h->fuse_on && h->want_bw) ? 2 : 0));
                    if (walk_fuse) HIPCHK(h, hipMemsetAsync(h->d_prog, 0, (size_t)2 * h->K * sizeof(int), s));
                    const void *walk[6] = {h->d_wgtab, h->d_xinit, h->d_fw, h->d_bw, h->d_prog, h->d_splan};
                    if (h->asm18gp) {
                        hipLaunchKernelGGL(dte_kernel, dim3((unsigned)((h->N_T + 255) / 256)), dim3(256), 0, s, (const double *)h->d_eps,
                                           (const double *)h->d_shape, (const double *)h->d_dts, h->L, h->N_T, 2, h->d_dte);
                        HIPCHK(h, hipGetLastError());
                    }
                    e = (hipError_t)grape_t18g_asm_launch(&ea, sizeof(ea), h->d_cellflag, (void *)s, h->asm_blocks, walk, walk_fuse, h->K,
                                                          h->asm18gp ? h->d_dte : nullptr);
                    h->credit_pending = !h->asm18gp;
                }
                else if (t16 && h->asm16p) {   // control operators per trajectory: the cell fetches them itself
                    walk_fuse = h->asm_walk & (1 | ((h->fuse && h->fuse_on && h->want_bw) ? 2 : 0));
                    if (walk_fuse) HIPCHK(h, hipMemsetAsync(h->d_prog, 0, (size_t)2 * h->K * sizeof(int), s));
                    hipLaunchKernelGGL(dte_kernel, dim3((unsigned)((h->N_T + 255) / 256)), dim3(256), 0, s, (const double *)h->d_eps,
                                       (const double *)h->d_shape, (const double *)h->d_dts, h->L, h->N_T, h->L <= 2 ? 2 : 4, h->d_dte);
                    HIPCHK(h, hipGetLastError());
                    const void *walk[6] = {h->d_wgtab, h->d_xinit, h->d_fw, h->d_bw, h->d_prog, h->d_splan};
                    e = (hipError_t)grape_t16p_asm_launch(&ea, sizeof(ea), h->d_cellflag, (void *)s, h->asm_blocks, walk, walk_fuse, h->K, h->d_dte);
                }
                else if (t16 && h->asm16) {   // (the verdicts of the assembly kernel go through the flag array of the Pade path)
                    // what the walks may carry along: Psi always (the forward sweep is the same whatever follows), conj(chi~)
                    // when this evaluation runs the backward sweep from the unit targets (concurrent sweeps)
                    walk_fuse = h->asm_walk & (1 | ((h->fuse && h->fuse_on && h->want_bw) ? 2 : 0));
                    if (walk_fuse) HIPCHK(h, hipMemsetAsync(h->d_prog, 0, (size_t)2 * h->K * sizeof(int), s));
                    const void *walk[6] = {h->d_wgtab, h->d_xinit, h->d_fw, h->d_bw, h->d_prog, h->d_splan};
                    e = (hipError_t)grape_t16_asm_launch(&ea, sizeof(ea), h->d_cellflag, (void *)s, h->asm_blocks, walk, walk_fuse, h->K);
                    h->credit_pending = true;
                }
                else
                    e = (hipError_t)grape_t18_launch(h->NT, h->herm ? 1 : 0, t16 ? 1 : 0, &ea, sizeof(ea), (void *)s, blocks16);
                if (t16 && e == hipSuccess) {   // the cells it listed, by the degree-18 variant (none: the launch ends at once)
                    ea.listed = 1;
                    e = (hipError_t)grape_t18_launch(h->NT, 1, 0, &ea, sizeof(ea), (void *)s, t18_blocks);
                }
            } else
            switch (h->NT) {
                case 1: e = launch_expm<1>(ea, h->herm, s, 0, h->expm_lds_pad_kb); break;
                case 2: e = launch_expm<2>(ea, h->herm, s, 0, h->expm_lds_pad_kb); break;
                case 3: e = launch_expm<3>(ea, h->herm, s, 0, h->expm_lds_pad_kb); break;
                default: {
                    // persistent variant (one workgroup per CU) when every CU gets at least a few cells
                    const int nb = 8 * (h->num_cus / 8);
                    const bool persist = h->expm_persist && nb >= 8 && (long)ea.K * ea.N_T >= 4L * nb;
                    e = launch_expm<4>(ea, h->herm, s, persist ? nb : 0, h->expm_lds_pad_kb);
                    break;
                }
            }
        }
        HIPCHK(h, e);
        if (h->deriv_econ) {   // which derivative batches this evaluation's verdicts certify for the economized series
            DerivEconArgs ec{};
            ec.verdict = h->d_cellflag; ec.splan = h->d_splan; ec.cls = h->d_cls; ec.flags = h->d_flags;
            ec.cell_deg = h->large ? h->d_celldeg : nullptr;
            ec.deg0 = ECON_DEG[0]; ec.deg1 = ECON_DEG[ECON_NSETS - 1];
            static_assert(ECON_NSETS == 4, "segments 1.36 (verdict), 1.6, 2.0 (blocked path), 2.72 (one planned squaring)");
            ec.K = h->K; ec.KC = h->KC; ec.N_T = h->N_T; ec.batches_per_k = (h->N_T + 15) / 16;
            ec.nbatch_total = h->K * ec.batches_per_k; ec.batch_econ = h->d_batchflag + ec.nbatch_total;
            hipLaunchKernelGGL(deriv_econ_kernel, dim3((unsigned)((ec.nbatch_total + 255) / 256)), dim3(256), 0, s, ec);
            HIPCHK(h, hipGetLastError());
        }
        phase_end(h, 0, s);
    }
#ifdef GRAPE_DIAG
    if (ea.stamps) {
        hipStreamSynchronize(s);
        const size_t nb = (size_t)h->KC * h->N_T;
        std::vector<unsigned long long> st(nb * 32);
        hipMemcpy(st.data(), ea.stamps, nb * 32 * 8, hipMemcpyDeviceToHost);
        auto avg = [&](int i1, int i0) {   // over the workgroups that stamped (the persistent kernel has one per CU)
            double sum = 0;
            size_t cnt = 0;
            for (size_t b = 0; b < nb; ++b)
                if (st[b * 32 + i1] && st[b * 32 + i0]) { sum += (double)(st[b * 32 + i1] - st[b * 32 + i0]); ++cnt; }
            return cnt ? sum / cnt : 0.0;
        };
        if (h->t18 && h->NT >= 3) {
            const char *tn[] = {"form A (+norm)", "A2 = A A + half exchange", "A3 = A A2", "store A3, norm A2, barrier", "A6 = A3 A3",
                                "A6 exchanges + norms", "scaling, B1 -> X, barrier", "A9 = B1 B5", "+B4, B3+A9 -> X, B2", "p = B2 + (B3+A9) A9",
                                "squarings", "store U", "end barrier"};
            for (int j = 0; j < 13; ++j) fprintf(stderr, "  stamp %-28s %9.0f cycles\n", tn[j], avg(j + 1, j));
            fprintf(stderr, "  stamp %-28s %9.0f cycles\n", "TOTAL per cell", avg(13, 0));
        } else {
        const char *pn[] = {"form A", "norm", "A2", "A4,A6 (+store A2)", "store A6 + combos", "dual", "U=A*T"};
        for (int j = 0; j < 7; ++j) fprintf(stderr, "  stamp %-24s %9.0f cycles\n", pn[j], avg(11 + j, j ? 10 + j : 0));
        fprintf(stderr, "  stamp %-24s %9.0f cycles\n", "P,Q + first inversion", avg(5, 17));
        fprintf(stderr, "  stamp   barrier %6.0f, P/Q stores %6.0f, barrier %6.0f, strip loads + barrier + I_0 %6.0f\n", avg(26, 17), avg(27, 26), avg(28, 27), avg(5, 28));
        fprintf(stderr, "  stamp   strip loads + barrier %6.0f, then wave 0 (I_0) %6.0f, wave 1 (tiles + prefetch) %6.0f, wave 2 (prefetch) %6.0f\n", avg(22, 28), avg(29, 22), avg(30, 22), avg(31, 22));
        for (int j = 0; j < 4; ++j) fprintf(stderr, "  stamp solve step %d             %9.0f cycles\n", j, avg(6 + j, 5 + j));
        for (int j = 0; j < 1; ++j)
            fprintf(stderr, "  stamp   step %d, waves done after %6.0f %6.0f %6.0f %6.0f cycles\n", j, avg(18 + 4 * j, 5 + j),
                    avg(19 + 4 * j, 5 + j), avg(20 + 4 * j, 5 + j), avg(21 + 4 * j, 5 + j));
        fprintf(stderr, "  stamp %-24s %9.0f cycles\n", "squarings + store U", avg(4, 3));
        fprintf(stderr, "  stamp %-24s %9.0f cycles\n", "TOTAL per cell", avg(4, 0));
        }
    }
#endif
    h->last_walk_fuse = walk_fuse;
    // ---- phase 1: forward sweep + tau ----
    SweepArgs sa{};
    sa.U = h->d_U; sa.cls = h->d_cls; sa.psi0 = h->d_psi0; sa.target = h->d_target; sa.weights = h->d_weights;
    sa.store = h->d_fw; sa.tau = (double2 *)d_out; sa.f = nullptr; sa.rho = h->d_rho; sa.flags = h->d_flags;
    sa.chi_min_norm = h->chi_min_norm;
    sa.K = h->K; sa.K_total = h->K_total; sa.N = h->N; sa.N_T = h->N_T; sa.functional = h->p.functional;
    sa.xmode = h->coop_xmode; sa.xcc = h->d_xcc_sw;
    if (walk_fuse & 1) sa.resume = h->d_prog;
    if (h->test_hooks) {   // fault injection (test suite only, see grape_handle::test_hooks)
        const char *envd = getenv("GRAPE_TEST_DROP_SIBLING");
        sa.drop_sibling = envd ? atoi(envd) : 0;
    }
    phase_begin(h, 1, s);
    const bool pair = h->fuse && h->fuse_on && h->want_bw;
    h->bw_done = h->bw_unit = pair;
    h->z_valid = false;
    if (pair) {
        // backward sweep from the unit targets in the same launch: K more workgroups on the other CUs
        SweepArgs sb = sa;
        sb.store = h->d_bw; sb.tau = (double2 *)h->d_out; sb.f = nullptr; sb.unit_chi = 1; sb.inv_tnorm = h->d_inv_tnorm;
        sb.resume = (walk_fuse & 2) ? h->d_prog + h->K : nullptr;
        if (h->series && h->large) {
            e = launch_cheby(h, &sa, &sb, s);
        } else if (h->series) {
            const SeriesArgs rf = series_args(h, sa, false), rb = series_args(h, sb, true);
            const bool sc = 2 * h->K > h->num_cus;
            e = h->NP == 16 ? launch_series_pair<16>(rf, rb, sc, s) : h->NP == 32 ? launch_series_pair<32>(rf, rb, sc, s)
                                                                                 : launch_series_pair<64>(rf, rb, sc, s);
        } else if (h->scan16) {
            e = launch_scan(h, &sa, &sb, true, s);
        } else {
            e = h->NP == 16 ? launch_sweep_pair<16>(sa, sb, s) : h->NP == 32 ? launch_sweep_pair<32>(sa, sb, s)
                : h->NP == 48 ? launch_sweep_pair<48>(sa, sb, s) : launch_sweep_pair<64>(sa, sb, s);
        }
    } else if (h->series && h->large) {
        e = launch_cheby(h, &sa, nullptr, s);
    } else if (h->series) {
        const SeriesArgs ra = series_args(h, sa, false);
        e = h->NP == 16 ? launch_series<16>(ra, false, s) : h->NP == 32 ? launch_series<32>(ra, false, s)
                                                                       : launch_series<64>(ra, false, s);
    } else if (h->scan16) {
        e = launch_scan(h, &sa, nullptr, true, s);
    } else
    switch (h->NP) {
        case 16: e = launch_sweep<16>(sa, false, s); break;
        case 32: e = launch_sweep<32>(sa, false, s); break;
        case 48: e = launch_sweep<48>(sa, false, s); break;
        case 64: e = launch_sweep<64>(sa, false, s); break;
        default:
            if (h->coop_S)
                e = h->NP == 128 ? launch_coop<2>(sa, false, h->coop_S_fw, h->coop_rpw_fw, h->coop_nw_fw, h->d_coop, s)
                                 : launch_coop<4>(sa, false, h->coop_S_fw, h->coop_rpw_fw, h->coop_nw_fw, h->d_coop, s);
            else {
                hipLaunchKernelGGL((sweep_lg_kernel<false>), dim3(sa.K), dim3(1024), 0, s, sa, h->NP);
                e = hipGetLastError();
            }
    }
    HIPCHK(h, e);
    hipLaunchKernelGGL(tau_reduce_kernel, dim3(1), dim3(64), 0, s, (const double2 *)d_out, h->d_weights, h->K,
                       d_out + 2 * (size_t)h->K);
    HIPCHK(h, hipGetLastError());
    // trajectories without a target_state: tau_k = NaN (optimize.jl:753) and so are the sums built from it (an all-ones
    // bit pattern is a quiet NaN)
    if (h->no_target) HIPCHK(h, hipMemsetAsync(d_out, 0xFF, ((size_t)2 * h->K + 4) * 8, s));
    if (h->have_gb) {   // xi_k(t_n) = -D Psi_k(t_n), g_b values and their trapezoid sum (optimize.jl:727-750)
        GbArgs ga{};
        ga.Dt = h->d_Dt; ga.fw = h->d_fw; ga.xi = h->d_xi; ga.gb = h->d_gb;
        ga.NP = h->NP; ga.N_T = h->N_T; ga.d_per_traj = h->p.dpen_per_traj;
        hipLaunchKernelGGL(gb_kernel, dim3(h->K * (h->N_T + 1)), dim3(256), 0, s, ga);
        hipLaunchKernelGGL(jb_reduce_kernel, dim3(1), dim3(256), 0, s, (const double *)h->d_gb, (const double *)h->d_wq,
                           h->K, h->N_T, d_out + 2 * (size_t)h->K + 4);
        HIPCHK(h, hipGetLastError());
    }
    phase_end(h, 1, s);
    if (d_out != h->d_out)
        HIPCHK(h, hipMemcpyAsync(h->d_out, d_out, ((size_t)2 * h->K + 8) * 8, hipMemcpyDeviceToDevice, s));
    h->have_forward = true;
    if (!h->in_eval) h->n_fwd++;
    return GRAPE_OK;
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

int grape_backward_device(grape_handle *h, const double *d_f, double *d_G, void *stream_) try {
    if (!h || !d_f || !d_G) return GRAPE_ERR_INVALID;
    if (h->no_target) { h->err = "this handle has no target states (grape_problem.target == NULL): the built-in chi does not exist, use grape_backward_chi"; return GRAPE_ERR_INVALID; }
    if (!h->shards.empty()) { h->err = "device-pointer entry points need a single-device handle (ndev <= 1)"; return GRAPE_ERR_INVALID; }
    return backward_device_impl(h, d_f, d_G, (hipStream_t)stream_, nullptr);
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

}  // extern "C"

namespace {
// d_chi != nullptr: the boundary states are the caller's (grape_backward_chi); the sequential backward sweep runs
// from them whatever the forward call did
int backward_device_impl(grape_handle *h, const double *d_f, double *d_G, hipStream_t s, const double2 *d_chi) {
    if (!h->have_forward) { h->err = "grape_backward called before grape_forward"; return GRAPE_ERR_INVALID; }
    HIPCHK(h, hipSetDevice(h->device));
    (void)hipGetLastError();   // see grape_forward_device
    hipError_t e;
    // ---- phase 2: chi boundary + backward sweep ----
    SweepArgs sa{};
    sa.U = h->d_U; sa.cls = h->d_cls; sa.psi0 = h->d_psi0; sa.target = h->d_target; sa.weights = h->d_weights;
    sa.store = h->d_bw; sa.tau = (double2 *)h->d_out; sa.f = d_f; sa.rho = h->d_rho; sa.flags = h->d_flags;
    sa.xi = h->xi_user ? h->xi_user : (h->have_gb ? h->d_xi : nullptr); sa.wq = h->d_wq;
    sa.lambda_b = h->xi_user ? h->lambda_user : h->p.lambda_b;
    sa.chi_min_norm = h->chi_min_norm;
    sa.K = h->K; sa.K_total = h->K_total; sa.N = h->N; sa.N_T = h->N_T; sa.functional = h->p.functional;
    sa.xmode = h->coop_xmode; sa.xcc = h->d_xcc_sw ? h->d_xcc_sw + (size_t)h->K * 32 : nullptr;
    sa.chi_in = d_chi;
    phase_begin(h, 2, s);
    // (a caller-supplied chi or xi breaks the linearity the concurrent sweeps rely on: the backward sweep runs here)
    const bool unit = h->bw_done && !d_chi && !h->xi_user;
    h->bw_done = false;
    if (d_chi || h->xi_user) h->bw_unit = false;   // d_bw is about to hold the true (normalised) backward states
    if (unit) {
        // the backward states are already there (unit targets): only rho_k and the factors z_k are left
        ChiCoeffArgs ca{};
        ca.s = sa; ca.s.inv_tnorm = h->d_inv_tnorm; ca.rho = h->d_rho; ca.z = h->d_z;
        hipLaunchKernelGGL(chi_coeff_kernel, dim3((h->K + 63) / 64), dim3(64), 0, s, ca);
        e = hipGetLastError();
        h->z_valid = true;
    } else if (h->series && h->large) {
        e = launch_cheby(h, nullptr, &sa, s);
    } else if (h->series) {
        const SeriesArgs ra = series_args(h, sa, true);
        e = h->NP == 16 ? launch_series<16>(ra, true, s) : h->NP == 32 ? launch_series<32>(ra, true, s)
                                                                      : launch_series<64>(ra, true, s);
    } else if (h->scan16 && !sa.xi) {
        // (the block propagators of the forward call belong to the same propagators; the running-cost inhomogeneity enters
        // every fine step: the sequential sweep keeps that case)
        e = launch_scan(h, nullptr, &sa, false, s);
    } else
    switch (h->NP) {
        case 16: e = launch_sweep<16>(sa, true, s); break;
        case 32: e = launch_sweep<32>(sa, true, s); break;
        case 48: e = launch_sweep<48>(sa, true, s); break;
        case 64: e = launch_sweep<64>(sa, true, s); break;
        default:
            if (h->coop_S)
                e = h->NP == 128 ? launch_coop<2>(sa, true, h->coop_S, h->coop_rpw, h->coop_nw, h->d_coop + h->K, s)
                                 : launch_coop<4>(sa, true, h->coop_S, h->coop_rpw, h->coop_nw, h->d_coop + h->K, s);
            else {
                hipLaunchKernelGGL((sweep_lg_kernel<true>), dim3(sa.K), dim3(1024), 0, s, sa, h->NP);
                e = hipGetLastError();
            }
    }
    HIPCHK(h, e);
    phase_end(h, 2, s);
    // ---- phase 3: per-cell derivative overlaps ----
    DerivArgs da{};
    da.H0t = h->d_H0t; da.Hct = h->d_Hct; da.eps = h->d_eps; da.shape = h->d_shape; da.dts = h->d_dts;
    da.fw = h->d_fw; da.bw = h->d_bw; da.rho = unit ? h->d_ones : h->d_rho; da.tg = h->d_tg; da.flags = h->d_flags; da.stats = h->d_stats;
    da.K = h->K; da.L = h->L; da.N_T = h->N_T; da.hc_per_traj = h->p.hc_per_traj;
    da.max_order = h->taylor_max_order; da.tol = h->taylor_tol;
    da.rb = h->d_rb; da.rb_k = h->K; da.sub_theta = h->sub_theta;
    // enough blocks to fill 256 CUs a few times over, but long runs per block to amortise the tile loads
    int cpb = 16;
    while (cpb > 1 && (long)h->K * ((h->N_T + cpb - 1) / cpb) < 2048) cpb >>= 1;
    da.cells_per_block = cpb;
    const int nblocks = h->K * ((h->N_T + cpb - 1) / cpb);
    phase_begin(h, 3, s);
    if (h->NP >= 48 && h->sub_theta > 0.0) {   // which batches of 16 cells need a sub-stepped derivative series?
        HIPCHK(h, hipMemsetAsync(h->d_flags + 3, 0, sizeof(int), s));
        DerivFlagArgs fa{};
        fa.rb = h->d_rb; fa.eps = h->d_eps; fa.shape = h->d_shape; fa.dts = h->d_dts;
        fa.rb_k = h->K; fa.K = h->K; fa.L = h->L; fa.N_T = h->N_T; fa.hc_per_traj = h->p.hc_per_traj;
        fa.batches_per_k = (h->N_T + 15) / 16; fa.nbatch_total = h->K * fa.batches_per_k;
        fa.sub_theta = h->sub_theta; fa.batch_flag = h->d_batchflag; fa.flags = h->d_flags;
        hipLaunchKernelGGL(deriv_flag_kernel, dim3((fa.nbatch_total + 255) / 256), dim3(256), 0, s, fa);
        HIPCHK(h, hipGetLastError());
    }
    if (h->NP >= 48 && h->deriv2) {
        Deriv2Args d2{};
        d2.H0p = h->d_H0p; d2.Hcp = h->d_Hcp; d2.H0q = h->d_H0q; d2.Hcq = h->d_Hcq;
        d2.eps = h->d_eps; d2.shape = h->d_shape; d2.dts = h->d_dts;
        d2.fw = h->d_fw; d2.bw = h->d_bw; d2.rho = unit ? h->d_ones : h->d_rho; d2.tg = h->d_tg; d2.park = h->d_park2;
        d2.flags = h->d_flags; d2.stats = h->d_stats;
        d2.K = h->K; d2.L = h->L; d2.N_T = h->N_T; d2.hc_per_traj = h->p.hc_per_traj;
        d2.max_order = h->taylor_max_order; d2.maxm = h->deriv2_maxm; d2.tol = h->taylor_tol;
        d2.batches_per_k = (h->N_T + 15) / 16;
        d2.nbatch_total = h->K * d2.batches_per_k;
        if (h->series) { d2.gpark = h->d_gpark; d2.morder = h->d_morder; d2.maxp = h->maxp; }
        if (h->sub_theta > 0.0) d2.batch_flag = h->d_batchflag;
#ifdef GRAPE_DIAG
        d2.ablate = getenv("GRAPE_DIAG_ABLATE_D2") ? atoi(getenv("GRAPE_DIAG_ABLATE_D2")) : 0;
#endif
        if (h->d_park3 && !d2.gpark) {
            d2.park = h->d_park3;
            if (h->deriv_econ && !h->deriv3_general && !h->deriv3_h0g) {   // (without sub-steps the batch flags stay zero, from grape_create)
                d2.batch_flag = h->d_batchflag;
                d2.batch_econ = 1; d2.econ_pairs = h->d_econ_pairs;
            }
            e = (hipError_t)grape_deriv3_launch(h->NT, &d2, sizeof(d2), h->d_H0f, h->d_Hcf, h->deriv3_wpt, 0, h->deriv3_general ? 2 : (h->deriv3_h0g ? 1 : 0), h->deriv3_asm ? 1 : 0, (void *)s, h->deriv3_blocks);
        } else if (h->deriv4_blocks && !d2.gpark) {
            if (h->deriv_econ && h->d_celldeg) {
                d2.batch_flag = h->d_batchflag;
                d2.batch_econ = 1; d2.econ_pairs = h->d_econ_pairs;
            }
            e = (hipError_t)grape_deriv4_launch(h->NP, &d2, sizeof(d2), h->d_H0q3, h->d_Hcq3, h->d_H0p3, h->d_Hcp3, (void *)s, h->deriv4_blocks);
        } else {
            // (deriv2_kernel: Hermitian operators at 48 and 64 with the verdicts of the four-product kernel; the blocked path's
            // compiled twin keeps the Taylor sum -- its differential tests compare orders)
            if (h->deriv_econ && !d2.gpark && !h->large && h->herm) {
                d2.batch_flag = h->d_batchflag;
                d2.batch_econ = 1; d2.econ_pairs = h->d_econ_pairs;
            }
            e = launch_deriv2(h->NP, d2, h->deriv_blocks, s, h->deriv_stream, h->deriv_stream_never);
        }
    } else if (h->NP >= 48) {
        DerivMfmaArgs dm{};
        dm.H0p = h->d_H0p; dm.Hcp = h->d_Hcp; dm.eps = h->d_eps; dm.shape = h->d_shape; dm.dts = h->d_dts;
        dm.fw = h->d_fw; dm.bw = h->d_bw; dm.rho = unit ? h->d_ones : h->d_rho; dm.tg = h->d_tg; dm.vecs = h->d_vecs;
        dm.flags = h->d_flags; dm.stats = h->d_stats;
        dm.K = h->K; dm.L = h->L; dm.N_T = h->N_T; dm.hc_per_traj = h->p.hc_per_traj;
        dm.max_order = h->taylor_max_order; dm.tol = h->taylor_tol;
        dm.batches_per_k = (h->N_T + 15) / 16;
        dm.nbatch_total = h->K * dm.batches_per_k;
        if (h->sub_theta > 0.0) dm.batch_flag = h->d_batchflag;
        e = launch_deriv_mfma(h->NP, dm, h->deriv_blocks, s);
    } else {
        bool deep = false;
        if (h->d_park3) {
            // one wave per batch (deriv3_kernel); cells whose series needs sub-steps are deriv_kernel's: it is launched behind
            // and ends at once unless deriv_flag_kernel counted such a batch -- in which case deriv3_kernel ends at once
            Deriv2Args d2{};
            d2.eps = h->d_eps; d2.shape = h->d_shape; d2.dts = h->d_dts;
            d2.fw = h->d_fw; d2.bw = h->d_bw; d2.rho = unit ? h->d_ones : h->d_rho; d2.tg = h->d_tg; d2.park = h->d_park3;
            d2.flags = h->d_flags; d2.stats = h->d_stats;
            d2.K = h->K; d2.L = h->L; d2.N_T = h->N_T; d2.hc_per_traj = h->p.hc_per_traj;
            d2.max_order = h->taylor_max_order; d2.maxm = h->deriv2_maxm; d2.tol = h->taylor_tol;
            d2.batches_per_k = (h->N_T + 15) / 16;
            d2.nbatch_total = h->K * d2.batches_per_k;
            const bool sub = h->sub_theta > 0.0;
            d2.deep_redo = (!sub && h->taylor_max_order > h->deriv2_maxm) ? 1 : 0;
            if (h->deriv_econ && h->herm && h->NT == 2) {   // two tiles per side behind the compiled four-product kernel (round 6, 4.3)
                d2.batch_flag = h->d_batchflag;              // (a flagged batch ends this kernel at once: deriv_kernel does every cell)
                d2.batch_econ = 1; d2.econ_pairs = h->d_econ_pairs;
            }
            if (sub) {
                HIPCHK(h, hipMemsetAsync(h->d_flags + 3, 0, sizeof(int), s));
                DerivFlagArgs fa{};
                fa.rb = h->d_rb; fa.eps = h->d_eps; fa.shape = h->d_shape; fa.dts = h->d_dts;
                fa.rb_k = h->K; fa.K = h->K; fa.L = h->L; fa.N_T = h->N_T; fa.hc_per_traj = h->p.hc_per_traj;
                fa.batches_per_k = d2.batches_per_k; fa.nbatch_total = d2.nbatch_total;
                fa.sub_theta = h->sub_theta; fa.batch_flag = h->d_batchflag; fa.flags = h->d_flags;
                hipLaunchKernelGGL(deriv_flag_kernel, dim3((fa.nbatch_total + 255) / 256), dim3(256), 0, s, fa);
                HIPCHK(h, hipGetLastError());
            }
            HIPCHK(h, (hipError_t)grape_deriv3_launch(h->NT, &d2, sizeof(d2), h->d_H0f, h->d_Hcf, h->deriv3_wpt, sub ? 1 : 0, 0, h->deriv3_asm ? 1 : 0, (void *)s,
                                                      h->deriv3_blocks));
            da.only_if = sub ? h->d_flags + 3 : nullptr;
            // gradient_method = :taylor with taylor_grad_max_order beyond the terms deriv3_kernel parks (the reference's default
            // is 100, src/optimize.jl:914): a batch that is not converged by then raises flags[7] and deriv_kernel, which
            // honours any order, redoes the evaluation's derivatives behind it (it leaves at once while flags[7] is zero)
            if (d2.deep_redo) { deep = true; da.only_if = h->d_flags + 7; }
        }
        if (!h->d_park3 || h->sub_theta > 0.0 || deep) {   // (no sub-stepping, :taylor route: deriv3_kernel has done every cell)
            switch (h->NP) {
                case 16: e = launch_deriv<16>(da, nblocks, s); break;
                default: e = launch_deriv<32>(da, nblocks, s); break;
            }
        }
    }
    HIPCHK(h, e);
    if (h->NP >= 48 && h->sub_theta > 0.0) {
        // second pass over the batches the fast kernel flagged (cells whose series needs sub-steps); it ends at once
        // when there are none
        DerivSubArgs ds{};
        DerivMfmaArgs &dm = ds.m;
        dm.H0p = h->d_H0p; dm.Hcp = h->d_Hcp; dm.eps = h->d_eps; dm.shape = h->d_shape; dm.dts = h->d_dts;
        dm.fw = h->d_fw; dm.bw = h->d_bw; dm.rho = unit ? h->d_ones : h->d_rho; dm.tg = h->d_tg; dm.vecs = h->d_vecs;
        dm.flags = h->d_flags; dm.stats = h->d_stats;
        dm.K = h->K; dm.L = h->L; dm.N_T = h->N_T; dm.hc_per_traj = h->p.hc_per_traj;
        dm.max_order = h->taylor_max_order; dm.tol = h->taylor_tol;
        dm.batches_per_k = (h->N_T + 15) / 16;
        dm.nbatch_total = h->K * dm.batches_per_k;
        ds.rb = h->d_rb; ds.rb_k = h->K; ds.sub_theta = h->sub_theta; ds.batch_flag = h->d_batchflag;
        HIPCHK(h, launch_deriv_sub(h->NP, ds, h->deriv_blocks, s));
    }
    phase_end(h, 3, s);
    // ---- phase 4: sum over trajectories ----
    phase_begin(h, 4, s);
    const int LN = h->L * h->N_T;
    hipLaunchKernelGGL(grad_reduce_kernel, dim3((LN + 15) / 16), dim3(256), 0, s, h->d_tg, h->K, LN, d_G,
                       unit ? (const double2 *)h->d_z : (const double2 *)nullptr);
    HIPCHK(h, hipGetLastError());
    phase_end(h, 4, s);
    if (!h->capturing) h->n_bwd++;
    return GRAPE_OK;
}
}  // namespace

namespace {
// flags of the last evaluation in pinned memory: behind the staged pulses, forward outputs and gradient (the layout of the
// result slab, so that the single-wait grape_eval fetches all of it with one copy)
int *pinned_flags(grape_handle *h) {
    return (int *)(h->h_pin + (size_t)2 * h->L * h->N_T + 2 * (size_t)h->K + 8);
}
// what the host learns from the flags of an evaluation that has completed: launch plans for the next one, then the status
int digest_flags(grape_handle *h, const int *flags) {
    // blocked path: the squaring plan follows the counts seen on the device (one spare launch costs microseconds)
    if (h->large && !h->series) h->sq_plan = std::max(2, flags[1] + 1);
    return status_from_flags(h, flags[0]);
}

}  // namespace

extern "C" {

int grape_set_fused_sweeps(grape_handle *h, int on) try {
    if (!h) return GRAPE_ERR_INVALID;
    if (!h->shards.empty()) {
        int r = 1;
        for (grape_handle *c : h->shards) r &= grape_set_fused_sweeps(c, on);
        return r;
    }
    h->fuse_on = on != 0;
    return (h->fuse && h->fuse_on) ? 1 : 0;
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

int grape_check(grape_handle *h, void *stream_) try {
    if (!h) return GRAPE_ERR_INVALID;
    if (!h->shards.empty()) {
        for (grape_handle *c : h->shards) {
            const int rc = grape_check(c, c->stream);
            if (rc) { h->err = c->err; return rc; }
        }
        return GRAPE_OK;
    }
    // the flags travel to pinned memory on the caller's stream and ONE wait serves both (a blocking hipMemcpy behind the
    // stream synchronisation was a second round trip of ~10 us: 3 % of a C2 evaluation)
    int *flags = pinned_flags(h);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(flags, h->d_flags, 8 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream_));
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream_));
    return digest_flags(h, flags);
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

}  // extern "C"

namespace {

// The host-pointer calls are split into an asynchronous half (copies and launches on the handle's stream) and a
// half that waits and reads the pinned staging area, so that a handle with several devices can put all of them to work
// from one host thread before it waits for the first.
int forward_enqueue(grape_handle *h, const double *pulsevals, bool copy_out = true) {
    HIPCHK(h, hipSetDevice(h->device));
    const size_t nl = (size_t)h->L * h->N_T;
    memcpy(h->h_pin, pulsevals, nl * 8);
    HIPCHK(h, hipMemcpyAsync(h->d_eps, h->h_pin, nl * 8, hipMemcpyHostToDevice, h->stream));
    const int rc = grape_forward_device(h, h->d_eps, h->d_out, h->stream);
    if (rc) return rc;
    if (copy_out)
        HIPCHK(h, hipMemcpyAsync(h->h_pin + nl, h->d_out, ((size_t)2 * h->K + 8) * 8, hipMemcpyDeviceToHost, h->stream));
    return GRAPE_OK;
}
int forward_finish(grape_handle *h, double *tau) {
    const int rc = grape_check(h, h->stream);
    if (rc) return rc;
    if (tau) memcpy(tau, h->h_pin + (size_t)h->L * h->N_T, (size_t)2 * h->K * 8);
    return GRAPE_OK;
}
const double *forward_sums(const grape_handle *h) { return h->h_pin + (size_t)h->L * h->N_T + 2 * (size_t)h->K; }

int backward_enqueue(grape_handle *h, const double f_total[2], const double *chi, const double *xi = nullptr,
                     double lambda_b = 0.0, bool copy_out = true) {
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->bal.empty()) {   // chi~ = D chi, xi~ = D xi (the backward recursion runs with U~^dagger = D U^dagger D^-1)
        auto scaled = [&](const double *src, size_t rows, std::vector<double> &dst) {
            dst.assign(src, src + 2 * rows * h->N);
            for (size_t r = 0; r < rows; ++r)
                for (int i = 0; i < h->N; ++i) { dst[2 * (r * h->N + i)] *= h->bal[i]; dst[2 * (r * h->N + i) + 1] *= h->bal[i]; }
            return dst.data();
        };
        if (chi) chi = scaled(chi, (size_t)h->K, h->bal_stage_chi);
        if (xi) xi = scaled(xi, (size_t)h->K * (h->N_T + 1), h->bal_stage_xi);
    }
    if (xi) {   // [K][N_T+1][N] complex -> d_xi [K][N_T+1][NP] (zero padded); trapezoid weights of optimize.jl:727-750
        const size_t rows = (size_t)h->K * (h->N_T + 1);
        if (!h->d_xi) HIPCHK(h, dmalloc(&h->d_xi, rows * h->NP));
        if (!h->d_wq) {
            const int N_T = h->N_T;
            std::vector<double> tl(N_T + 1), wq(N_T + 1);
            HIPCHK(h, hipMemcpy(wq.data(), h->d_dts, (size_t)N_T * 8, hipMemcpyDeviceToHost));   // dt_n
            for (int m = 0; m <= N_T; ++m)
                tl[m] = m == 0 ? 0.5 * wq[0] : (m < N_T ? 0.5 * (wq[m - 1] + wq[m]) : 0.5 * wq[N_T - 1]);
            HIPCHK(h, dmalloc(&h->d_wq, (size_t)N_T + 1));
            HIPCHK(h, hipMemcpy(h->d_wq, tl.data(), tl.size() * 8, hipMemcpyHostToDevice));
        }
        if (h->NP != h->N) HIPCHK(h, hipMemsetAsync(h->d_xi, 0, rows * h->NP * 16, h->stream));
        HIPCHK(h, hipMemcpy2DAsync(h->d_xi, (size_t)h->NP * 16, xi, (size_t)h->N * 16, (size_t)h->N * 16, rows,
                                   hipMemcpyHostToDevice, h->stream));
    }
    const double2 *d_chi = nullptr;
    if (chi) {
        if (!h->d_chi_in) HIPCHK(h, dmalloc(&h->d_chi_in, (size_t)h->K * h->N));
        HIPCHK(h, hipMemcpyAsync(h->d_chi_in, chi, (size_t)h->K * h->N * 16, hipMemcpyHostToDevice, h->stream));
        d_chi = h->d_chi_in;
    }
    const double f0[2] = {0.0, 0.0};
    HIPCHK(h, hipMemcpyAsync(h->d_f, f_total ? f_total : f0, 16, hipMemcpyHostToDevice, h->stream));
    h->xi_user = xi ? h->d_xi : nullptr;
    h->lambda_user = lambda_b;
    const int rc = backward_device_impl(h, h->d_f, h->d_G, h->stream, d_chi);
    h->xi_user = nullptr;
    if (rc) return rc;
    if (copy_out) HIPCHK(h, hipMemcpyAsync(h->h_pin, h->d_G, (size_t)h->L * h->N_T * 8, hipMemcpyDeviceToHost, h->stream));
    return GRAPE_OK;
}
int backward_finish(grape_handle *h, double *G, bool accumulate) {
    const int rc = grape_check(h, h->stream);
    if (rc) return rc;
    if (!G) return GRAPE_OK;
    const size_t nl = (size_t)h->L * h->N_T;
    if (accumulate) for (size_t i = 0; i < nl; ++i) G[i] += h->h_pin[i];
    else memcpy(G, h->h_pin, nl * 8);
    return GRAPE_OK;
}

// ---- several devices behind one handle: every call walks the shards twice (enqueue, then wait) ----
int multi_fail(grape_handle *h, grape_handle *c, int rc) { h->err = c->err; return rc; }

// The enqueue half of a call for every shard (copies and launches on the shard's stream, nothing waits): one host thread
// per shard, so that a shard's several hundred launches (blocked path: ~260 per evaluation) do not delay the start of the
// next device by their host time; the waiting half and both reductions stay on the calling thread, in shard order
// (bitwise repeatable).  Returns the first failing shard's status.
template <class F>
int multi_enqueue(grape_handle *h, F fn) {
    const size_t G = h->shards.size();
    std::vector<int> rcs(G, 0);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<char> started(G, 0);
    bool threaded = h->multi_threads && G > 1;
    if (threaded) {
        std::vector<std::thread> pool;
        pool.reserve(G);
        try {   // (std::thread's constructor may throw std::system_error: nothing may cross the extern "C" boundary)
            for (size_t g = 0; g < G; ++g) {
                // (an exception may not leave a thread function -- std::terminate, whatever the entry point's barrier says:
                // the shard reports it as its own GRAPE_ERR_HOST and the other shards are waited for below)
                pool.emplace_back([&, g]() {
                    try {
                        if (h->test_hooks && g + 1 == G) test_throw_point("shard");
                        rcs[g] = fn(h->shards[g], g);
                    }
                    catch (const std::bad_alloc &) { rcs[g] = barrier_fail(&h->shards[g]->err, "std::bad_alloc (out of host memory) in a shard thread"); }
                    catch (const std::exception &e) { rcs[g] = barrier_fail(&h->shards[g]->err, e.what()); }
                    catch (...) { rcs[g] = barrier_fail(&h->shards[g]->err, "unknown exception in a shard thread"); }
                });
                started[g] = 1;
            }
        } catch (...) {
            threaded = false;   // the shards without a thread are enqueued from this one, below
        }
        for (auto &t : pool) t.join();
    }
    if (!threaded) {
        for (size_t g = 0; g < G; ++g) {
            if (started[g]) continue;
            started[g] = 1;
            rcs[g] = fn(h->shards[g], g);
            if (rcs[g]) break;
        }
    }
    h->host_enqueue_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    h->host_enqueue_calls += 1;
    for (size_t g = 0; g < G; ++g)
        if (rcs[g]) {
            // a shard failed: the launches and asynchronous copies of the OTHER shards are still in flight, some of them
            // from the caller's pageable buffers (xi, chi), which the caller may free after the error return -- wait
            for (size_t q = 0; q < G; ++q)
                if (started[q] && q != g && hipSetDevice(h->shards[q]->device) == hipSuccess) (void)hipStreamSynchronize(h->shards[q]->stream);
            return multi_fail(h, h->shards[g], rcs[g]);
        }
    return GRAPE_OK;
}

// one all-reduce (sum) of `count` doubles per shard, src(c) -> d_red[g] + off, on the shard streams; timed on shard 0
template <typename Src>
int multi_allreduce(grape_handle *h, Src src, size_t off, size_t count, bool timed) {
    const RcclApi &api = rccl_api();
    CommSet *cs = (CommSet *)h->comm_set;
    std::lock_guard<std::mutex> lock(cs->mtx);   // (the set may be shared with other handles on the same devices)
    if (timed) { hipSetDevice(h->shards[0]->device); hipEventRecord(h->ar0, h->shards[0]->stream); }
    ncclResult_t r = api.GroupStart();
    for (size_t g = 0; g < h->shards.size() && r == ncclSuccess; ++g) {
        grape_handle *c = h->shards[g];
        hipSetDevice(c->device);
        r = api.AllReduce(src(c), h->d_red[g] + off, count, ncclDouble, ncclSum, cs->comms[g], c->stream);
    }
    const ncclResult_t re = api.GroupEnd();
    if (r == ncclSuccess) r = re;
    if (r != ncclSuccess) {
        h->err = std::string("RCCL all-reduce failed: ") + (api.GetErrorString ? api.GetErrorString(r) : "?");
        return GRAPE_ERR_HIP;
    }
    if (timed) { hipSetDevice(h->shards[0]->device); hipEventRecord(h->ar1, h->shards[0]->stream); }
    return GRAPE_OK;
}

// First evaluation of a handle that reduces with RCCL: the all-reduced numbers against the host-staged sum of the shards'
// own partial results (shard order), and every rank's copy against rank 0's.  All streams have been waited for.
// part(c): device pointer of shard c's `count` partial doubles; off: offset of the totals in d_red[g]
template <typename Part>
int rccl_validate(grape_handle *h, Part part, size_t off, size_t count, const char *what) {
    std::vector<double> staged(count, 0.0), tmp(count), tot0(count);
    for (size_t g = 0; g < h->shards.size(); ++g) {
        grape_handle *c = h->shards[g];
        HIPCHK(h, hipSetDevice(c->device));
        HIPCHK(h, hipMemcpy(tmp.data(), part(c), count * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < count; ++i) staged[i] += tmp[i];
        HIPCHK(h, hipMemcpy(tmp.data(), h->d_red[g] + off, count * 8, hipMemcpyDeviceToHost));
        if (g == 0) tot0 = tmp;
        else if (memcmp(tmp.data(), tot0.data(), count * 8)) {
            h->err = std::string("RCCL all-reduce of the ") + what + ": rank " + std::to_string(g) + " holds other totals than rank 0 "
                     "(GRAPE_MULTI_RCCL=0 selects the host-staged reductions)";
            return GRAPE_ERR_HIP;
        }
    }
    double scale = 0.0, worst = 0.0;
    bool finite_mismatch = false;
    for (size_t i = 0; i < count; ++i) {
        if (std::isfinite(staged[i]) != std::isfinite(tot0[i])) finite_mismatch = true;
        if (!std::isfinite(staged[i])) continue;
        scale = std::max(scale, std::fabs(staged[i]));
        worst = std::max(worst, std::fabs(staged[i] - tot0[i]));
    }
    // another summation order: a few ulps of the largest partial per shard
    if (finite_mismatch || worst > 1e-12 * scale + 1e-300) {
        char buf[256];
        snprintf(buf, sizeof(buf), "RCCL all-reduce of the %s disagrees with the host-staged sum of the shards (max |difference| %.3e at scale %.3e; "
                                   "GRAPE_MULTI_RCCL=0 selects the host-staged reductions)", what, worst, scale);
        h->err = buf;
        return GRAPE_ERR_HIP;
    }
    return GRAPE_OK;
}

int multi_forward(grape_handle *h, const double *pulsevals, double *tau) {
    {
        const bool want_bw = h->want_bw;
        const int rc = multi_enqueue(h, [&](grape_handle *c, size_t) {
            c->want_bw = want_bw;
            const int r = forward_enqueue(c, pulsevals, !h->use_rccl);
            c->want_bw = true;
            return r;
        });
        if (rc) return rc;
    }
    if (h->use_rccl) {
        // the 8 partial sums of every shard: RCCL all-reduce on the shard streams, then the slab copies (tau of the shard
        // and, in the place of its partial sums, the totals)
        const int rc = multi_allreduce(h, [](grape_handle *c) { return (const void *)(c->d_out + 2 * (size_t)c->K); }, 0, 8, false);
        if (rc) return rc;
        for (size_t g = 0; g < h->shards.size(); ++g) {
            grape_handle *c = h->shards[g];
            const size_t nl = (size_t)c->L * c->N_T;
            HIPCHK(h, hipSetDevice(c->device));
            HIPCHK(h, hipMemcpyAsync(c->h_pin + nl, c->d_out, (size_t)2 * c->K * 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(h, hipMemcpyAsync(c->h_pin + nl + 2 * (size_t)c->K, h->d_red[g], 8 * 8, hipMemcpyDeviceToHost, c->stream));
        }
    }
    double *sums = h->h_multi.data();   // [8]: shard sums added in shard order (fixed: reproducible) / the all-reduced totals
    std::fill(sums, sums + 8, 0.0);
    int again = 0;
    for (size_t g = 0; g < h->shards.size(); ++g) {
        grape_handle *c = h->shards[g];
        const int rc = forward_finish(c, tau ? tau + 2 * (size_t)h->shard_lo[g] : nullptr);
        if (rc == GRAPE_ERR_AGAIN) { again = 1; h->err = c->err; continue; }   // every shard adapts its own plan
        if (rc) return multi_fail(h, c, rc);
        const double *cs = forward_sums(c);
        if (h->use_rccl) { if (g == 0) for (int i = 0; i < 8; ++i) sums[i] = cs[i]; }
        else for (int i = 0; i < 8; ++i) sums[i] += cs[i];
    }
    if (again) return GRAPE_ERR_AGAIN;
    if (h->use_rccl && !h->rccl_fwd_checked) {
        const int rc = rccl_validate(h, [](grape_handle *c) { return (const double *)(c->d_out + 2 * (size_t)c->K); }, 0, 8, "partial sums");
        if (rc) return rc;
        h->rccl_fwd_checked = true;
    }
    h->have_forward = true;
    return GRAPE_OK;
}

int multi_backward(grape_handle *h, const double f_total[2], const double *chi, double *G, const double *xi = nullptr,
                   double lambda_b = 0.0) {
    {
        const int rc = multi_enqueue(h, [&](grape_handle *c, size_t g) {
            return backward_enqueue(c, f_total, chi ? chi + 2 * (size_t)h->shard_lo[g] * h->N : nullptr,
                                    xi ? xi + 2 * (size_t)h->shard_lo[g] * (h->N_T + 1) * h->N : nullptr, lambda_b, !h->use_rccl);
        });
        if (rc) return rc;
    }
    if (h->use_rccl) {
        // sum over k of optimize.jl:579 across the shards: RCCL all-reduce of the L N_T doubles on the shard streams; the
        // total comes back from shard 0 (the summation order is RCCL's, fixed for a given set of devices)
        const size_t nl = (size_t)h->L * h->N_T;
        int rc = multi_allreduce(h, [](grape_handle *c) { return (const void *)c->d_G; }, 8, nl, true);
        if (rc) return rc;
        grape_handle *c0 = h->shards[0];
        HIPCHK(h, hipSetDevice(c0->device));
        HIPCHK(h, hipMemcpyAsync(c0->h_pin, h->d_red[0] + 8, nl * 8, hipMemcpyDeviceToHost, c0->stream));
        for (size_t g = 0; g < h->shards.size(); ++g) {
            rc = backward_finish(h->shards[g], g == 0 ? G : nullptr, false);
            if (rc) return multi_fail(h, h->shards[g], rc);
        }
        float ms = 0.f;
        if (hipSetDevice(c0->device) == hipSuccess && hipEventElapsedTime(&ms, h->ar0, h->ar1) == hipSuccess) {
            h->allreduce_ms += ms;
            h->allreduce_calls += 1;
        }
        if (!h->rccl_bwd_checked) {
            rc = rccl_validate(h, [](grape_handle *c) { return (const double *)c->d_G; }, 8, nl, "gradient");
            if (rc) return rc;
            h->rccl_bwd_checked = true;
        }
        return GRAPE_OK;
    }
    for (size_t g = 0; g < h->shards.size(); ++g) {   // sum over k of optimize.jl:579 across the shards, in shard order
        grape_handle *c = h->shards[g];
        const int rc = backward_finish(c, G, g > 0);
        if (rc) return multi_fail(h, c, rc);
    }
    return GRAPE_OK;
}

}  // namespace

extern "C" {

int grape_forward(grape_handle *h, const double *pulsevals, double *tau) try {
    if (!h || !pulsevals) return GRAPE_ERR_INVALID;
    if (!h->shards.empty()) {
        int rc = multi_forward(h, pulsevals, tau);
        if (rc == GRAPE_ERR_AGAIN) rc = multi_forward(h, pulsevals, tau);   // launch plan adapted: once more
        return rc;
    }
    int rc = forward_enqueue(h, pulsevals);
    if (rc) return rc;
    rc = forward_finish(h, tau);
    if (rc == GRAPE_ERR_AGAIN) {   // the squaring plan was too short and has been adapted by grape_check
        rc = forward_enqueue(h, pulsevals);
        if (rc) return rc;
        rc = forward_finish(h, tau);
    }
    return rc;
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

int grape_backward(grape_handle *h, const double f_total[2], double *G_partial) try {
    if (!h || !f_total || !G_partial) return GRAPE_ERR_INVALID;
    if (!h->have_forward) { h->err = "grape_backward called before grape_forward"; return GRAPE_ERR_INVALID; }
    if (h->no_target) { h->err = "this handle has no target states (grape_problem.target == NULL): the built-in chi does not exist, use grape_backward_chi"; return GRAPE_ERR_INVALID; }
    if (!h->shards.empty()) return multi_backward(h, f_total, nullptr, G_partial);
    int rc = backward_enqueue(h, f_total, nullptr);
    if (rc) return rc;
    return backward_finish(h, G_partial, false);
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

int grape_backward_chi(grape_handle *h, const double *chi, double *G) try {
    if (!h || !chi || !G) return GRAPE_ERR_INVALID;
    if (!h->have_forward) { h->err = "grape_backward_chi called before grape_forward"; return GRAPE_ERR_INVALID; }
    if (!h->shards.empty()) return multi_backward(h, nullptr, chi, G);
    int rc = backward_enqueue(h, nullptr, chi);
    if (rc) return rc;
    return backward_finish(h, G, false);
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

int grape_backward_xi(grape_handle *h, const double f_total[2], const double *chi, const double *xi, double lambda_b,
                      double *G) try {
    if (!h || !xi || !G || (!chi && !f_total)) return GRAPE_ERR_INVALID;
    if (!h->have_forward) { h->err = "grape_backward_xi called before grape_forward"; return GRAPE_ERR_INVALID; }
    if (h->no_target && !chi) { h->err = "this handle has no target states (grape_problem.target == NULL): grape_backward_xi needs the caller's chi"; return GRAPE_ERR_INVALID; }
    if (!h->shards.empty()) return multi_backward(h, chi ? nullptr : f_total, chi, G, xi, lambda_b);
    int rc = backward_enqueue(h, chi ? nullptr : f_total, chi, xi, lambda_b);
    if (rc) return rc;
    return backward_finish(h, G, false);
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

int grape_get_sums(grape_handle *h, double sums[8]) try {
    if (!h || !sums) return GRAPE_ERR_INVALID;
    if (!h->have_forward) { h->err = "grape_get_sums called before grape_forward"; return GRAPE_ERR_INVALID; }
    if (!h->shards.empty()) {
        std::fill(sums, sums + 8, 0.0);
        for (grape_handle *c : h->shards) {
            double cs[8];
            const int rc = grape_get_sums(c, cs);
            if (rc) return multi_fail(h, c, rc);
            for (int i = 0; i < 8; ++i) sums[i] += cs[i];
        }
        return GRAPE_OK;
    }
    HIPCHK(h, hipSetDevice(h->device));
    // (the handle's stream only: other handles of the process keep running -- unless the last device-pointer call ran on a
    // stream of the caller's, which a blocking copy is not ordered against: then the device)
    HIPCHK(h, h->foreign_stream ? hipDeviceSynchronize() : hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(sums, h->d_out + 2 * (size_t)h->K, 8 * sizeof(double), hipMemcpyDeviceToHost));
    return GRAPE_OK;
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

int grape_get_final_states(grape_handle *h, double *psiT) try {
    if (!h || !psiT) return GRAPE_ERR_INVALID;
    if (!h->have_forward) { h->err = "grape_get_final_states called before grape_forward"; return GRAPE_ERR_INVALID; }
    if (!h->shards.empty()) {
        for (size_t g = 0; g < h->shards.size(); ++g) {
            const int rc = grape_get_final_states(h->shards[g], psiT + 2 * (size_t)h->shard_lo[g] * h->N);
            if (rc) return multi_fail(h, h->shards[g], rc);
        }
        return GRAPE_OK;
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, h->foreign_stream ? hipDeviceSynchronize() : hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy2D(psiT, (size_t)h->N * 16, h->d_fw + (size_t)h->N_T * h->NP,
                          (size_t)(h->N_T + 1) * h->NP * 16, (size_t)h->N * 16, h->K, hipMemcpyDeviceToHost));
    if (!h->bal.empty())   // Psi = D Psi~
        for (int k = 0; k < h->K; ++k)
            for (int i = 0; i < h->N; ++i) { psiT[2 * ((size_t)k * h->N + i)] *= h->bal[i]; psiT[2 * ((size_t)k * h->N + i) + 1] *= h->bal[i]; }
    return GRAPE_OK;
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

int grape_eval(grape_handle *h, const double *pulsevals, double *J, double *G, double *tau, double *psiT) try {
    if (!h || !pulsevals || !J) return GRAPE_ERR_INVALID;
    if (h->K != h->K_total) {
        h->err = "grape_eval needs K == K_total; use grape_forward/grape_backward for shards";
        return GRAPE_ERR_INVALID;
    }
    if (h->no_target) {
        h->err = "this handle has no target states (grape_problem.target == NULL, optimize.jl:753): J_T and chi are the caller's -- "
                 "grape_forward + grape_get_final_states + grape_backward_chi";
        return GRAPE_ERR_INVALID;
    }
    const bool multi = !h->shards.empty();
    if (!multi && G && !h->large) {
        // One device, functional and gradient: the whole evaluation is enqueued without a host round trip in the middle --
        // the backward half takes f = sum_k w_k tau_k straight from the device-side sums of the forward half (this handle
        // owns all trajectories), and ONE wait at the end reads J's sums, tau, G and the error flags.  (A small system is
        // launch- and latency-bound: at C2 the wait between the halves was a tenth of the evaluation.)
        const size_t nl_ = (size_t)h->L * h->N_T;
        double *gpin = h->h_pin + nl_ + 2 * (size_t)h->K + 8;
        int rc;
        const long ne = h->n_eval++;
        bool replayed = false;
        if (h->graph_ok && ne >= 2 && (ne & 15) != 0) {   // (see grape_handle::graph)
            if (h->graph_exec && h->graph_fuse_on != h->fuse_on) {
                hipGraphExecDestroy(h->graph_exec); hipGraphDestroy(h->graph);
                h->graph_exec = nullptr; h->graph = nullptr;
            }
            if (!h->graph_exec) {
                HIPCHK(h, hipSetDevice(h->device));
                (void)hipGetLastError();
                bool ok = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
                if (ok) {
                    h->capturing = true; h->in_eval = true; h->want_bw = true;
                    rc = forward_enqueue(h, pulsevals, false);
                    if (!rc) rc = backward_device_impl(h, h->d_out + 2 * (size_t)h->K, h->d_G, h->stream, nullptr);
                    if (!rc && hipMemcpyAsync(h->h_pin + nl_, h->d_ret, ((size_t)2 * h->K + 8 + nl_ + 4) * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = GRAPE_ERR_HIP;
                    h->capturing = false; h->in_eval = false;
                    hipGraph_t gph = nullptr;
                    ok = hipStreamEndCapture(h->stream, &gph) == hipSuccess && gph && !rc;
                    if (ok) ok = hipGraphInstantiate(&h->graph_exec, gph, nullptr, nullptr, 0) == hipSuccess;
                    if (ok) {
                        h->graph = gph; h->graph_fuse_on = h->fuse_on;
                        h->graph_bw_unit = h->bw_unit; h->graph_z_valid = h->z_valid; h->graph_credit = h->credit_pending;
                    } else {
                        if (gph) hipGraphDestroy(gph);
                        h->graph_exec = nullptr;
                    }
                }
                if (!ok) { h->graph_ok = false; (void)hipGetLastError(); }   // this runtime cannot capture the sequence: launches as before
            }
            if (h->graph_exec) {
                memcpy(h->h_pin, pulsevals, nl_ * 8);
                HIPCHK(h, hipSetDevice(h->device));
                HIPCHK(h, hipGraphLaunch(h->graph_exec, h->stream));
                h->foreign_stream = false; h->have_forward = true; h->bw_done = false;
                h->bw_unit = h->graph_bw_unit; h->z_valid = h->graph_z_valid; h->credit_pending = h->graph_credit;
                replayed = true;
            }
        }
        if (!replayed) {
            phase_begin(h, 5, h->stream);
            h->in_eval = true;
            h->want_bw = true;
            rc = forward_enqueue(h, pulsevals, false);
            h->in_eval = false;
            if (rc) { h->n_fwd++; return rc; }
            rc = backward_device_impl(h, h->d_out + 2 * (size_t)h->K, h->d_G, h->stream, nullptr);
            if (rc) { h->n_fwd++; return rc; }
            // forward outputs | G | flags: contiguous in the result slab and in the staging area -- ONE copy
            HIPCHK(h, hipMemcpyAsync(h->h_pin + nl_, h->d_ret, ((size_t)2 * h->K + 8 + nl_ + 4) * 8, hipMemcpyDeviceToHost, h->stream));
            phase_end(h, 5, h->stream);
            h->n_fwd++;
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        rc = digest_flags(h, pinned_flags(h));
        if (rc) return rc;
        if (tau) memcpy(tau, h->h_pin + (size_t)h->L * h->N_T, (size_t)2 * h->K * 8);
        *J = functional_from_sums(h, forward_sums(h));
        memcpy(G, gpin, (size_t)h->L * h->N_T * 8);
        if (psiT) return grape_get_final_states(h, psiT);
        return GRAPE_OK;
    }
    if (!multi) phase_begin(h, 5, h->stream);
    h->in_eval = true;  // keep the forward slot open until the whole evaluation has been recorded
    h->want_bw = G != nullptr;   // functional only: no backward sweep alongside the forward one
    int rc = grape_forward(h, pulsevals, tau);
    h->want_bw = true;
    h->in_eval = false;
    if (rc) { h->n_fwd++; return rc; }
    const double *sums = multi ? h->h_multi.data() : forward_sums(h);  // f_re, f_im, sum w|tau|^2, Re sum w tau, sum J_b
    *J = functional_from_sums(h, sums);   // J_parts[1] (+ J_parts[3] = lambda_b * sum(J_b_trajectory), optimize.jl:764-766)
    const double f[2] = {sums[0], sums[1]};
    if (psiT) {
        rc = grape_get_final_states(h, psiT);
        if (rc) { h->n_fwd++; return rc; }
    }
    if (G) {
        rc = grape_backward(h, f, G);
        if (rc) { h->n_fwd++; return rc; }
    }
    if (!multi) phase_end(h, 5, h->stream);
    h->n_fwd++;
    return GRAPE_OK;
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

int grape_get_tau_grads(grape_handle *h, double *out) try {
    if (!h || !out) return GRAPE_ERR_INVALID;
    if (!h->shards.empty()) {   // [k][l][n]: the shards are contiguous blocks of k
        for (size_t g = 0; g < h->shards.size(); ++g) {
            const int rc = grape_get_tau_grads(h->shards[g], out + 2 * (size_t)h->shard_lo[g] * h->L * h->N_T);
            if (rc) return multi_fail(h, h->shards[g], rc);
        }
        return GRAPE_OK;
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out, h->d_tg, (size_t)h->K * h->L * h->N_T * 16, hipMemcpyDeviceToHost));
    return GRAPE_OK;
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

int grape_get_storage(grape_handle *h, int which, double *out) try {
    if (!h || !out || which < 0 || which > 1) return GRAPE_ERR_INVALID;
    if (!h->shards.empty()) {
        for (size_t g = 0; g < h->shards.size(); ++g) {
            const int rc = grape_get_storage(h->shards[g], which, out + 2 * (size_t)h->shard_lo[g] * (h->N_T + 1) * h->N);
            if (rc) return multi_fail(h, h->shards[g], rc);
        }
        return GRAPE_OK;
    }
    if (which == 1 && h->bw_unit && !h->z_valid) {
        h->err = "backward states requested between grape_forward and grape_backward (concurrent sweeps: the boundary "
                 "coefficient of chi is applied by grape_backward)";
        return GRAPE_ERR_INVALID;
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const double2 *src = which == 0 ? h->d_fw : h->d_bw;
    HIPCHK(h, hipMemcpy2D(out, (size_t)h->N * 16, src, (size_t)h->NP * 16, (size_t)h->N * 16,
                          (size_t)h->K * (h->N_T + 1), hipMemcpyDeviceToHost));
    if (which == 1 && h->bw_unit) {
        // concurrent sweeps stored chi~_k(t_n); chi_k(t_n) = (c_k / |c_k|) chi~_k(t_n) with z_k = conj(c_k) ||target_k||
        std::vector<double> z(2 * (size_t)h->K);
        HIPCHK(h, hipMemcpy(z.data(), h->d_z, z.size() * 8, hipMemcpyDeviceToHost));
        for (int k = 0; k < h->K; ++k) {
            const double a = std::hypot(z[2 * k], z[2 * k + 1]);
            const double pr = a > 0 ? z[2 * k] / a : 1.0, pi = a > 0 ? -z[2 * k + 1] / a : 0.0;
            double *o = out + (size_t)k * (h->N_T + 1) * h->N * 2;
            for (size_t j = 0; j < (size_t)(h->N_T + 1) * h->N; ++j) {
                const double xr = o[2 * j], xi = o[2 * j + 1];
                o[2 * j] = pr * xr - pi * xi;
                o[2 * j + 1] = pr * xi + pi * xr;
            }
        }
    }
    if (!h->bal.empty()) {
        // forward states Psi = D Psi~; backward states chi = D^-1 chi~, normalised like the reference's (||chi_k(T)|| = 1 in
        // the caller's frame: the stored ones are normalised in the balanced frame)
        const size_t per_k = (size_t)(h->N_T + 1) * h->N;
        for (int k = 0; k < h->K; ++k) {
            double *o = out + 2 * (size_t)k * per_k;
            for (size_t j = 0; j < per_k; ++j) {
                const double f = which == 0 ? h->bal[j % h->N] : 1.0 / h->bal[j % h->N];
                o[2 * j] *= f; o[2 * j + 1] *= f;
            }
            if (which == 1) {
                double nrm = 0.;
                const double *last = o + 2 * (size_t)h->N_T * h->N;
                for (int i = 0; i < 2 * h->N; ++i) nrm += last[i] * last[i];
                nrm = std::sqrt(nrm);
                if (nrm > 0.) for (size_t j = 0; j < 2 * per_k; ++j) o[j] /= nrm;
            }
        }
    }
    return GRAPE_OK;
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

int grape_get_propagator(grape_handle *h, int k, int n, double *out) try {
    // U_kn as N x N column-major complex (debug / parity of the expm kernel)
    if (!h || !out || k < 0 || k >= h->K || n < 0 || n >= h->N_T) return GRAPE_ERR_INVALID;
    if (!h->shards.empty()) {
        size_t g = 0;
        while (g + 1 < h->shards.size() && k >= h->shard_lo[g + 1]) ++g;
        const int rc = grape_get_propagator(h->shards[g], k - h->shard_lo[g], n, out);
        return rc ? multi_fail(h, h->shards[g], rc) : GRAPE_OK;
    }
    if (h->series) {
        h->err = h->u_fallback ? "the propagators did not fit the device: this handle evaluates matrix-free (grape_get_work[12]), no propagator is materialised"
                               : "prop_method = GRAPE_PROP_SERIES is matrix-free: no propagator is materialised";
        return GRAPE_ERR_INVALID;
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const size_t pp = (size_t)h->NP * h->NP;
    std::vector<double> tmp(2 * pp);
    HIPCHK(h, hipMemcpy(tmp.data(), h->d_U + ((size_t)h->cls[k] * h->N_T + n) * pp, pp * 16, hipMemcpyDeviceToHost));
    for (int j = 0; j < h->N; ++j)
        for (int i = 0; i < h->N; ++i) {
            const double f = h->bal.empty() ? 1.0 : h->bal[i] / h->bal[j];   // U = D U~ D^-1 (exact: powers of two)
            out[2 * ((size_t)j * h->N + i)] = f * tmp[2 * ((size_t)i * h->NP + j)];
            out[2 * ((size_t)j * h->N + i) + 1] = f * tmp[2 * ((size_t)i * h->NP + j) + 1];
        }
    return GRAPE_OK;
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

int grape_get_timings(grape_handle *h, double *ms, int n) try {
    if (!h || !ms) return GRAPE_ERR_INVALID;
    if (!h->shards.empty()) {   // the devices work side by side: a phase takes as long as its slowest shard
        int cnt = 0;
        for (size_t g = 0; g < h->shards.size(); ++g) {
            double cm[kPhases];
            cnt = grape_get_timings(h->shards[g], cm, n < kPhases ? n : kPhases);
            if (cnt < 0) return multi_fail(h, h->shards[g], cnt);
            for (int i = 0; i < cnt; ++i) ms[i] = g == 0 ? cm[i] : std::max(ms[i], cm[i]);
        }
        // [6]: host wall time of the enqueue halves (forward + backward) per evaluation -- what the calling thread spends
        // before the first device can be waited for
        if (n > kPhases) { ms[kPhases] = h->host_enqueue_calls ? 2.0 * h->host_enqueue_ms / h->host_enqueue_calls : -1.0; cnt = kPhases + 1; }
        // [7]: RCCL all-reduce of the gradient across the shards, HIP events on shard 0's stream (includes waiting for the
        // slowest shard); -1: the reductions are host-staged
        if (n > kPhases + 1) { ms[kPhases + 1] = (h->use_rccl && h->allreduce_calls) ? h->allreduce_ms / h->allreduce_calls : -1.0; cnt = kPhases + 2; }
        return cnt;
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    int cnt = 0;
    for (int i = 0; i < kPhases && i < n; ++i, ++cnt) {
        double sum = 0.0;
        int used = 0;
        for (int r = 0; r < kRing; ++r) {
            float t = 0.f;
            Phase &p = h->ph[r][i];
            if (p.used && hipEventElapsedTime(&t, p.e0, p.e1) == hipSuccess) { sum += t; ++used; }
        }
        ms[i] = used ? sum / used : -1.0;
    }
    return cnt;
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

int grape_reset_timings(grape_handle *h) try {
    if (!h) return GRAPE_ERR_INVALID;
    if (!h->shards.empty()) {
        for (grape_handle *c : h->shards) {
            const int rc = grape_reset_timings(c);
            if (rc) return multi_fail(h, c, rc);
        }
        h->host_enqueue_ms = 0.0;
        h->allreduce_ms = 0.0; h->allreduce_calls = 0;
        h->host_enqueue_calls = 0;
        return GRAPE_OK;
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    for (auto &ring : h->ph)
        for (auto &p : ring) p.used = false;
    h->n_fwd = h->n_bwd = 0;
    h->n_eval = 0;   // (the next two evaluations run uncaptured: fresh phase timings behind every reset, see grape_handle::graph)
    return GRAPE_OK;
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

int grape_get_work(grape_handle *h, double *out, int n) try {
    if (!h || !out || n < 4) return GRAPE_ERR_INVALID;
    if (!h->shards.empty()) {   // every entry is a count: the shards add up
        const int m = n < 19 ? n : 19;
        std::fill(out, out + m, 0.0);
        for (grape_handle *c : h->shards) {
            double cw[19] = {0.};
            const int rc = grape_get_work(c, cw, m);
            if (rc < 0) return multi_fail(h, c, rc);
            for (int i = 0; i < m; ++i) out[i] = (i == 15 || i == 16 || i == 18) ? cw[i] : out[i] + cw[i];   // ([15], [16]: kernel ids, the same in every shard)
        }
        if (m > 14 && out[14] > 0.0) out[14] = h->shards[0]->asm16p ? 3.0 : h->shards[0]->asm18gp ? 4.0 : h->shards[0]->asm18g ? 2.0 : 1.0;   // (an id, not a count)
        return 4;
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    if (h->credit_pending) {   // credited work of the assembly route's last evaluation: pulses, S_n and flags are still in place
        ExpmArgs ea{};
        ea.H0f = h->d_H0f; ea.Hcf = h->d_Hcf; ea.eps = h->d_eps; ea.shape = h->d_shape; ea.dts = h->d_dts;
        ea.flags = h->d_flags; ea.stats = h->d_stats; ea.Sf = h->d_Sf;
        ea.K = h->KC; ea.rep = h->d_rep; ea.L = h->L; ea.N_T = h->N_T; ea.hc_per_traj = h->p.hc_per_traj;
        ea.n1 = h->d_n1; ea.n1_k = h->K;
        HIPCHK(h, (hipError_t)grape_t16_credit_launch(&ea, sizeof(ea), (void *)h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->credit_pending = false;
    }
    unsigned long long sh[GRAPE_STAT_SHARDS * GRAPE_STAT_SLOTS], st[GRAPE_STAT_SLOTS] = {0};
    HIPCHK(h, hipMemcpy(sh, h->d_stats, sizeof(sh), hipMemcpyDeviceToHost));
    for (int q = 0; q < GRAPE_STAT_SHARDS; ++q)
        for (int i = 0; i < GRAPE_STAT_SLOTS; ++i) st[i] += sh[q * GRAPE_STAT_SLOTS + i];
    const double N3 = (double)h->N * h->N * h->N, N2 = (double)h->N * h->N;
    const double cells = (double)h->K * h->N_T, ecells = (double)h->KC * h->N_T;
    // F_exp = (g + s) * 8 N^3 + (32/3) N^3, g = GEMMs of the Pade order (13:6, 9:5, 7:4, 5:3, 3:2)
    const double gemms = 2.0 * st[3] + 3.0 * st[4] + 4.0 * st[5] + 5.0 * st[6] + 6.0 * st[7];
    out[0] = cells;
    out[1] = (double)st[0];
    out[2] = (gemms + (double)st[0]) * 8.0 * N3 + ecells * (32.0 / 3.0) * N3;
    // derivative series: per order (1 + 2L) complex mat-vecs of 8 N^2 flop
    out[3] = (double)st[8] * (1.0 + 2.0 * h->L) * 8.0 * N2;
    if (n > 4) out[4] = (double)st[8];  // sum of series orders
    if (n > 5) out[5] = (double)st[9];  // cells solved by the pivoted fallback
    if (n > 6) out[6] = h->series ? 0.0 : ecells;   // propagators actually exponentiated (generator classes x time steps)
    if (n > 7) out[7] = (double)st[10];  // matrix-free propagator: series terms summed over both sweeps
    if (n > 8) out[8] = (double)st[11];  // ... and (sub-)steps
    // inverse-free exponential (grape_t18.hip.h): EXECUTED matrix-instruction flop (2048 per v_mfma_f64_16x16x4), its
    // own squarings and cells
    if (n > 9) out[9] = (double)st[12] * 2048.0;
    if (n > 10) out[10] = (double)st[13];
    if (n > 11) out[11] = (double)st[14];
    // mode of the ExpProp request: 0 the propagators are materialised, 1 they did not fit the device and the evaluation
    // runs matrix-free (see grape_create)
    if (n > 12) out[12] = h->u_fallback ? 1.0 : 0.0;
    if (n > 13) out[13] = (double)st[15];   // cells of [11] that took the four-product degree-16 route
    if (n > 14) out[14] = h->asm16p ? 3.0 : h->asm16 ? 1.0 : h->asm18gp ? 4.0 : h->asm18g ? 2.0 : 0.0;   // 1: the four-product route of this handle is the hand-allocated assembly kernel; 2: general matrices, expm_t18g_asm
    // which derivative kernel the ExpProp route of this handle launches: 0 a compiled one, 1 deriv3_asm, 2 deriv3s_asm (streamed
    // controls), 3 deriv3g_asm (general operators), 4 deriv4_asm (blocked path); [16]: the products of the blocked polynomial
    // route are lg_gemm_asm
    if (n > 15) {
        double kind = 0.0;
        if (!h->series && h->d_park3 && h->NT == 4) {
            if (h->deriv3_general) kind = 3.0;
            else if (h->L > 2) kind = 2.0;
            else if (!h->deriv3_h0g && h->deriv3_asm) kind = 1.0;
        } else if (!h->series && h->deriv4_blocks) kind = 4.0;
        out[15] = kind;
    }
    if (n > 16) out[16] = (h->large && !h->series && h->t18 && h->lg_asm) ? 1.0 : 0.0;
    // steps of the two sweeps the walks of the exponential kernel carried in the last evaluation (the sweep launch did
    // the other 2 K N_T - [17]; bench.py prices its HBM rate with that)
    if (n > 17) {
        double carried = 0.0;
        if (h->last_walk_fuse && h->d_prog) {
            std::vector<int> prog((size_t)2 * h->K);
            HIPCHK(h, hipMemcpy(prog.data(), h->d_prog, prog.size() * sizeof(int), hipMemcpyDeviceToHost));
            for (int k = 0; k < h->K; ++k) {
                if (h->last_walk_fuse & 1) carried += prog[(size_t)k];
                if (h->last_walk_fuse & 2) carried += prog[(size_t)h->K + k];
            }
        }
        out[17] = carried;
    }
    if (n > 18) out[18] = h->scan16 ? (double)h->scan_Bk : 0.0;   // block length of the scanned sweeps (N <= 16), 0: sequential sweeps
    return 4;
}
GRAPE_BARRIER(h ? &h->err : &g_create_error)

}  // extern "C"

namespace {
// grape_problem.ndev > 1: the trajectories are dealt to the devices in contiguous blocks whose sizes differ by at most
// one (the partition of SURVEY 8e), every block is an ordinary single-device handle created with the job's K_total.
int multi_create(grape_handle **out, const grape_problem *p) {
    const int G = std::min<int>(p->ndev, p->K);
    grape_handle *h = new grape_handle();
    HandleGuard guard(h);
    h->p = *p;
    h->no_target = p->target == nullptr;
    // ONE balancing similarity for the whole problem (the shards would each choose their own from their trajectories)
    const std::vector<double> bal_all = balance_of_problem(p);
    struct ForcedBal {   // (reset on every way out, exceptions included)
        explicit ForcedBal(const std::vector<double> *b) { tl_forced_bal = b; }
        ~ForcedBal() { tl_forced_bal = nullptr; }
    } forced(&bal_all);
    h->N = p->N; h->L = p->L; h->K = p->K; h->N_T = p->N_T;
    h->K_total = p->K_total > 0 ? p->K_total : p->K;
    h->device = p->devices ? p->devices[0] : p->device;
    h->h_multi.assign(8, 0.0);
    {
        const char *envm = getenv("GRAPE_MULTI_THREADS");
        h->multi_threads = !(envm && atoi(envm) == 0);
        const char *envk = getenv("GRAPE_TEST_HOOKS");
        h->test_hooks = envk && atoi(envk) == 1;
    }
    const size_t nn2 = (size_t)2 * p->N * p->N;
    const int base = p->K / G, rem = p->K % G;
    for (int g = 0; g < G; ++g) {
        const int lo = g * base + std::min(g, rem), Kg = base + (g < rem ? 1 : 0);
        grape_problem cp = *p;
        cp.ndev = 0; cp.devices = nullptr;
        cp.device = p->devices ? p->devices[g] : p->device + g;
        cp.K = Kg; cp.K_total = h->K_total;
        cp.H0 = p->H0 + (size_t)lo * nn2;
        if (p->hc_per_traj) cp.Hc = p->Hc + (size_t)lo * p->L * nn2;
        cp.psi0 = p->psi0 + (size_t)lo * 2 * p->N;
        cp.target = p->target ? p->target + (size_t)lo * 2 * p->N : nullptr;
        if (p->weights) cp.weights = p->weights + lo;
        if (p->Dpen && p->dpen_per_traj) cp.Dpen = p->Dpen + (size_t)lo * nn2;
        grape_handle *c = nullptr;
        const int rc = grape_create(&c, &cp);
        if (rc) {   // g_create_error already holds the child's message
            g_create_error = "device shard " + std::to_string(g) + " (device " + std::to_string(cp.device) + "): " + g_create_error;
            return rc;   // (the guard releases the shards built so far)
        }
        h->shards.push_back(c);
        h->shard_lo.push_back(lo);
        h->shard_dev.push_back(cp.device);
    }
    h->shard_lo.push_back(p->K);
    {   // RCCL for the two reductions when every shard has a device of its own
        const char *envr = getenv("GRAPE_MULTI_RCCL");
        bool distinct = true;
        for (int a = 0; a < G; ++a)
            for (int b = a + 1; b < G; ++b) distinct = distinct && h->shard_dev[a] != h->shard_dev[b];
        if (distinct && !(envr && atoi(envr) == 0) && rccl_api().ok) {
            // (the set and every buffer belong to the handle from the moment they exist: grape_destroy releases them
            // whatever happens next -- round-4 advisor finding: communicators of a half-built set leaked)
            h->comm_set = comm_set_acquire(h->shard_dev);
            if (h->comm_set) {
                h->use_rccl = true;
                h->d_red.assign((size_t)G, nullptr);
                for (int g = 0; g < G && h->use_rccl; ++g)
                    if (hipSetDevice(h->shard_dev[g]) != hipSuccess ||
                        hipMalloc((void **)&h->d_red[g], (8 + (size_t)p->L * p->N_T) * sizeof(double)) != hipSuccess) h->use_rccl = false;
                if (h->use_rccl && (hipSetDevice(h->shard_dev[0]) != hipSuccess || hipEventCreate(&h->ar0) != hipSuccess ||
                                    hipEventCreate(&h->ar1) != hipSuccess)) h->use_rccl = false;
            }
            (void)hipGetLastError();
        }
    }
    *out = guard.release();
    return GRAPE_OK;
}
}  // namespace
