// grape_t18.hip -- translation unit of the inverse-free exponential kernel (grape_t18.hip.h).
//
// Its own translation unit because it is compiled with -mllvm -amdgpu-mfma-vgpr-form: the partial products of the 3M
// scheme then live in the vector half of the register file, where the vector ALU combines them without
// v_accvgpr_read copies (8.4 cycles each, profiles/r03_mfma_f64_filler_probe.txt).  The switch crashes this compiler
// (ROCm 7.2, AMDGPURewriteAGPRCopyMFMA) on deriv_mfma_kernel, so the rest of the library is built without it.
// The device helpers of grape_kernels.hip.h are included into an anonymous namespace: this unit gets private copies of
// the constant tables and instantiates nothing but expm_t18_kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include <mutex>
#include <vector>
#include <stdlib.h>
#include <stdio.h>
namespace {
#include "grape_t18.hip.h"
#include "grape_deriv3.hip.h"
#include "grape_econ_coeffs.h"

template <int NT, bool SYM, bool CHEB, bool T16 = false>
hipError_t launch(const ExpmArgs &a, hipStream_t s, int blocks) {
    static size_t lds_set[64] = {0};
    const size_t lds = sizeof(double) * (size_t)T18Lds<NT>::TOTAL;
    int dev = 0;
    hipGetDevice(&dev);
    if (lds_set[dev & 63] < lds) {
        hipError_t e = hipFuncSetAttribute((const void *)expm_t18_kernel<NT, SYM, CHEB, T16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set[dev & 63] = lds;
    }
    hipLaunchKernelGGL((expm_t18_kernel<NT, SYM, CHEB, T16>), dim3(blocks), dim3(NT * 64), lds, s, a);
    return hipGetLastError();
}
template <int NT, int LMAX, bool H0G = false>
hipError_t launch_d3(const Deriv3Args &a, hipStream_t s, int blocks) {
    static size_t lds_set[64] = {0};
    // the operators actually present (a general drift with all its tiles)
    const size_t lds = sizeof(double) * ((H0G ? (size_t)NT * NT * 2 * D3Lds<NT>::TILE : (size_t)D3Lds<NT>::MAT) + (size_t)a.d.L * D3Lds<NT>::MAT);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    int dev = 0;
    hipGetDevice(&dev);
    if (lds_set[dev & 63] < lds) {
        hipError_t e = hipFuncSetAttribute((const void *)deriv3_kernel<NT, LMAX, H0G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set[dev & 63] = lds;
    }
    hipLaunchKernelGGL((deriv3_kernel<NT, LMAX, H0G>), dim3(blocks), dim3(256), lds, s, a);
    return hipGetLastError();
}
}  // namespace

namespace { hipError_t launch_d3_asm(const Deriv3Args &a, hipStream_t s, int blocks, bool general = false); }

// derivative overlaps, one wave per batch (grape_deriv3.hip.h): Hermitian operators whose upper tiles fit the LDS
extern "C" int grape_deriv3_launch(int NT, const void *d2args, size_t d2size, const double *H0f, const double *Hcf, int wpt,
                                   int skip_if_flagged, int h0_general, int asm_ok, void *stream, int blocks) {
    if (d2size != sizeof(Deriv2Args)) return (int)hipErrorInvalidValue;
    Deriv3Args a;
    memcpy(&a.d, d2args, sizeof(a.d));
    a.H0f = H0f; a.Hcf = Hcf; a.wpt = wpt; a.skip_if_flagged = skip_if_flagged;
    hipStream_t s = (hipStream_t)stream;
    const int L = a.d.L;
    if (L < 1 || L > 8) return (int)hipErrorInvalidValue;
    if (h0_general == 2) {   // general operators at four tiles per side: the streamed assembly kernel with all tiles
        if (NT != 4 || skip_if_flagged || a.d.gpark) return (int)hipErrorInvalidValue;
        return (int)launch_d3_asm(a, s, blocks, true);
    }
    if (h0_general) {   // general drift, Hermitian controls: three and four tiles per side, up to two controls
        if (L > 2) return (int)hipErrorInvalidValue;
        if (NT == 3) return (int)(L == 1 ? launch_d3<3, 1, true>(a, s, blocks) : launch_d3<3, 2, true>(a, s, blocks));
        if (NT == 4) return (int)(L == 1 ? launch_d3<4, 1, true>(a, s, blocks) : launch_d3<4, 2, true>(a, s, blocks));
        return (int)hipErrorInvalidValue;
    }
    // more than two controls where the operators still fit the LDS: N <= 32 up to eight, N <= 48 up to five
#define D3_CASES(NT_)                                                                                                   \
    if (NT == NT_) {                                                                                                    \
        if (L == 1) return (int)launch_d3<NT_, 1>(a, s, blocks);                                                        \
        if (L == 2) return (int)launch_d3<NT_, 2>(a, s, blocks);                                                        \
        if (L <= 4) return (int)launch_d3<NT_, 4>(a, s, blocks);                                                        \
        return (int)launch_d3<NT_, 8>(a, s, blocks);                                                                    \
    }
    D3_CASES(1) D3_CASES(2) D3_CASES(3)
#undef D3_CASES
    if (NT == 4 && L > 2) {     // the streamed-controls assembly kernel; no compiled twin (the operators do not fit the LDS)
        if (skip_if_flagged || a.d.gpark) return (int)hipErrorInvalidValue;
        return (int)launch_d3_asm(a, s, blocks);
    }
    if (NT == 4 && L <= 2) {
        // four tiles per side: the hand-allocated assembly kernel (asm/gen_d3.py); GRAPE_DERIV3_ASM=0 (read once in
        // grape_create and handed down as asm_ok) keeps the compiled kernel (same series, same stopping rule -- the
        // differential tests run both)
        if (asm_ok && !skip_if_flagged && !a.d.gpark) return (int)launch_d3_asm(a, s, blocks);
        return (int)(L == 1 ? launch_d3<4, 1>(a, s, blocks) : launch_d3<4, 2>(a, s, blocks));
    }
    return (int)hipErrorInvalidValue;
}

// ---- the four-product route as hand-allocated assembly (asm/gen_t16.py -> expm_t16_asm.co, embedded by asm_embed.S) ----
// Four tiles per side, Hermitian generators, controls shared by the trajectories (the cell fetches H0_k and the summed
// controls S_n).  The assembly kernel does the arithmetic and writes one verdict per cell (0: inside the spectral bound);
// t16_post_kernel behind it lists the cells beyond the bound for the five-product launch and books the statistics the
// C++ kernel books in its cell loop (credited work of Julia's exp!, executed matrix instructions).
extern "C" const unsigned char grape_asm_co_start[], grape_asm_co_end[];
namespace {
struct T16AsmArgs {           // kernel argument block of expm_t16_asm (gen_t16.py: KERNARG = 128 bytes)
    const double *H0f, *Sf, *dts;
    double2 *U;
    int *verdict;
    const int *rep;
    int KC, N_T, nblk;
    int fuse;                   // bit 0: ascending walks carry Psi along, bit 1: descending walks carry conj(chi) (round 5)
    unsigned long long *diag;   // diagnostic builds: stamp area
    const int *flags;           // flags[6]: cells predicted beyond the route's range (t16_plan_kernel)
    const int *wgtab;           // [nblk][4]: first cell, number of cells, step (+1 / -1), - of every workgroup's walk
    const double2 *xinit;       // [2][K][64]: Psi0_k; conj(target_k) / ||target_k||
    double2 *fw, *bw;           // [K][N_T + 1][64] stored states
    int *prog;                  // [2][K] steps each end of each trajectory was propagated by the walks
    int K, s_per_cell;          // s_per_cell: Sf is [KC][N_T] (control operators per trajectory) instead of [N_T]
    const int *splan;           // [KC * N_T] squarings planned per cell (t16_plan_kernel): the cell exponentiates A / 2^s
};
static_assert(sizeof(T16AsmArgs) == 136, "argument block of the assembly kernel");

// ||A||_1 of A = -i dt (H0_k + S_n) from the operator planes (t18_norm1 on the same numbers), all 256 threads
__device__ __forceinline__ double t16_post_norm1(const ExpmArgs &a, const int cell, double *red, const int tid) {
    constexpr int NP = 64;
    const int kc = cell / a.N_T, n = cell - kc * a.N_T, k = a.rep ? a.rep[kc] : kc;
    const double *h0 = a.H0f + (size_t)k * 2 * NP * NP, *sn = a.Sf + (size_t)(a.hc_per_traj ? cell : n) * 2 * NP * NP;
    const double dt = a.dts[n];
    const int j = tid & 63, part = tid >> 6;
    double sum = 0.;
    for (int i = part; i < NP; i += 4) {
        const double xr = dt * (h0[NP * NP + i * NP + j] + sn[NP * NP + i * NP + j]), xi = -dt * (h0[i * NP + j] + sn[i * NP + j]);
        sum += fast_sqrt(xr * xr + xi * xi);
    }
    red[tid] = sum;
    __syncthreads();
    if (tid < 64) {
        double c = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c = fmax(c, __shfl_xor(c, off, 64));
        if (tid == 0) red[256] = c;
    }
    __syncthreads();
    const double nA = red[256];
    __syncthreads();
    return nA;
}

// behind the assembly kernel, every evaluation: the cells beyond the bound go to the hand-over list; executed work
__global__ void __launch_bounds__(256) t16_post_kernel(ExpmArgs a, const int *verdict, const int *prog, int nprog, const int *splan) {
    const int tid = threadIdx.x, lane = tid & 63, ncell = a.K * a.N_T;
    if (t16_skipped(a.flags, ncell)) return;   // the route was not tried: the five-product launch books all cells
    if (prog && blockIdx.x == 0) {   // two matrix instructions per wave for every step a walk carried its state over (round 5)
        unsigned long long steps = 0;
        for (int i = tid; i < nprog; i += 256) steps += (unsigned long long)prog[i];
        for (int off = 32; off >= 1; off >>= 1) steps += __shfl_xor(steps, off, 64);
        if (lane == 0 && steps) stat_add(a.stats, 12, steps * 4ull * 2ull);
    }
    const int cell = blockIdx.x * 256 + tid;
    const bool valid = cell < ncell;
    const bool ok = valid && verdict[cell] == 0;
    if (valid && !ok) a.cell_list[atomicAdd(&a.flags[4], 1)] = cell;   // to be redone by the five-product launch
    const unsigned long long m_ok = __ballot(ok), m_valid = __ballot(valid);
    {   // the squarings the kernel executed (whatever the verdict): 192 matrix instructions per wave each
        unsigned long long sq = valid ? (unsigned long long)splan[cell] : 0ull;
        unsigned long long sq_ok = ok ? sq : 0ull;
        for (int off = 32; off >= 1; off >>= 1) { sq += __shfl_xor(sq, off, 64); sq_ok += __shfl_xor(sq_ok, off, 64); }
        if (lane == 0 && sq) stat_add(a.stats, 12, sq * 4ull * 192ull);
        if (lane == 0 && sq_ok) stat_add(a.stats, 13, sq_ok);   // squarings of cells this route finished
    }
    if (lane == 0 && m_valid) {
        // executed matrix instructions of the assembly kernel: four waves x (120 + 3 * 192 + 1 for the column sums) per cell
        stat_add(a.stats, 12, (unsigned long long)__popcll(m_valid) * 4ull * (unsigned long long)(T16Count<4>::CELL + 1));
        if (m_ok) {
            stat_add(a.stats, 14, (unsigned long long)__popcll(m_ok));
            stat_add(a.stats, 15, (unsigned long long)__popcll(m_ok));
        }
        atomicAdd(&a.flags[5], (int)__popcll(m_valid));
    }
}

// ON DEMAND (grape_get_work): the credited work of the cells the assembly kernel processed -- what Julia's exp! would do
// for them, order and squarings from ||A||_1, the rule of the cell loop of expm_t18_kernel.  Outside the certifying window
// of the operator-norm bound the norm has to be measured: 128 KB of operator planes per cell from L2, 0.6 ms at the
// headline configuration when done in every evaluation -- and nothing but grape_get_work reads the result.
__global__ void __launch_bounds__(256) t16_credit_kernel(ExpmArgs a) {
    __shared__ double red[264];
    __shared__ int todo[256], ntodo;
    __shared__ unsigned long long cnt[8];   // squarings, max, orders 0..4
    const int tid = threadIdx.x, lane = tid & 63, ncell = a.K * a.N_T;
    if (t16_skipped(a.flags, ncell)) return;
    if (tid < 8) cnt[tid] = 0;
    if (tid == 0) ntodo = 0;
    __syncthreads();
    auto order_of = [](double nA) { return nA > 2.1 ? 4 : nA > 0.95 ? 3 : nA > 0.25 ? 2 : nA > 0.015 ? 1 : 0; };
    auto squarings_of = [](double nA) {
        if (!(nA > 5.4)) return 0;
        const double r = nA / 5.4;
        const int e = ilogb(r);
        return (r == ldexp(1.0, e)) ? e : e + 1;
    };
    const int cell = blockIdx.x * 256 + tid;
    bool cert = false;
    if (cell < ncell) {
        const double bound = expm_norm_bound(a, cell);
        cert = bound > 2.1 && bound <= 5.4;
        if (!cert) todo[atomicAdd(&ntodo, 1)] = cell;
    }
    const unsigned long long m_cert = __ballot(cert);
    if (lane == 0 && m_cert) atomicAdd(&cnt[2 + 4], (unsigned long long)__popcll(m_cert));   // certified: order 13, no squaring
    __syncthreads();
    const int nt = ntodo;
    for (int q = 0; q < nt; ++q) {          // cells outside the certifying window: the measured norm
        const double nA = t16_post_norm1(a, todo[q], red, tid);
        if (tid == 0) {
            const int sj = squarings_of(nA);
            if (sj) { cnt[0] += (unsigned long long)sj; cnt[1] = max(cnt[1], (unsigned long long)sj); }
            cnt[2 + order_of(nA)] += 1ull;
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (cnt[0]) stat_add(a.stats, 0, cnt[0]);
        for (int o = 0; o < 5; ++o)
            if (cnt[2 + o]) stat_add(a.stats, 3 + o, cnt[2 + o]);
        if (cnt[1]) atomicMax(&a.flags[1], (int)cnt[1]);
    }
}

constexpr int D3_INV_TABLE = 2048;
constexpr int D3_PAIRS_OFF = 32768, D3_ECON_OFF = 32768, D3_ECON_TAB_B = 512;   // (gen_d3.py: PAIRS_OFF, ECON_OFF, ECON_TAB_B)
struct AsmModule {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr, fn_d3 = nullptr, fn_d3s = nullptr, fn_lg = nullptr, fn_d3g = nullptr, fn_d4[2] = {nullptr, nullptr};
    hipFunction_t fn_t18g = nullptr, fn_t16p = nullptr, fn_t16p4 = nullptr, fn_t18gp = nullptr;
    double *inv = nullptr;      // 1 / m, m < D3_INV_TABLE
};
hipError_t asm_function(int dev, hipFunction_t *fn, hipFunction_t *fn_d3 = nullptr, const double **inv = nullptr,
                        hipFunction_t *fn_d3s = nullptr, hipFunction_t *fn_lg = nullptr, hipFunction_t *fn_d3g = nullptr,
                        hipFunction_t *fn_d4 = nullptr, int d4_index = 0, hipFunction_t *fn_t18g = nullptr,
                        hipFunction_t *fn_t16p = nullptr, int t16p_slots = 2) {
    static AsmModule mods[64];
    static std::mutex mtx;
    std::lock_guard<std::mutex> lock(mtx);
    AsmModule &m = mods[dev & 63];
    if (!m.fn) {
        // GRAPE_ASM_CO=<file>: a code object from a file instead of the embedded one (tools/lg_ablate.sh: timing variants of
        // the generators without rebuilding the library; read once, when the module of a device is first loaded)
        static std::vector<char> override_co;
        const void *image = (const void *)grape_asm_co_start;
        if (const char *path = getenv("GRAPE_ASM_CO")) {
            if (override_co.empty()) {
                if (FILE *f = fopen(path, "rb")) {
                    fseek(f, 0, SEEK_END);
                    const long n = ftell(f);
                    fseek(f, 0, SEEK_SET);
                    if (n > 0) {
                        override_co.resize((size_t)n);
                        if (fread(override_co.data(), 1, (size_t)n, f) != (size_t)n) override_co.clear();
                    }
                    fclose(f);
                }
            }
            if (override_co.empty()) return hipErrorFileNotFound;
            image = override_co.data();
        }
        hipError_t e = hipModuleLoadData(&m.mod, image);
        if (e != hipSuccess) return e;
        e = hipModuleGetFunction(&m.fn_d3, m.mod, "deriv3_asm");
        if (e != hipSuccess) return e;
        e = hipModuleGetFunction(&m.fn_d3s, m.mod, "deriv3s_asm");
        if (e != hipSuccess) return e;
        e = hipModuleGetFunction(&m.fn_lg, m.mod, "lg_gemm_asm");
        if (e != hipSuccess) return e;
        e = hipModuleGetFunction(&m.fn_d3g, m.mod, "deriv3g_asm");
        if (e != hipSuccess) return e;
        e = hipModuleGetFunction(&m.fn_d4[0], m.mod, "deriv4_asm_128");
        if (e != hipSuccess) return e;
        e = hipModuleGetFunction(&m.fn_d4[1], m.mod, "deriv4_asm_256");
        if (e != hipSuccess) return e;
        e = hipModuleGetFunction(&m.fn_t18g, m.mod, "expm_t18g_asm");
        if (e != hipSuccess) return e;
        e = hipModuleGetFunction(&m.fn_t16p, m.mod, "expm_t16p_asm");
        if (e != hipSuccess) return e;
        e = hipModuleGetFunction(&m.fn_t16p4, m.mod, "expm_t16p4_asm");
        if (e != hipSuccess) return e;
        e = hipModuleGetFunction(&m.fn_t18gp, m.mod, "expm_t18gp_asm");
        if (e != hipSuccess) return e;
        // 1 / m for the series orders (the kernels read them with scalar loads; gfx9 has no scalar floating point); behind
        // them the piece table of the streamed kernel (gen_d3s.py piece_table: source offset of piece 4 tile + 2 plane + half)
        // round 6: at D3_PAIRS_OFF bytes the scalars (omega_a, sigma_a) of pass 2, a < D3_INV_TABLE, of the Taylor series (both
        // 1 / (a + 1)); D3_ECON_OFF bytes further those of the economized series (grape_econ_coeffs.h, tools/econ_coeffs.py)
        static_assert(D3_PAIRS_OFF >= (D3_INV_TABLE + 20) * 8 && D3_ECON_OFF == D3_INV_TABLE * 16 && ECON_MAXDEG <= 31, "layout of gen_d3.py");
        std::vector<double> tab((D3_PAIRS_OFF + D3_ECON_OFF + 16 * D3_ECON_TAB_B) / 8);
        tab[0] = 0.0;
        for (int i = 1; i < D3_INV_TABLE; ++i) tab[i] = 1.0 / (double)i;
        for (int a = 0; a < D3_INV_TABLE; ++a) tab[D3_PAIRS_OFF / 8 + 2 * a] = tab[D3_PAIRS_OFF / 8 + 2 * a + 1] = 1.0 / (double)(a + 1);
        for (int i = 0; i < ECON_NSETS; ++i)   // the polynomial of degree M at (M - 16) D3_ECON_TAB_B
            for (int a = 0; a < ECON_DEG[i]; ++a)
                for (int j = 0; j < 2; ++j)
                    tab[(D3_PAIRS_OFF + D3_ECON_OFF + (ECON_DEG[i] - 16) * D3_ECON_TAB_B) / 8 + 2 * a + j] = ECON_TABS[i][a][j];
        int *pt = (int *)(tab.data() + D3_INV_TABLE);
        int np_ = 0;
        for (int ti = 0; ti < 4; ++ti)
            for (int tj = ti; tj < 4; ++tj)
                for (int plane = 0; plane < 2; ++plane)
                    for (int half = 0; half < 2; ++half) pt[np_++] = plane * 64 * 64 * 8 + ti * 16 * 64 * 8 + tj * 16 * 8;
        e = hipMalloc((void **)&m.inv, sizeof(double) * tab.size());
        if (e != hipSuccess) return e;
        e = hipMemcpy(m.inv, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
        e = hipModuleGetFunction(&m.fn, m.mod, "expm_t16_asm");
        if (e != hipSuccess) return e;
    }
    if (fn) *fn = m.fn;
    if (fn_d3) *fn_d3 = m.fn_d3;
    if (fn_d3s) *fn_d3s = m.fn_d3s;
    if (fn_lg) *fn_lg = m.fn_lg;
    if (fn_d3g) *fn_d3g = m.fn_d3g;
    if (fn_d4) *fn_d4 = m.fn_d4[d4_index & 1];
    if (fn_t18g) *fn_t18g = m.fn_t18g;
    if (fn_t16p) *fn_t16p = t16p_slots == 4 ? m.fn_t16p4 : t16p_slots == 0 ? m.fn_t18gp : m.fn_t16p;   // (0: the general-matrix variant)
    if (inv) *inv = m.inv;
    return hipSuccess;
}
}  // namespace

namespace {
struct D3AsmArgs {            // kernel argument block of deriv3_asm (gen_d3.py: KERNARG = 160 bytes)
    const double *H0f, *Hcf, *eps, *shape, *dts;
    const double2 *fw, *bw;
    const double *rho;
    double2 *tg;
    double *park;             // [blocks][4][slots][NP][16] complex
    int *flags;
    unsigned long long *stats;
    const int *batch_flag;
    const double *inv;        // 1 / m
    int K, L, N_T, hc_per_traj, wpt, batches_per_k, mcap, slots;
    double tol2;
    int deep, nblocks;
};
static_assert(sizeof(D3AsmArgs) == 160, "argument block of the assembly kernel");

hipError_t launch_d3_asm(const Deriv3Args &g, hipStream_t s, int blocks, bool general) {
    const Deriv2Args &a = g.d;
    if (a.L < 1 || a.L > (general ? 7 : 8) || blocks < 1 || a.maxm + 2 > D3_INV_TABLE) return hipErrorInvalidValue;
    if ((long)a.N_T + 1 >= (1L << 21) || (long)a.K * a.L * a.N_T >= (1L << 27)) return hipErrorInvalidValue;   // (32-bit offsets in the kernel)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipFunction_t fn, fn_s, fn_g;
    const double *inv;
    e = asm_function(dev, nullptr, &fn, &inv, &fn_s, nullptr, &fn_g);
    if (e != hipSuccess) return e;
    if (a.L > 2) fn = fn_s;     // more than two controls: the operators stream through the LDS (asm/gen_d3s.py)
    if (general) fn = fn_g;     // general operators: all tiles, streamed; pass 2 applies the adjoint
    D3AsmArgs k{};
    k.H0f = g.H0f; k.Hcf = g.Hcf; k.eps = a.eps; k.shape = a.shape; k.dts = a.dts; k.fw = a.fw; k.bw = a.bw; k.rho = a.rho;
    k.tg = a.tg; k.park = a.park; k.flags = a.flags; k.stats = a.stats; k.batch_flag = a.batch_flag; k.inv = inv;
    k.K = a.K; k.L = a.L; k.N_T = a.N_T; k.hc_per_traj = a.hc_per_traj; k.wpt = g.wpt; k.batches_per_k = a.batches_per_k;
    // the parking area holds maxm + 1 terms per wave (grape_create): every order the kernel forms has a slot
    k.mcap = a.max_order < a.maxm ? a.max_order : a.maxm;
    k.slots = a.maxm + 1;
    k.tol2 = a.tol * a.tol;
    k.deep = (a.deep_redo && a.max_order > k.mcap) ? 1 : 0;
    // bit 1: flags of the economized series behind the batch flags (Hermitian kernels; the orders pass 1 forms must fit)
    if (a.batch_econ && !general && a.batch_flag && k.mcap >= ECON_MAXDEG) k.deep |= 2;
    k.nblocks = blocks;
    size_t size = sizeof(k);
    void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(fn, (unsigned)blocks, 1, 1, 256, 1, 1, 0, s, nullptr, cfg);
}
}  // namespace

// the tables of the economized derivative series on the current device, for the compiled kernels: degree M at
// 64 (M - 16) doubles from the pointer, (omega_a, sigma_a); nullptr when the module cannot be loaded
extern "C" const double *grape_econ_pairs(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    hipFunction_t fn;
    const double *inv = nullptr;
    if (asm_function(dev, nullptr, &fn, &inv) != hipSuccess || !inv) return nullptr;
    static_assert(D3_ECON_TAB_B == 64 * 8, "64 doubles per degree");
    return inv + (D3_PAIRS_OFF + D3_ECON_OFF) / 8;
}

// derivative overlaps of the blocked path (asm/gen_d4.py): one workgroup per batch, operator fragments with three planes
namespace {
struct D4AsmArgs {            // kernel argument block of deriv4_asm_{128,256} (gen_d4.py: KERNARG = 176 bytes)
    const double *H0q, *Hcq, *H0p, *Hcp, *eps, *shape, *dts;
    const double2 *fw, *bw;
    const double *rho;
    double2 *tg;
    double *park;             // [blocks][slots][NP][16] complex
    int *flags;
    unsigned long long *stats;
    const int *batch_flag;
    const double *inv;        // 1 / m
    int K, L, N_T, hc_per_traj, nbatch_total, batches_per_k, mcap, slots;
    double tol2;
    int deep, nblocks;
};
static_assert(sizeof(D4AsmArgs) == 176, "argument block of the assembly kernel");
}  // namespace
extern "C" int grape_deriv4_launch(int NP, const void *d2args, size_t d2size, const double *H0q3, const double *Hcq3, const double *H0p3,
                                   const double *Hcp3, void *stream, int blocks) {
    if (d2size != sizeof(Deriv2Args) || (NP != 128 && NP != 256) || blocks < 1) return (int)hipErrorInvalidValue;
    Deriv2Args a;
    memcpy(&a, d2args, sizeof(a));
    if (a.L < 1 || a.L > 8 || a.gpark || a.maxm + 2 > D3_INV_TABLE || !H0q3 || !Hcq3 || !H0p3 || !Hcp3) return (int)hipErrorInvalidValue;
    if ((long)a.N_T + 1 >= (1L << 19) || (long)a.K * a.L * a.N_T >= (1L << 27) || a.nbatch_total >= (1 << 24)) return (int)hipErrorInvalidValue;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipFunction_t fn;
    const double *inv;
    e = asm_function(dev, nullptr, nullptr, &inv, nullptr, nullptr, nullptr, &fn, NP == 256);
    if (e != hipSuccess) return (int)e;
    D4AsmArgs k{};
    k.H0q = H0q3; k.Hcq = Hcq3; k.H0p = H0p3; k.Hcp = Hcp3; k.eps = a.eps; k.shape = a.shape; k.dts = a.dts; k.fw = a.fw; k.bw = a.bw;
    k.rho = a.rho; k.tg = a.tg; k.park = a.park; k.flags = a.flags; k.stats = a.stats; k.batch_flag = a.batch_flag; k.inv = inv;
    k.K = a.K; k.L = a.L; k.N_T = a.N_T; k.hc_per_traj = a.hc_per_traj; k.nbatch_total = a.nbatch_total; k.batches_per_k = a.batches_per_k;
    k.mcap = a.max_order < a.maxm ? a.max_order : a.maxm;
    k.slots = a.maxm + 1;
    k.tol2 = a.tol * a.tol;
    k.deep = (a.batch_econ && a.batch_flag && k.mcap >= ECON_MAXDEG) ? 2 : 0;   // bit 1: degrees of the economized series behind the batch flags
    k.nblocks = blocks;
    size_t size = sizeof(k);
    void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    return (int)hipModuleLaunchKernel(fn, (unsigned)blocks, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, nullptr, cfg);
}

// batched complex block product of the blocked path (asm/gen_lg.py): `k` is the 416-byte argument block of lg_gemm_asm,
// filled by the caller (grape_hip.hip: lg_asm_args), one workgroup per 64 x 64 output block
extern "C" int grape_lg_asm_launch(const void *k, size_t size, unsigned blocks, void *stream) {
    if (size != 416 || blocks == 0) return (int)hipErrorInvalidValue;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipFunction_t fn;
    e = asm_function(dev, nullptr, nullptr, nullptr, nullptr, &fn);
    if (e != hipSuccess) return (int)e;
    unsigned char buf[416];
    memcpy(buf, k, sizeof(buf));
    size_t sz = sizeof(buf);
    void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, buf, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    return (int)hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, nullptr, cfg);
}

// The deal of the cells of expm_t16_asm (round 5): workgroup b walks the contiguous range [b ncell / nblk, (b + 1) ncell / nblk)
// of the flattened index kc N_T + n -- ascending when the range begins with the first step of a trajectory (it can carry
// Psi along from t = 0), descending from its last cell when it ends with the last step of one (conj(chi) from t = T), else
// ascending.  At the headline shape (128 trajectories, 256 workgroups): one half of a trajectory each, both anchored.
// tab: [nblk][4] = first cell, count, step, 0 (tests/test_asm_kernel.py t16_walks is the same rule)
extern "C" void grape_t16_walks(int KC, int N_T, int nblk, int *tab) {
    const long ncell = (long)KC * N_T;
    for (int b = 0; b < nblk; ++b) {
        const long lo = b * ncell / nblk, hi = (b + 1) * ncell / nblk;
        int *e = tab + 4 * b;
        e[3] = 0;
        if (hi == lo) { e[0] = 0; e[1] = 0; e[2] = 1; }
        else if (lo % N_T == 0 || hi % N_T != 0) { e[0] = (int)lo; e[1] = (int)(hi - lo); e[2] = 1; }
        else { e[0] = (int)(hi - 1); e[1] = (int)(hi - lo); e[2] = -1; }
    }
}

// args: ExpmArgs with Sf set (summed controls of every time step) and cell_list / flags / stats as for the C++ kernel;
// walk: {wgtab, xinit, fw, bw, prog, splan} device pointers, fuse / K as in T16AsmArgs
extern "C" int grape_t16_asm_launch(const void *args, size_t args_size, int *verdict, void *stream, int blocks,
                                    const void *const *walk, int fuse, int K) {
    if (args_size != sizeof(ExpmArgs)) return (int)hipErrorInvalidValue;
    ExpmArgs a;
    memcpy(&a, args, sizeof(a));
    if (!a.Sf || !verdict || blocks < 1 || !walk || !walk[0] || !walk[5]) return (int)hipErrorInvalidValue;
    if (fuse && (!walk[1] || !walk[2] || !walk[3] || !walk[4] || a.rep || K != a.K)) return (int)hipErrorInvalidValue;
    const long ncell = (long)a.K * a.N_T;
    if (ncell <= 0 || ncell >= (1L << 28)) return (int)hipErrorInvalidValue;   // (32-bit cell arithmetic in the kernel)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipFunction_t fn;
    e = asm_function(dev, &fn);
    if (e != hipSuccess) return (int)e;
    T16AsmArgs k{};
    k.H0f = a.H0f; k.Sf = a.Sf; k.dts = a.dts; k.U = a.U; k.verdict = verdict; k.rep = a.rep;
    k.KC = a.K; k.N_T = a.N_T; k.nblk = blocks; k.flags = a.flags;
    k.fuse = fuse; k.wgtab = (const int *)walk[0]; k.xinit = (const double2 *)walk[1]; k.fw = (double2 *)walk[2];
    k.bw = (double2 *)walk[3]; k.prog = (int *)walk[4]; k.K = K; k.splan = (const int *)walk[5];
    k.s_per_cell = a.hc_per_traj ? 1 : 0;
    size_t size = sizeof(k);
    void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    hipStream_t s = (hipStream_t)stream;
    e = hipModuleLaunchKernel(fn, (unsigned)blocks, 1, 1, 256, 1, 1, 0, s, nullptr, cfg);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(t16_post_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, s, a, (const int *)verdict,
                       fuse ? (const int *)walk[4] : (const int *)nullptr, 2 * K, (const int *)walk[5]);
    return (int)hipGetLastError();
}

// ---- control operators per trajectory, Hermitian generators, one or two controls (asm/gen_t16p.py -> expm_t16p_asm): the
// cell fetches H0_k and the control operators of its trajectory; `dte`: [N_T][4] = dt, e1, e2, - of every time step (built per
// evaluation by the caller); three or four controls: expm_t16p4_asm with rows of 8 (dt, e1 .. e4, -, -, -).  Same verdicts,
// hand-over list and statistics as expm_t16_asm (t16_post_kernel).
extern "C" int grape_t16p_asm_launch(const void *args, size_t args_size, int *verdict, void *stream, int blocks,
                                     const void *const *walk, int fuse, int K, const double *dte) {
    if (args_size != sizeof(ExpmArgs)) return (int)hipErrorInvalidValue;
    ExpmArgs a;
    memcpy(&a, args, sizeof(a));
    if (!a.hc_per_traj || a.L < 1 || a.L > 4 || !dte || !verdict || blocks < 1 || !walk || !walk[0] || !walk[5]) return (int)hipErrorInvalidValue;
    if (fuse && (!walk[1] || !walk[2] || !walk[3] || !walk[4] || a.rep || K != a.K)) return (int)hipErrorInvalidValue;
    const long ncell = (long)a.K * a.N_T;
    if (ncell <= 0 || ncell >= (1L << 28)) return (int)hipErrorInvalidValue;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipFunction_t fn;
    e = asm_function(dev, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, &fn, a.L <= 2 ? 2 : 4);
    if (e != hipSuccess) return (int)e;
    T16AsmArgs k{};
    k.H0f = a.H0f; k.Sf = a.Hcf; k.dts = dte; k.U = a.U; k.verdict = verdict; k.rep = a.rep;
    k.KC = a.K; k.N_T = a.N_T; k.nblk = blocks; k.flags = a.flags;
    k.fuse = fuse; k.wgtab = (const int *)walk[0]; k.xinit = (const double2 *)walk[1]; k.fw = (double2 *)walk[2];
    k.bw = (double2 *)walk[3]; k.prog = (int *)walk[4]; k.K = K; k.splan = (const int *)walk[5];
    k.s_per_cell = a.L;   // (this kernel reads the field as the number of controls)
    size_t size = sizeof(k);
    void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    hipStream_t s = (hipStream_t)stream;
    e = hipModuleLaunchKernel(fn, (unsigned)blocks, 1, 1, 256, 1, 1, 0, s, nullptr, cfg);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(t16_post_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, s, a, (const int *)verdict,
                       fuse ? (const int *)walk[4] : (const int *)nullptr, 2 * K, (const int *)walk[5]);
    return (int)hipGetLastError();
}

// ---- general matrices: the five-product cell with its own scaling decision (asm/gen_t18g.py -> expm_t18g_asm) ----
// Same argument block and walk tables as expm_t16_asm; the kernel writes verdict[cell] (2: norms not finite) and the number
// of squarings it chose into splan[cell].  Behind it: executed work (960 products + 2 column sums + 192 per squaring matrix
// instructions per wave and cell, two per carried state), squarings, cells; a cell that is not finite raises bit 6.
namespace {
__global__ void __launch_bounds__(256) t18g_post_kernel(ExpmArgs a, const int *verdict, const int *prog, int nprog, const int *splan) {
    const int tid = threadIdx.x, lane = tid & 63, ncell = a.K * a.N_T;
    if (prog && blockIdx.x == 0) {
        unsigned long long steps = 0;
        for (int i = tid; i < nprog; i += 256) steps += (unsigned long long)prog[i];
        for (int off = 32; off >= 1; off >>= 1) steps += __shfl_xor(steps, off, 64);
        if (lane == 0 && steps) stat_add(a.stats, 12, steps * 4ull * 2ull);
    }
    const int cell = blockIdx.x * 256 + tid;
    const bool valid = cell < ncell;
    const bool bad = valid && verdict[cell] != 0;
    unsigned long long sq = valid ? (unsigned long long)splan[cell] : 0ull;
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const unsigned long long m_valid = __ballot(valid), m_bad = __ballot(bad);
    if (lane == 0 && m_valid) {
        const unsigned long long nc = (unsigned long long)__popcll(m_valid);
        stat_add(a.stats, 12, 4ull * (nc * (5ull * 192ull + 2ull) + sq * 192ull));
        if (sq) stat_add(a.stats, 13, sq);
        stat_add(a.stats, 14, nc);
        if (m_bad) atomicOr(&a.flags[0], 64);
    }
}
}  // namespace

// dte != nullptr: control operators per trajectory, one or two controls -- expm_t18gp_asm fetches them itself (asm/gen_t18gp.py;
// table as for expm_t16p_asm); otherwise the summed controls a.Sf (per time step, or per cell with operators per trajectory)
extern "C" int grape_t18g_asm_launch(const void *args, size_t args_size, int *verdict, void *stream, int blocks,
                                     const void *const *walk, int fuse, int K, const double *dte) {
    if (args_size != sizeof(ExpmArgs)) return (int)hipErrorInvalidValue;
    ExpmArgs a;
    memcpy(&a, args, sizeof(a));
    if ((!a.Sf && !dte) || !verdict || blocks < 1 || !walk || !walk[0] || !walk[5]) return (int)hipErrorInvalidValue;
    if (dte && (!a.hc_per_traj || a.L < 1 || a.L > 2)) return (int)hipErrorInvalidValue;
    if (fuse && (!walk[1] || !walk[2] || !walk[3] || !walk[4] || a.rep || K != a.K)) return (int)hipErrorInvalidValue;
    const long ncell = (long)a.K * a.N_T;
    if (ncell <= 0 || ncell >= (1L << 28)) return (int)hipErrorInvalidValue;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipFunction_t fn;
    if (dte) e = asm_function(dev, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, &fn, 0);
    else e = asm_function(dev, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &fn);
    if (e != hipSuccess) return (int)e;
    T16AsmArgs k{};
    k.H0f = a.H0f; k.Sf = dte ? a.Hcf : a.Sf; k.dts = dte ? dte : a.dts; k.U = a.U; k.verdict = verdict; k.rep = a.rep;
    k.KC = a.K; k.N_T = a.N_T; k.nblk = blocks; k.flags = a.flags;
    k.fuse = fuse; k.wgtab = (const int *)walk[0]; k.xinit = (const double2 *)walk[1]; k.fw = (double2 *)walk[2];
    k.bw = (double2 *)walk[3]; k.prog = (int *)walk[4]; k.K = K; k.splan = (const int *)walk[5];
    k.s_per_cell = dte ? a.L : (a.hc_per_traj ? 1 : 0);
    size_t size = sizeof(k);
    void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    hipStream_t s = (hipStream_t)stream;
    e = hipModuleLaunchKernel(fn, (unsigned)blocks, 1, 1, 256, 1, 1, 0, s, nullptr, cfg);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(t18g_post_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, s, a, (const int *)verdict,
                       fuse ? (const int *)walk[4] : (const int *)nullptr, 2 * K, (const int *)walk[5]);
    return (int)hipGetLastError();
}

// credited statistics of the last evaluation of the assembly route, on demand (see t16_credit_kernel)
extern "C" int grape_t16_credit_launch(const void *args, size_t args_size, void *stream) {
    if (args_size != sizeof(ExpmArgs)) return (int)hipErrorInvalidValue;
    ExpmArgs a;
    memcpy(&a, args, sizeof(a));
    if (!a.Sf) return (int)hipErrorInvalidValue;
    const long ncell = (long)a.K * a.N_T;
    hipLaunchKernelGGL(t16_credit_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// args: the ExpmArgs of grape_kernels.hip.h (same header on both sides), passed as bytes because the type of this unit
// lives in an anonymous namespace
// t16: Hermitian generators -- the variant that takes the four-product route and lists the cells beyond its range
extern "C" int grape_t18_launch(int NT, int herm, int t16, const void *args, size_t args_size, void *stream, int blocks) {
    if (args_size != sizeof(ExpmArgs)) return (int)hipErrorInvalidValue;
    ExpmArgs a;
    memcpy(&a, args, sizeof(a));
    hipStream_t s = (hipStream_t)stream;
    // Hermitian generators: Chebyshev coefficient set and spectral scaling for every size, the tile symmetry from three
    // tiles per side on; general matrices: Taylor set
    switch (NT) {
        case 1: return (int)(herm ? (t16 ? launch<1, false, true, true>(a, s, blocks) : launch<1, false, true>(a, s, blocks)) : launch<1, false, false>(a, s, blocks));
        case 2: return (int)(herm ? (t16 ? launch<2, false, true, true>(a, s, blocks) : launch<2, false, true>(a, s, blocks)) : launch<2, false, false>(a, s, blocks));
        case 3: return (int)(herm ? (t16 ? launch<3, true, true, true>(a, s, blocks) : launch<3, true, true>(a, s, blocks)) : launch<3, false, false>(a, s, blocks));
        default: return (int)(herm ? (t16 ? launch<4, true, true, true>(a, s, blocks) : launch<4, true, true>(a, s, blocks)) : launch<4, false, false>(a, s, blocks));
    }
}
#ifdef GRAPE_DIAG
extern "C" void grape_t18_set_stamps(unsigned long long *d_stamps, void *stream) {
    hipMemcpyToSymbolAsync(HIP_SYMBOL(g_diag_slot_base), &d_stamps, sizeof(d_stamps), 0, hipMemcpyHostToDevice, (hipStream_t)stream);
}
#endif
