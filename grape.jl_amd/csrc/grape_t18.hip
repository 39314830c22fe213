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
namespace {
#include "grape_t18.hip.h"
#include "grape_deriv3.hip.h"

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

// derivative overlaps, one wave per batch (grape_deriv3.hip.h): Hermitian operators whose upper tiles fit the LDS
extern "C" int grape_deriv3_launch(int NT, const void *d2args, size_t d2size, const double *H0f, const double *Hcf, int wpt,
                                   int skip_if_flagged, int h0_general, void *stream, int blocks) {
    if (d2size != sizeof(Deriv2Args)) return (int)hipErrorInvalidValue;
    Deriv3Args a;
    memcpy(&a.d, d2args, sizeof(a.d));
    a.H0f = H0f; a.Hcf = Hcf; a.wpt = wpt; a.skip_if_flagged = skip_if_flagged;
    hipStream_t s = (hipStream_t)stream;
    const int L = a.d.L;
    if (L < 1 || L > 8) return (int)hipErrorInvalidValue;
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
    if (NT == 4 && L <= 2) return (int)(L == 1 ? launch_d3<4, 1>(a, s, blocks) : launch_d3<4, 2>(a, s, blocks));
    return (int)hipErrorInvalidValue;
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
