// Diagnostic: dependent-issue latencies (cycles per op in a serial chain, ONE wave on its SIMD) of the
// instructions the 16x16 tile inversion is made of.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
template <int MODE>
__global__ void probe(double *out, unsigned long long *t) {
    const int lane = threadIdx.x;
    double x = out[lane], y = out[64 + lane], z = 1.0000001;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int outer = 0; outer < 4; ++outer) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (MODE == 0) x = fma(x, z, y);                                         // dependent v_fma_f64
            if (MODE == 1) x = __builtin_amdgcn_rcp(x);                               // dependent v_rcp_f64
            if (MODE == 2) {                                                          // readlane -> VALU -> readlane
                int lo = __builtin_amdgcn_readlane(__double2loint(x), 5), hi = __builtin_amdgcn_readlane(__double2hiint(x), 5);
                x = fma(__hiloint2double(hi, lo), z, y);
            }
            if (MODE == 3) x = __shfl(x, (lane + 17) & 63, 64) * z;                   // ds_bpermute round trip (+1 mul)
            if (MODE == 4) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(z));
            if (MODE == 5) {                                                          // permlane32_swap + permlane16_swap broadcast of one dword pair
                int lo = __double2loint(x), hi = __double2hiint(x);
                int lo2 = lo, hi2 = hi;
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(lo2));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(hi), "+v"(hi2));
                int lo3 = lo, hi3 = hi;
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(lo3));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(hi), "+v"(hi3));
                x = __hiloint2double(hi, lo) * z;
            }
            if (MODE == 6) x = x * z;                                                 // dependent v_mul_f64
            if (MODE == 7) { float f = (float)x; f = f * 1.0001f + 0.5f; x = (double)f; }   // cvt + f32 fma + cvt
            if (MODE == 8) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x));   // 64-bit DPP move (dependent)
            if (MODE == 9) {                                                          // two 32-bit DPP moves + fma (what invert16 does per element)
                const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x153, 0xf, 0xf, true);
                const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x153, 0xf, 0xf, true);
                x = fma(__hiloint2double(hi, lo), z, y);
            }
            if (MODE == 10) {                                                         // 64-bit DPP move + fma
                double b = x;
                asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(b) : "v"(x));
                x = fma(b, z, y);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[128 + lane] = x;
    if (lane == 0) t[MODE] = t1 - t0;
}
int main() {
    double *out; unsigned long long *t;
    hipMalloc(&out, 4096); hipMalloc(&t, 128);
    double h[192]; for (int i = 0; i < 192; ++i) h[i] = 1.0 + 0.001 * i;
    hipMemcpy(out, h, sizeof(h), hipMemcpyHostToDevice);
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, out, t); hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, out, t);
        hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, out, t); hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, out, t);
        hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, out, t); hipLaunchKernelGGL(probe<5>, dim3(1), dim3(64), 0, 0, out, t);
        hipLaunchKernelGGL(probe<6>, dim3(1), dim3(64), 0, 0, out, t); hipLaunchKernelGGL(probe<7>, dim3(1), dim3(64), 0, 0, out, t);
        hipLaunchKernelGGL(probe<8>, dim3(1), dim3(64), 0, 0, out, t); hipLaunchKernelGGL(probe<9>, dim3(1), dim3(64), 0, 0, out, t);
        hipLaunchKernelGGL(probe<10>, dim3(1), dim3(64), 0, 0, out, t);
    }
    hipDeviceSynchronize();
    unsigned long long ht[16]; hipMemcpy(ht, t, 128, hipMemcpyDeviceToHost);
    const char *names[] = {"v_fma_f64 dependent", "v_rcp_f64 dependent", "readlane x2 + fma", "ds_bpermute x2 + mul", "v_fmac_f64_dpp (+s_nop 1)", "permlane32/16 swap bcast + mul", "v_mul_f64 dependent", "cvt f64->f32, fma f32, cvt back", "v_mov_b64_dpp dependent", "2x v_mov_b32_dpp + fma", "v_mov_b64_dpp + fma"};
    for (int m = 0; m < 11; ++m) printf("%-34s %7.1f cycles per iteration\n", names[m], (double)ht[m] / (4.0 * REP));
    return 0;
}
