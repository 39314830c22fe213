// Diagnostic: cycles per invert16 / gj_update on one wave (s_memtime deltas).  Not part of the product.
#include "../grape.jl_amd/csrc/grape_kernels.hip.h"
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) probe(double *out, unsigned long long *t, int reps) {
    __shared__ double pan[3 * 2 * 64 * 18];
    __shared__ double dv[3 * 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 3 * 2 * 64 * 18; i += 256) pan[i] = 0.001 * (i % 37);
    for (int i = tid; i < 1536; i += 256) dv[i] = 0.002 * (i % 17);
    __syncthreads();
    double ar[4], ai[4];
    for (int c = 0; c < 4; ++c) {
        const int i = lane & 15, j = 4 * c + (lane >> 4);
        ar[c] = (i == j ? 4.0 : 0.0) + 0.01 * ((i * 7 + j * 3) % 11);
        ai[c] = 0.02 * ((i * 5 + j) % 7);
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double mr = 0;
    if (wave == 0) {
        for (int r = 0; r < reps; ++r) {
            mr += invert16(ar, ai, lane, 1.0);
            asm volatile("" : "+v"(ar[0]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    Strip<4> S;
    for (int t_ = 0; t_ < 4; ++t_) for (int r = 0; r < 4; ++r) { S.re[t_][r] = 0.1 * lane + r; S.im[t_][r] = 0.01 * t_; }
    __syncthreads();
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        gj_update<4>(S, 1, pan, dv, lane);
        asm volatile("" : "+v"(S.re[0][0]));
    }
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { t[wave * 4 + 0] = t1 - t0; t[wave * 4 + 1] = t3 - t2; }
    double acc = mr;
    for (int c = 0; c < 4; ++c) acc += ar[c] + ai[c];
    for (int t_ = 0; t_ < 4; ++t_) for (int r = 0; r < 4; ++r) acc += S.re[t_][r] + S.im[t_][r];
    out[tid] = acc;
}
int main() {
    double *out; unsigned long long *t;
    hipMalloc(&out, 256 * 8); hipMalloc(&t, 16 * 8);
    const int reps = 50;
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, out, t, reps);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    printf("invert16: %.0f ticks each (wave 0);  gj_update: %.0f ticks each (4 waves concurrently)\n", (double)h[0] / reps, (double)h[1] / reps);
    // s_memtime runs at a fixed 100 MHz; convert with the shader clock
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("shader clock %d kHz -> invert16 %.0f cycles, gj_update %.0f cycles\n", clk, (double)h[0] / reps * clk / 1e5, (double)h[1] / reps * clk / 1e5);
    return 0;
}
