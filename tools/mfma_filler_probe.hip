// Probe: what does ONE wave per SIMD pay for vector / LDS instructions placed between fp64 matrix instructions?
// Loop body = 4 independent v_mfma_f64_16x16x4_f64, each followed by F filler instructions of one kind.
// Output: cycles per MFMA (s_memtime) for F = 0..16 -- 64 means the fillers are hidden, 64 + c F means they are not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);exit(1);}}while(0)

template <int KIND, int F>
__device__ __forceinline__ void fillers(double (&x)[16], int (&y)[16], d4 &spare, const double *lds, int l) {
#pragma unroll
  for (int f = 0; f < F; ++f) {
    if (KIND == 0) asm volatile("v_mov_b32 %0, %1" : "=v"(y[f & 15]) : "v"(y[(f + 1) & 15]));
    if (KIND == 1) asm volatile("v_add_f64 %0, %1, %2" : "=v"(x[f & 15]) : "v"(x[(f + 1) & 15]), "v"(x[(f + 2) & 15]));
    if (KIND == 2) asm volatile("v_accvgpr_read_b32 %0, a40" : "=v"(y[f & 15]) :: "a40");
    if (KIND == 3) asm volatile("ds_read_b64 %0, %1" : "=v"(x[f & 15]) : "v"(l * 8));
    if (KIND == 4) asm volatile("v_add_u32 %0, %1, %2" : "=v"(y[f & 15]) : "v"(y[(f + 1) & 15]), "v"(y[(f + 2) & 15]));
    if (KIND == 5) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x[f & 15]) : "v"(x[(f + 1) & 15]), "v"(x[(f + 2) & 15]), "v"(x[(f + 3) & 15]));
    if (KIND == 6) asm volatile("s_mov_b32 s20, s21" ::: "s20");
    if (KIND == 7) asm volatile("v_accvgpr_write_b32 a41, %0" :: "v"(y[(f + 1) & 15]) : "a41");
  }
}

template <int KIND, int F, bool MFMA = true>
__global__ void __launch_bounds__(256) probe(double *out, unsigned long long *stamps, int iters) {
  __shared__ double lds[1024];
  int l = threadIdx.x;
  lds[l] = l; lds[l + 256] = l; lds[l + 512] = 1; lds[l + 768] = 2;
  __syncthreads();
  double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
  // the accumulators are fixed accumulation registers a[0:31], managed by hand (the compiler would carry them in vector
  // registers around the loop and copy them in and out every iteration)
#define CLOB "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31"
  asm volatile("v_accvgpr_write_b32 a0, 0\n v_accvgpr_write_b32 a1, 0\n v_accvgpr_write_b32 a2, 0\n v_accvgpr_write_b32 a3, 0\n v_accvgpr_write_b32 a4, 0\n v_accvgpr_write_b32 a5, 0\n v_accvgpr_write_b32 a6, 0\n v_accvgpr_write_b32 a7, 0\n"
               "v_accvgpr_write_b32 a8, 0\n v_accvgpr_write_b32 a9, 0\n v_accvgpr_write_b32 a10, 0\n v_accvgpr_write_b32 a11, 0\n v_accvgpr_write_b32 a12, 0\n v_accvgpr_write_b32 a13, 0\n v_accvgpr_write_b32 a14, 0\n v_accvgpr_write_b32 a15, 0\n"
               "v_accvgpr_write_b32 a16, 0\n v_accvgpr_write_b32 a17, 0\n v_accvgpr_write_b32 a18, 0\n v_accvgpr_write_b32 a19, 0\n v_accvgpr_write_b32 a20, 0\n v_accvgpr_write_b32 a21, 0\n v_accvgpr_write_b32 a22, 0\n v_accvgpr_write_b32 a23, 0\n"
               "v_accvgpr_write_b32 a24, 0\n v_accvgpr_write_b32 a25, 0\n v_accvgpr_write_b32 a26, 0\n v_accvgpr_write_b32 a27, 0\n v_accvgpr_write_b32 a28, 0\n v_accvgpr_write_b32 a29, 0\n v_accvgpr_write_b32 a30, 0\n v_accvgpr_write_b32 a31, 0\n" ::: CLOB);
  double x[16]; int y[16];
  for (int i = 0; i < 16; i++) { x[i] = 1.0 + i * 1e-3 + l * 1e-6; y[i] = i + l; }
  d4 spare = (d4){0, 0, 0, 0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (MFMA && i == 0) asm volatile("v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]" :: "v"(a), "v"(b) : CLOB);
      if (MFMA && i == 1) asm volatile("v_mfma_f64_16x16x4_f64 a[8:15], %0, %1, a[8:15]" :: "v"(a), "v"(b) : CLOB);
      if (MFMA && i == 2) asm volatile("v_mfma_f64_16x16x4_f64 a[16:23], %0, %1, a[16:23]" :: "v"(a), "v"(b) : CLOB);
      if (MFMA && i == 3) asm volatile("v_mfma_f64_16x16x4_f64 a[24:31], %0, %1, a[24:31]" :: "v"(a), "v"(b) : CLOB);
      fillers<KIND, F>(x, y, spare, lds, l);
    }
    if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  { int r0; asm volatile("s_nop 15\n s_nop 15\n v_accvgpr_read_b32 %0, a0" : "=v"(r0) :: CLOB); s += r0; }
  for (int i = 0; i < 16; i++) s += x[i] + y[i];
  out[blockIdx.x * blockDim.x + l] = s;
  if ((l & 63) == 0) stamps[blockIdx.x * 4 + (l >> 6)] = t1 - t0;
}

template <int KIND, int F>
void run(const char *name, double *out, unsigned long long *st, int nb) {
  const int iters = 20000;
  probe<KIND, F><<<nb, 256>>>(out, st, iters / 10); CK(hipDeviceSynchronize());
  probe<KIND, F><<<nb, 256>>>(out, st, iters); CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(nb * 4); CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  printf("%-22s F=%2d  %7.1f cycles per MFMA\n", name, F, (double)h[h.size() / 2] / (4.0 * iters));
}
template <int KIND, int F>
void run_bare(const char *name, double *out, unsigned long long *st, int nb) {
  const int iters = 20000;
  probe<KIND, F, false><<<nb, 256>>>(out, st, iters / 10); CK(hipDeviceSynchronize());
  probe<KIND, F, false><<<nb, 256>>>(out, st, iters); CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(nb * 4); CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  printf("%-22s alone (no MFMA), %2d per group: %6.2f cycles per instruction\n", name, F, (double)h[h.size() / 2] / (4.0 * iters * F));
}
#define ROW(K, NAME) run<K,0>(NAME,out,st,nb); run<K,1>(NAME,out,st,nb); run<K,2>(NAME,out,st,nb); run<K,4>(NAME,out,st,nb); run<K,8>(NAME,out,st,nb); run<K,12>(NAME,out,st,nb); run<K,16>(NAME,out,st,nb);
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); int nb = p.multiProcessorCount;
  double *out; CK(hipMalloc(&out, (size_t)nb * 256 * 8));
  unsigned long long *st; CK(hipMalloc(&st, (size_t)nb * 4 * 8));
  ROW(0, "v_mov_b32") ROW(4, "v_add_u32") ROW(2, "v_accvgpr_read_b32") ROW(7, "v_accvgpr_write_b32") ROW(1, "v_add_f64") ROW(5, "v_fma_f64") ROW(3, "ds_read_b64") ROW(6, "s_mov_b32")
  run_bare<0,16>("v_mov_b32",out,st,nb); run_bare<4,16>("v_add_u32",out,st,nb); run_bare<2,16>("v_accvgpr_read_b32",out,st,nb);
  run_bare<7,16>("v_accvgpr_write_b32",out,st,nb); run_bare<1,16>("v_add_f64",out,st,nb); run_bare<5,16>("v_fma_f64",out,st,nb);
  run_bare<3,8>("ds_read_b64",out,st,nb);
  return 0;
}
