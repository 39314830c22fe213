// Probe: v_mfma_f64_16x16x4_f64 lane layout check + issue-rate microbenchmark on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);exit(1);}}while(0)

__global__ void layout_kernel(const double* A /*16x4 row-major*/, const double* B /*4x16 row-major*/, double* D /*16x16 row-major*/){
  int l = threadIdx.x;
  double a = A[(l&15)*4 + (l>>4)];
  double b = B[(l>>4)*16 + (l&15)];
  d4 c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a,b,c,0,0,0);
  for(int r=0;r<4;r++) D[((l>>4)+4*r)*16 + (l&15)] = c[r];
}

template<int NACC>
__global__ void __launch_bounds__(256) rate_kernel(double* out, int iters){
  int l = threadIdx.x;
  double a = 1.0 + 1e-9*l, b = 1.0 - 1e-9*l;
  d4 acc[NACC];
  for(int i=0;i<NACC;i++) acc[i] = (d4){0,0,0,0};
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int i=0;i<NACC;i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
  }
  double s=0; for(int i=0;i<NACC;i++) s += acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  out[blockIdx.x*blockDim.x + l] = s;
}

__global__ void __launch_bounds__(256) dfma_kernel(double* out, int iters){
  double x[16]; for(int i=0;i<16;i++) x[i]=1.0+i*1e-3+threadIdx.x*1e-6;
  double a=1.0000001, b=1e-9;
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int i=0;i<16;i++) x[i] = __builtin_fma(x[i],a,b);
  }
  double s=0; for(int i=0;i<16;i++) s+=x[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}

__global__ void copy_kernel(const double4* __restrict__ in, double4* __restrict__ out, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x*blockDim.x;
  for(; i<n; i+=stride) out[i]=in[i];
}

int main(){
  // layout check
  std::vector<double> A(64),B(64),D(256),R(256,0.0);
  for(int i=0;i<16;i++)for(int k=0;k<4;k++)A[i*4+k]= (i+1)*0.5 + k*7.0;
  for(int k=0;k<4;k++)for(int j=0;j<16;j++)B[k*16+j]= (k+1)*3.0 - j*j*0.25;
  for(int i=0;i<16;i++)for(int j=0;j<16;j++){double s=0;for(int k=0;k<4;k++)s+=A[i*4+k]*B[k*16+j];R[i*16+j]=s;}
  double *dA,*dB,*dD; CK(hipMalloc(&dA,64*8));CK(hipMalloc(&dB,64*8));CK(hipMalloc(&dD,256*8));
  CK(hipMemcpy(dA,A.data(),64*8,hipMemcpyHostToDevice));CK(hipMemcpy(dB,B.data(),64*8,hipMemcpyHostToDevice));
  layout_kernel<<<1,64>>>(dA,dB,dD); CK(hipDeviceSynchronize());
  CK(hipMemcpy(D.data(),dD,256*8,hipMemcpyDeviceToHost));
  double err=0; for(int i=0;i<256;i++) err=fmax(err,fabs(D[i]-R[i]));
  printf("layout max err = %g (%s)\n", err, err<1e-9?"OK":"MISMATCH");

  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  double* out; CK(hipMalloc(&out, 4096*256*8));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit=[&](auto fn, const char* name, double flops){
    fn(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1));
    printf("%-34s %8.3f ms  %8.2f TFLOP/s\n", name, ms, flops/ms*1e-9);
  };
  int iters=20000;
  int nb = p.multiProcessorCount;
  // 1 wave per SIMD (256 thr block, 1 block/CU)
  timeit([&]{rate_kernel<1><<<nb,256>>>(out,iters);},"mfma f64 1acc 1w/SIMD", (double)nb*4*iters*1*2048);
  timeit([&]{rate_kernel<2><<<nb,256>>>(out,iters);},"mfma f64 2acc 1w/SIMD", (double)nb*4*iters*2*2048);
  timeit([&]{rate_kernel<4><<<nb,256>>>(out,iters);},"mfma f64 4acc 1w/SIMD", (double)nb*4*iters*4*2048);
  timeit([&]{rate_kernel<8><<<nb,256>>>(out,iters);},"mfma f64 8acc 1w/SIMD", (double)nb*4*iters*8*2048);
  timeit([&]{rate_kernel<4><<<nb*2,256>>>(out,iters);},"mfma f64 4acc 2w/SIMD", (double)nb*2*4*iters*4*2048);
  timeit([&]{rate_kernel<4><<<nb*4,256>>>(out,iters);},"mfma f64 4acc 4w/SIMD", (double)nb*4*4*iters*4*2048);
  timeit([&]{rate_kernel<2><<<nb*8,256>>>(out,iters);},"mfma f64 2acc 8w/SIMD", (double)nb*8*4*iters*2*2048);
  timeit([&]{rate_kernel<8><<<nb,256>>>(out,iters*20);},"mfma f64 8acc 1w/SIMD long", (double)nb*4*iters*20.0*8*2048);
  timeit([&]{rate_kernel<4><<<nb*2,256>>>(out,iters*20);},"mfma f64 4acc 2w/SIMD long", (double)nb*2*4*iters*20.0*4*2048);
  timeit([&]{rate_kernel<4><<<1,64>>>(out,iters);},"mfma f64 4acc ONE wave", (double)iters*4*2048);
  timeit([&]{rate_kernel<1><<<1,64>>>(out,iters);},"mfma f64 1acc ONE wave (dep)", (double)iters*1*2048);
  timeit([&]{dfma_kernel<<<1,64>>>(out,iters);},"v_fma_f64 ONE wave", (double)64.0*iters*16*2);
  timeit([&]{dfma_kernel<<<nb*8,256>>>(out,iters*10);},"v_fma_f64 8w/SIMD long", (double)nb*8*256.0*iters*10.0*16*2);
  timeit([&]{dfma_kernel<<<nb*8,256>>>(out,iters);},"v_fma_f64 8w/SIMD", (double)nb*8*256.0*iters*16*2);
  timeit([&]{dfma_kernel<<<nb,256>>>(out,iters);},"v_fma_f64 1w/SIMD", (double)nb*256.0*iters*16*2);
  // HBM copy
  size_t nbytes = (size_t)2<<30; double4 *src,*dst; CK(hipMalloc(&src,nbytes)); CK(hipMalloc(&dst,nbytes));
  CK(hipMemset(src,1,nbytes)); size_t n4=nbytes/sizeof(double4);
  copy_kernel<<<2048,256>>>(src,dst,n4); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for(int r=0;r<5;r++) copy_kernel<<<2048,256>>>(src,dst,n4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  printf("HBM copy: %.1f GB/s (read+write)\n", 5.0*2*nbytes/ms*1e-6);
  return 0;
}
