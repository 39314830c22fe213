// Probe 2: in-kernel clock under fp64 MFMA / VALU load, MFMA+VALU co-issue, 4x4x4 f64 MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);exit(1);}}while(0)

// mode bit0: even waves MFMA; mode bit1: odd waves VALU; waves not selected do the other selected thing
__global__ void __launch_bounds__(512) mix_kernel(double* out, unsigned long long* stamps, int iters, int mode){
  int l = threadIdx.x; int wave = __builtin_amdgcn_readfirstlane(threadIdx.x>>6);
  bool do_mfma = (mode==1) || (mode==3 && (wave&1)==0);
  bool do_valu = (mode==2) || (mode==3 && (wave&1)==1);
  double s=0;
  unsigned long long t0=__builtin_amdgcn_s_memtime(), r0=__builtin_amdgcn_s_memrealtime();
  if(do_mfma){
    double a = 1.0 + 1e-9*l, b = 1.0 - 1e-9*l;
    d4 acc[4]; for(int i=0;i<4;i++) acc[i]=(d4){0,0,0,0};
    for(int it=0; it<iters; ++it){
#pragma unroll
      for(int i=0;i<4;i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
    }
    for(int i=0;i<4;i++) s += acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  }
  if(do_valu){
    double x[16]; for(int i=0;i<16;i++) x[i]=1.0+i*1e-3+l*1e-6;
    double a=1.0000001, b=1e-9;
    for(int it=0; it<iters; ++it){
#pragma unroll
      for(int i=0;i<16;i++) x[i] = __builtin_fma(x[i],a,b);
    }
    for(int i=0;i<16;i++) s+=x[i];
  }
  unsigned long long t1=__builtin_amdgcn_s_memtime(), r1=__builtin_amdgcn_s_memrealtime();
  out[blockIdx.x*blockDim.x + l] = s;
  if((l&63)==0){ stamps[(blockIdx.x*8+wave)*2]=t1-t0; stamps[(blockIdx.x*8+wave)*2+1]=r1-r0; }
}

__global__ void __launch_bounds__(256) mfma444_kernel(double* out, int iters){
  int l=threadIdx.x; double a=1.0+1e-9*l,b=1.0-1e-9*l;
  double acc[8]; for(int i=0;i<8;i++)acc[i]=0;
  for(int it=0;it<iters;++it){
#pragma unroll
    for(int i=0;i<8;i++) acc[i]=__builtin_amdgcn_mfma_f64_4x4x4f64(a,b,acc[i],0,0,0);
  }
  double s=0; for(int i=0;i<8;i++) s+=acc[i];
  out[blockIdx.x*blockDim.x+l]=s;
}

int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0)); int nb=p.multiProcessorCount;
  double* out; CK(hipMalloc(&out,(size_t)nb*8*512*8));
  unsigned long long* st; CK(hipMalloc(&st,(size_t)nb*8*8*2*8));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int iters=200000;
  auto run=[&](const char* name,int blocks,int threads,int mode,double mfma_waves,double valu_waves){
    mix_kernel<<<blocks,threads>>>(out,st,iters/10,mode); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); mix_kernel<<<blocks,threads>>>(out,st,iters,mode); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1));
    int nw=blocks*(threads/64); std::vector<unsigned long long> h(blocks*8*2); CK(hipMemcpy(h.data(),st,h.size()*8,hipMemcpyDeviceToHost));
    std::vector<double> clk; for(int b=0;b<blocks;b++)for(int w=0;w<threads/64;w++){double c=(double)h[(b*8+w)*2], r=(double)h[(b*8+w)*2+1]; if(r>0)clk.push_back(c/r*100.0);} std::sort(clk.begin(),clk.end());
    double mf = mfma_waves*iters*4.0*2048, vf = valu_waves*64.0*iters*16*2;
    printf("%-30s %8.2f ms  clk(med)=%7.1f MHz  mfma %6.2f TF  valu %6.2f TF  sum %6.2f\n",name,ms,clk[clk.size()/2],mf/ms*1e-9,vf/ms*1e-9,(mf+vf)/ms*1e-9);
    (void)nw;
  };
  run("MFMA only 1w/SIMD",nb,256,1,nb*4.0,0);
  run("MFMA only 2w/SIMD",nb,512,1,nb*8.0,0);
  run("VALU only 1w/SIMD",nb,256,2,0,nb*4.0);
  run("VALU only 2w/SIMD",nb,512,2,0,nb*8.0);
  run("VALU only 4w/SIMD",nb*2,512,2,0,nb*16.0);
  run("MIX 1 MFMA + 1 VALU /SIMD",nb,512,3,nb*4.0,nb*4.0);
  run("MIX 2 MFMA + 2 VALU /SIMD",nb*2,512,3,nb*8.0,nb*8.0);
  CK(hipEventRecord(e0)); mfma444_kernel<<<nb*2,256>>>(out,iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  printf("mfma_f64_4x4x4 8acc 2w/SIMD: %.2f ms %.2f TF\n",ms,(double)nb*2*4*iters*8.0*512/ms*1e-9);
  return 0;
}
