// Second-round FP64 pipe calibration: in-kernel s_memtime cycles + wall clock, MFMA/VALU mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

// MODE 0: mfma 16x16x4 only; 1: fma only; 2: interleave NM mfma + NF fma per iter; 3: mfma 4x4x4_4b
template<int MODE,int NM,int NF>
__global__ __launch_bounds__(256) void k(double* out, long long* clk, int iters, double a0, double b0) {
  d4 acc[NM>0?NM:1]; double f[NF>0?NF:1];
  for (int i=0;i<(NM>0?NM:1);i++) acc[i]=(d4){0,0,0,0};
  for (int i=0;i<(NF>0?NF:1);i++) f[i]=i;
  double a=a0+threadIdx.x*1e-9, b=b0;
  long long c0=clock64(), w0=wall_clock64();
  for (int it=0; it<iters; ++it) {
    if (MODE==0||MODE==2) {
#pragma unroll
      for (int i=0;i<NM;i++) { acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
        if (MODE==2) {
#pragma unroll
          for (int j=0;j<NF/NM;j++) f[i*(NF/NM)+j]=__builtin_fma(a,f[i*(NF/NM)+j],b);
        } }
    }
    if (MODE==1) {
#pragma unroll
      for (int i=0;i<NF;i++) f[i]=__builtin_fma(a,f[i],b);
    }
    if (MODE==3) {
#pragma unroll
      for (int i=0;i<NM;i++) { double t=__builtin_amdgcn_mfma_f64_4x4x4f64(a,b,acc[i][0],0,0,0); acc[i][0]=t; }
    }
  }
  long long c1=clock64(), w1=wall_clock64();
  double s=0; for (int i=0;i<(NM>0?NM:1);i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  for (int i=0;i<(NF>0?NF:1);i++) s+=f[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
  if (blockIdx.x==0&&threadIdx.x==0){clk[0]=c1-c0; clk[1]=w1-w0;}
}
template<class F> float timeit(F f,int reps){ hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for(int i=0;i<reps;i++) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); return ms/reps; }
double* out; long long* clk;
template<int MODE,int NM,int NF> void run(const char* name,int wps,int threads,int nblk_scale=1){
  int blocks=256*wps*256/threads*nblk_scale; int iters=8000;
  float ms=timeit([&]{ k<MODE,NM,NF><<<blocks,threads>>>(out,clk,iters,1.0000001,0.5); },3);
  long long h[2]; CK(hipMemcpy(h,clk,16,hipMemcpyDeviceToHost));
  double waves=(double)blocks*threads/64;
  double fl_m=(MODE==0||MODE==2)? waves*iters*NM*2048.0 : (MODE==3? waves*iters*NM*2.0*4*4*4*4:0);
  double fl_f=(MODE==1||MODE==2)? waves*64.0*iters*NF*2.0:0;
  printf("%-34s wps=%d thr=%d: %.3f ms  mfma %.1f TF + valu %.1f TF = %.1f TF | memtime-cyc/iter %.1f, wallclk ticks %lld (100MHz => %.3f ms)\n",
    name,wps,threads,ms,fl_m/ms/1e9,fl_f/ms/1e9,(fl_m+fl_f)/ms/1e9,(double)h[0]/iters,h[1],h[1]/1e5);
}
int main(){
  CK(hipMalloc(&out,(size_t)256*8*256*8*8)); CK(hipMalloc(&clk,16));
  run<0,4,0>("mfma16x16x4 4acc",1,256); run<0,8,0>("mfma16x16x4 8acc",1,256); run<0,16,0>("mfma16x16x4 16acc",1,256);
  run<0,8,0>("mfma16x16x4 8acc",2,256); run<0,8,0>("mfma16x16x4 8acc",4,256); run<0,4,0>("mfma16x16x4 4acc",8,256);
  run<0,8,0>("mfma16x16x4 8acc 1wave-blocks",1,64);
  run<3,8,0>("mfma4x4x4_4b 8acc",1,256); run<3,8,0>("mfma4x4x4_4b 8acc",2,256);
  run<1,0,16>("fma 16acc",1,256); run<1,0,16>("fma 16acc",4,256); run<1,0,16>("fma 16acc",8,256);
  run<2,4,16>("mix 4mfma+16fma",1,256); run<2,4,16>("mix 4mfma+16fma",2,256); run<2,4,32>("mix 4mfma+32fma",2,256);
  run<2,4,64>("mix 4mfma+64fma",1,256); run<2,4,64>("mix 4mfma+64fma",2,256);run<2,4,8>("mix 4mfma+8fma",2,256);
  return 0;
}
