// Issue cadence of v_mfma_f64_4x4x4_4b in the register pattern of mma_step<NJ>: acc[8][NJ] += a[i] x b[j],
// with and without the fragment ds_reads interleaved.  s_memtime ticks per MFMA, one..three waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;} }while(0)
template<int NJ, bool LDS>
__global__ __launch_bounds__(256) void k(double* out, long long* clk, int iters, double seed) {
  __shared__ double sm[192*17*2];
  for (int e=threadIdx.x; e<192*17*2; e+=256) sm[e]=seed+e*1e-9;
  __syncthreads();
  double acc[8][NJ]; for(int i=0;i<8;i++)for(int j=0;j<NJ;j++)acc[i][j]=0;
  const int lane=threadIdx.x&63, kq=lane>>4, b=(lane>>2)&3, t=lane&3, w=threadIdx.x>>6;
  const double* ap=sm+((w>>1)*64+4*(b>>1)+t)*17+kq; const double* bp=sm+128*17+((w&1)*8*NJ+4*(b&1)+t)*17+kq;
  double a0[8],b0[NJ];
  for(int i=0;i<8;i++)a0[i]=ap[i*8*17]; for(int j=0;j<NJ;j++)b0[j]=bp[j*8*17];
  long long c0=clock64();
  for(int it=0;it<iters;++it){
    #pragma unroll
    for(int s=0;s<4;++s){
      double a1[8],b1[NJ];
      if (LDS) {
        #pragma unroll
        for(int i=0;i<8;i++)a1[i]=ap[i*8*17+4*((s+1)&3)];
        #pragma unroll
        for(int j=0;j<NJ;j++)b1[j]=bp[j*8*17+4*((s+1)&3)];
        __builtin_amdgcn_sched_barrier(0);
      }
      #pragma unroll
      for(int i=0;i<8;i++)
        #pragma unroll
        for(int j=0;j<NJ;j++) acc[i][j]=__builtin_amdgcn_mfma_f64_4x4x4f64(a0[i],b0[j],acc[i][j],0,0,0);
      __builtin_amdgcn_sched_barrier(0);
      if (LDS) { for(int i=0;i<8;i++)a0[i]=a1[i]; for(int j=0;j<NJ;j++)b0[j]=b1[j]; }
    }
  }
  long long c1=clock64();
  double sacc=0; for(int i=0;i<8;i++)for(int j=0;j<NJ;j++)sacc+=acc[i][j];
  out[blockIdx.x*256+threadIdx.x]=sacc;
  if(blockIdx.x==0&&threadIdx.x==0) clk[0]=c1-c0;
}
template<int NJ,bool LDS> int run(const char* nm,int wps){
  double* out; long long* clk; CK(hipMalloc(&out,256*3*256*8)); CK(hipMalloc(&clk,8));
  int iters=2000, blocks=256*wps;
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NJ,LDS><<<blocks,256>>>(out,clk,iters,1.0); CK(hipDeviceSynchronize());
  hipEventRecord(e0); k<NJ,LDS><<<blocks,256>>>(out,clk,iters,1.0); hipEventRecord(e1); CK(hipEventSynchronize(e1));
  float ms; hipEventElapsedTime(&ms,e0,e1); long long h; CK(hipMemcpy(&h,clk,8,hipMemcpyDeviceToHost));
  double nm_=(double)iters*4*8*NJ;
  printf("%-26s wps=%d: %.1f TF/s, %.2f memtime-ticks/MFMA (per wave), wall %.3f ms\n",nm,wps,blocks*4.0*nm_*512/ms/1e9,(double)h/nm_,ms);
  hipFree(out); hipFree(clk); return 0;
}
int main(){
  for(int w=1;w<=3;w++){ run<4,false>("NJ=4 regs only",w); run<4,true>("NJ=4 + ds_read frags",w); }
  for(int w=1;w<=2;w++){ run<8,false>("NJ=8 regs only",w); run<8,true>("NJ=8 + ds_read frags",w); }
  return 0;
}
