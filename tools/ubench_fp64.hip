// Micro-benchmarks that calibrate the gfx950 ceilings this project is priced against:
//   (1) v_mfma_f64_16x16x4_f64 issue rate  -> FP64 matrix peak
//   (2) v_fma_f64 rate                     -> FP64 vector peak
//   (3) streaming copy / write             -> achievable HBM bandwidth
// Also checks the f64 MFMA fragment layout (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// D col=l&15,row=(l>>4)+4*reg) against a scalar product with an asymmetric B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

template<int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i=0;i<NACC;i++) acc[i]=(d4){0,0,0,0};
  double a=a0+threadIdx.x*1e-9, b=b0;
  for (int it=0; it<iters; ++it) {
#pragma unroll
    for (int i=0;i<NACC;i++) acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
  }
  double s=0; for (int i=0;i<NACC;i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i=0;i<NACC;i++) acc[i]=i;
  double a=a0+threadIdx.x*1e-9, b=b0;
  for (int it=0; it<iters; ++it) {
#pragma unroll
    for (int i=0;i<NACC;i++) acc[i]=__builtin_fma(a,acc[i],b);
  }
  double s=0; for (int i=0;i<NACC;i++) s+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void k_copy(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
  size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x, st=(size_t)gridDim.x*blockDim.x;
  for (; i<n; i+=st) out[i]=in[i];
}
__global__ void k_write(double2* __restrict__ out, size_t n, double v) {
  size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x, st=(size_t)gridDim.x*blockDim.x;
  for (; i<n; i+=st) out[i]=make_double2(v,v);
}
__global__ void k_layout(const double* A, const double* B, double* D) { // A 16x4 row-major, B 4x16 row-major
  int l=threadIdx.x;
  double a=A[(l&15)*4+(l>>4)], b=B[(l>>4)*16+(l&15)];
  d4 c=(d4){0,0,0,0};
  c=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,c,0,0,0);
  for(int r=0;r<4;r++) D[((l>>4)+4*r)*16+(l&15)]=c[r];
}
template<class F> float timeit(F f,int reps){ hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); f(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for(int i=0;i<reps;i++) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); return ms/reps; }
int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  { // layout
    double hA[64],hB[64],hD[256]; for(int i=0;i<64;i++){hA[i]=1+0.37*i; hB[i]=2-0.11*i*i;}
    double *dA,*dB,*dD; CK(hipMalloc(&dA,512)); CK(hipMalloc(&dB,512)); CK(hipMalloc(&dD,2048));
    CK(hipMemcpy(dA,hA,512,hipMemcpyHostToDevice)); CK(hipMemcpy(dB,hB,512,hipMemcpyHostToDevice));
    k_layout<<<1,64>>>(dA,dB,dD); CK(hipMemcpy(hD,dD,2048,hipMemcpyDeviceToHost));
    double maxerr=0; for(int i=0;i<16;i++)for(int j=0;j<16;j++){ double s=0; for(int k=0;k<4;k++) s=fma(hA[i*4+k],hB[k*16+j],s); maxerr=fmax(maxerr,fabs(s-hD[i*16+j])); }
    printf("f64 mfma layout check maxerr=%g (%s)\n",maxerr,maxerr==0?"bit-exact k-ordered fma chain":"differs");
  }
  double* out; CK(hipMalloc(&out, 256*8*256*8*sizeof(double)));
  int iters=20000;
  for (int wpb : {1,2}) { // waves per SIMD via blocks per CU
    int blocks=p.multiProcessorCount*wpb;
    float ms=timeit([&]{ k_mfma<8><<<blocks,256>>>(out,iters,1.0,0.5); },3);
    double fl=(double)blocks*4*iters*8*2048.0;
    printf("mfma_f64_16x16x4 x8acc, %d waves/SIMD: %.1f TFLOP/s (%.2f cyc/instr/SIMD @2.4GHz)\n",wpb,fl/ms/1e9, (ms*1e-3*2.4e9)/((double)iters*8*wpb));
  }
  for (int wpb : {1,2,4}) {
    int blocks=p.multiProcessorCount*wpb;
    float ms=timeit([&]{ k_fma<16><<<blocks,256>>>(out,iters,1.0000001,0.5); },3);
    double fl=(double)blocks*256*(double)iters*16*2.0;
    printf("v_fma_f64 x16acc, %d waves/SIMD: %.1f TFLOP/s\n",wpb,fl/ms/1e9);
  }
  size_t nbytes=(size_t)2<<30; double2 *a,*b; CK(hipMalloc(&a,nbytes)); CK(hipMalloc(&b,nbytes)); CK(hipMemset(a,1,nbytes));
  size_t n=nbytes/sizeof(double2);
  float ms=timeit([&]{ k_copy<<<2048,256>>>(a,b,n); },5);
  printf("copy 2GiB->2GiB: %.2f TB/s (read+write)\n", 2.0*nbytes/ms/1e9);
  ms=timeit([&]{ k_write<<<2048,256>>>(b,n,1.5); },5);
  printf("write 2GiB: %.2f TB/s\n", 1.0*nbytes/ms/1e9);
  return 0;
}
