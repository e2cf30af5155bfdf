// Phase-level cycle profile of k_potf2_inv (s_memtime deltas of thread 0).  Build with -DBOHIP_POTF2_CLOCKS.
#define BOHIP_POTF2_CLOCKS 1
#include "../bayesianoptimization.jl_amd/csrc/kernels_linalg.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace bohip;
int main(){
  const int n=128; std::vector<double> A(n*n);
  for(int i=0;i<n;i++)for(int j=0;j<n;j++){ double d=(i-j)*0.05; A[i*n+j]=exp(-0.5*d*d)+(i==j?0.1:0); }
  double *dA,*dW,*dWT; int* info; hipMalloc(&dA,n*n*8); hipMalloc(&dW,n*n*8); hipMalloc(&dWT,n*n*8); hipMalloc(&info,4); hipMemset(info,0,4);
  hipFuncSetAttribute((const void*)k_potf2_inv, hipFuncAttributeMaxDynamicSharedMemorySize, POTF2_LDS_BYTES);
  const char* names[10]={"load","A1 diag16 (x8)","A2 rowsolve (x8)","A3 trailing (x8)","L store","B0 inv16","inv16","inv32","inv64","W,W' store"};
  for(int rep=0;rep<3;rep++){
    { std::vector<double> Al(A); for(int i=0;i<n;i++)for(int j=i+1;j<n;j++)Al[i*n+j]=0; hipMemcpy(dA,Al.data(),n*n*8,hipMemcpyHostToDevice); hipMemset(dW,0,n*n*8); hipMemset(dWT,0,n*n*8); }
    long long z[16]={0}; hipMemcpyToSymbol(HIP_SYMBOL(pf_clocks),z,sizeof(z));
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
    hipLaunchKernelGGL(k_potf2_inv,dim3(1),dim3(PF_THREADS),POTF2_LDS_BYTES,0,dA,(int64_t)n,dW,dWT,(int64_t)n,info,0);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1);
    hipMemcpyFromSymbol(z,HIP_SYMBOL(pf_clocks),sizeof(z));
    long long tot=0; for(int i=0;i<10;i++) tot+=z[i];
    printf("rep %d: %.1f us total, %lld cycles (s_memtime)\n",rep,ms*1e3,tot);
    if(rep==2) { for(int i=0;i<10;i++) printf("  %-18s %8lld cyc  %5.1f%%\n",names[i],z[i],100.0*z[i]/tot);
      printf("  inside A3 (x7): factor16 alone %lld cyc, trailing alone wave1 %lld / wave3 %lld cyc\n", z[10], z[11], z[12]); }
  }
  std::vector<double> L(n*n),W(n*n); hipMemcpy(L.data(),dA,n*n*8,hipMemcpyDeviceToHost); hipMemcpy(W.data(),dW,n*n*8,hipMemcpyDeviceToHost);
  double e1=0,e2=0; for(int i=0;i<n;i++)for(int j=0;j<n;j++){ double s=0,t=0; for(int k=0;k<n;k++){ s+=L[i*n+k]*L[j*n+k]; t+=L[i*n+k]*W[k*n+j]; } e1=fmax(e1,fabs(s-A[i*n+j])); e2=fmax(e2,fabs(t-(i==j))); }
  printf("max|LL'-A| = %.2e   max|LW-I| = %.2e\n",e1,e2);
}
