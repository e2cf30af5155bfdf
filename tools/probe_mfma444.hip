// Empirically derive the lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
__global__ void probe(double* D) { // block (la,lb): A one-hot at la, B one-hot at lb
  int la=blockIdx.x/64, lb=blockIdx.x%64, l=threadIdx.x;
  double a=(l==la)?1.0:0.0, b=(l==lb)?1.0:0.0;
  double d=__builtin_amdgcn_mfma_f64_4x4x4f64(a,b,0.0,0,0,0);
  D[blockIdx.x*64+l]=d;
}
int main(){
  double* dD; CK(hipMalloc(&dD,4096*64*8)); probe<<<4096,64>>>(dD);
  double* h=(double*)malloc(4096*64*8); CK(hipMemcpy(h,dD,4096*64*8,hipMemcpyDeviceToHost));
  for(int la=0;la<64;la++){ printf("A lane %2d pairs with B lanes -> D lane: ",la);
    for(int lb=0;lb<64;lb++) for(int l=0;l<64;l++) if(h[(la*64+lb)*64+l]!=0) printf("(%d->%d) ",lb,l);
    printf("\n"); }
  return 0;
}
