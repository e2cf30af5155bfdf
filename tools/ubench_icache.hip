// ubench_icache.hip -- what does COLD straight-line code cost on gfx950?  One wave executes NB blocks of 32 dependent-free FP64 FMAs
// (a) unrolled NB times (every instruction fetched once, cold: the instruction cache is invalidated at every dispatch), (b) as a loop
// over ONE block (the same instruction count, 0.5 KB of code).  The difference is instruction fetch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_icache.hip -o tools/ubench_icache && tools/ubench_icache
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define BLOCK(a) asm volatile( \
    "v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n" \
    "v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n" \
    "v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n" \
    "v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n" \
    : "+v"(a) : "v"(b))
template <int NB>
__global__ void k_unrolled(double* out, double b) {
    double a = threadIdx.x;
    unsigned long long t0 = wall_clock64();
#pragma unroll
    for (int i = 0; i < NB; ++i) BLOCK(a);
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = a; ((unsigned long long*)out)[1] = t1 - t0; }
}
__global__ void k_loop(double* out, double b, int nb) {
    double a = threadIdx.x;
    unsigned long long t0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < nb; ++i) BLOCK(a);
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = a; ((unsigned long long*)out)[1] = t1 - t0; }
}
template <int NB>
int run(double* d) {
    unsigned long long h[2];
    double tu = 0, tl = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(k_unrolled<NB>, dim3(1), dim3(64), 0, 0, d, 0.5);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost)); tu = h[1] / 100.0;
        hipLaunchKernelGGL(k_loop, dim3(1), dim3(64), 0, 0, d, 0.5, NB);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost)); tl = h[1] / 100.0;
    }
    printf("%4d blocks = %6.1f KB of code: unrolled (cold) %7.2f us, loop (warm) %7.2f us -> fetch %.0f ns per 64-byte line\n", NB, NB * 16 * 8 / 1024.0, tu, tl,
           (tu - tl) * 1e3 / (NB * 16 * 8 / 64.0));
    return 0;
}
int main() {
    double* d;
    CK(hipMalloc(&d, 64));
    run<8>(d); run<32>(d); run<128>(d); run<512>(d);
    return 0;
}
