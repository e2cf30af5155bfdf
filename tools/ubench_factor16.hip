// ubench_factor16.hip -- the 16 x 16 pivot factorisation of the chain (factor16, kernels_linalg.hip) ALONE on one wave: cycles per block
// and per pivot step, with nothing else on the CU.  In the chain's trace a block takes 3.1 us (~190 ns = ~430 cycles per pivot step).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_factor16.hip -o tools/ubench_factor16 && tools/ubench_factor16
#include "../bayesianoptimization.jl_amd/csrc/kernels_linalg.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace bohip;
__global__ __launch_bounds__(64) void k_f16(const double* __restrict__ A, double* __restrict__ out, long long* cyc, int reps, int* info) {
    extern __shared__ double sm[];
    double* a = sm;
    double* dl = sm + TILE * PF_LD;
    double* idl = dl + TILE;
    const int lane = threadIdx.x;
    long long t_sum = 0;
    unsigned long long w_sum = 0;
    for (int r = 0; r < reps; ++r) {
        for (int e = lane; e < 256; e += 64) a[(e >> 4) * PF_LD + (e & 15)] = A[e];
        __syncthreads();
        const long long t0 = clock64();
        const unsigned long long w0 = wall_clock64();
        factor16(a, dl, idl, 0, lane, info, 0);
        __syncthreads();
        t_sum += clock64() - t0;
        w_sum += wall_clock64() - w0;
    }
    if (lane == 0) { cyc[0] = t_sum; cyc[1] = (long long)w_sum; }
    for (int e = lane; e < 256; e += 64) out[e] = ((e & 15) < (e >> 4)) ? a[(e & 15) * PF_LD + (e >> 4)] : ((e & 15) == (e >> 4) ? dl[e >> 4] : 0.0);
}
int main() {
    const int n = 16;
    std::vector<double> A(n * n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double d = (i - j) * 0.3; A[i * n + j] = exp(-0.5 * d * d) + (i == j ? 0.1 : 0); }
    double *dA, *dout; long long* dc; int* info;
    hipMalloc(&dA, 256 * 8); hipMalloc(&dout, 256 * 8); hipMalloc(&dc, 16); hipMalloc(&info, 4); hipMemset(info, 0, 4);
    hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k_f16, hipFuncAttributeMaxDynamicSharedMemorySize, POTF2_LDS_BYTES);
    const int reps = 2000;
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL(k_f16, dim3(1), dim3(64), POTF2_LDS_BYTES, 0, dA, dout, dc, reps, info);
        hipDeviceSynchronize();
        long long c[2]; hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
        printf("factor16 alone: %.0f shader cycles (s_memtime) = %.2f us (wall clock) per 16 x 16 block; per pivot step %.0f cycles, %.0f ns\n",
               (double)c[0] / reps, (double)c[1] / reps * 0.01, (double)c[0] / reps / 16, (double)c[1] / reps * 10.0 / 16);
    }
    std::vector<double> L(256); hipMemcpy(L.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
    double e1 = 0; for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) { double s = 0; for (int k = 0; k <= j; k++) s += L[i * n + k] * L[j * n + k]; e1 = fmax(e1, fabs(s - A[i * n + j])); }
    printf("max |L L' - A| = %.2e\n", e1);
    return 0;
}
