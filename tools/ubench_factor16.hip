// ubench_factor16.hip -- the 16 x 16 pivot factorisation of the chain (factor16, kernels_linalg.hip) ALONE on one wave: cycles per block
// and per pivot step, with nothing else on the CU.  In the chain's trace a block takes 3.1 us (~190 ns = ~430 cycles per pivot step).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_factor16.hip -o tools/ubench_factor16 && tools/ubench_factor16
#include "../bayesianoptimization.jl_amd/csrc/kernels_linalg.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace bohip;
template <bool GROW>
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
        if constexpr (GROW) factor16w(a, dl, idl, idl + TILE, 0, lane, info, 0);   // round 6: W16 grown inside
        else factor16(a, dl, idl, 0, lane, info, 0);
        __syncthreads();
        t_sum += clock64() - t0;
        w_sum += wall_clock64() - w0;
    }
    if (lane == 0) { cyc[0] = t_sum; cyc[1] = (long long)w_sum; }
    for (int e = lane; e < 256; e += 64) out[e] = ((e & 15) < (e >> 4)) ? a[(e & 15) * PF_LD + (e >> 4)] : ((e & 15) == (e >> 4) ? dl[e >> 4] : 0.0);
    if constexpr (GROW) for (int e = lane; e < 256; e += 64) out[256 + e] = idl[TILE + e];
}
int main() {
    const int n = 16;
    std::vector<double> A(n * n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double d = (i - j) * 0.3; A[i * n + j] = exp(-0.5 * d * d) + (i == j ? 0.1 : 0); }
    double *dA, *dout; long long* dc; int* info;
    hipMalloc(&dA, 256 * 8); hipMalloc(&dout, 512 * 8); hipMalloc(&dc, 16); hipMalloc(&info, 4); hipMemset(info, 0, 4);
    hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
    const int LDS = POTF2_LDS_BYTES + 256 * 8;
    hipFuncSetAttribute((const void*)k_f16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)k_f16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    const int reps = 2000;
    for (int it = 0; it < 6; ++it) {
        if (it < 3) hipLaunchKernelGGL(k_f16<false>, dim3(1), dim3(64), LDS, 0, dA, dout, dc, reps, info);
        else hipLaunchKernelGGL(k_f16<true>, dim3(1), dim3(64), LDS, 0, dA, dout, dc, reps, info);
        hipDeviceSynchronize();
        long long c[2]; hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
        printf(it < 3 ? "factor16 alone: " : "factor16w (W16 grown inside): ");
        printf(" %.0f shader cycles (s_memtime) = %.2f us (wall clock) per 16 x 16 block; per pivot step %.0f cycles, %.0f ns\n",
               (double)c[0] / reps, (double)c[1] / reps * 0.01, (double)c[0] / reps / 16, (double)c[1] / reps * 10.0 / 16);
    }
    std::vector<double> L(512); hipMemcpy(L.data(), dout, 512 * 8, hipMemcpyDeviceToHost);
    double e1 = 0; for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) { double s = 0; for (int k = 0; k <= j; k++) s += L[i * n + k] * L[j * n + k]; e1 = fmax(e1, fabs(s - A[i * n + j])); }
    printf("max |L L' - A| = %.2e\n", e1);
    double e2 = 0, e3 = 0;   // W16 L = I, zeros above the diagonal
    const double* W = L.data() + 256;
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = 0; for (int k = 0; k < n; k++) s += W[i * n + k] * L[k * n + j]; e2 = fmax(e2, fabs(s - (i == j))); if (j > i) e3 = fmax(e3, fabs(W[i * n + j])); }
    printf("max |W16 L - I| = %.2e, max |W16 above the diagonal| = %.2e\n", e2, e3);
    return 0;
}
