// ubench_exp.hip -- round 6 experiment behind "k_build_cov >= 3 TB/s": a branch-free exp for bounded arguments (Cody-Waite reduction, Taylor to
// t^13, v_ldexp_f64: 19 instructions) against OCML's exp -- largest difference to the host's exp in ulps over 10^7 uniform arguments in
// [-60, 0] and a sweep down to -800 (underflow), and the rate of both.  MEASURED: <= 1 ulp like OCML's, and NO faster (1282 against 1244 G
// evaluations/s): OCML's exp is already this.  The assembly kernel rewritten around it (two columns per thread, coordinates pre-scaled by
// 1 / l_k: 44 instead of ~84 vector instructions per entry, 16-byte stores) wrote the C4 matrix in 0.213 ms against 0.207: not kept.  PMC
// (tools/build_cov_pmc.sh): VALU issue 63 % busy at 4 cycles per instruction, 46 % of the wave cycles parked.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_exp.hip -o tools/ubench_exp && tools/ubench_exp
#include <hip/hip_runtime.h>
__device__ __forceinline__ double exp_cw13(double x) {
    const double n = rint(x * 1.44269504088896338700e+00);
    double t = __builtin_fma(n, -6.93147180369123816490e-01, x);
    t = __builtin_fma(n, -1.90821492927058770002e-10, t);
    double p = 1.6059043836821613e-10;                    // 1/13!
    p = __builtin_fma(p, t, 2.08767569878681e-09);        // 1/12!
    p = __builtin_fma(p, t, 2.505210838544172e-08);       // 1/11!
    p = __builtin_fma(p, t, 2.755731922398589e-07);       // 1/10!
    p = __builtin_fma(p, t, 2.7557319223985893e-06);      // 1/9!
    p = __builtin_fma(p, t, 2.48015873015873e-05);        // 1/8!
    p = __builtin_fma(p, t, 1.984126984126984e-04);       // 1/7!
    p = __builtin_fma(p, t, 1.388888888888889e-03);       // 1/6!
    p = __builtin_fma(p, t, 8.333333333333333e-03);       // 1/5!
    p = __builtin_fma(p, t, 4.1666666666666664e-02);      // 1/4!
    p = __builtin_fma(p, t, 1.6666666666666666e-01);      // 1/3!
    p = __builtin_fma(p, t, 0.5);
    p = __builtin_fma(p, t, 1.0);
    p = __builtin_fma(p, t, 1.0);
    return ldexp(p, (int)n);
}
namespace bohip {}
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
using namespace bohip;
template <int WHICH>
__global__ void k_exp(const double* __restrict__ x, double* __restrict__ y, size_t n, int reps) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i], acc = 0.0;
    for (int r = 0; r < reps; ++r) acc += WHICH ? exp_cw13(v - 1e-9 * r) : exp(v - 1e-9 * r);
    y[i] = reps == 1 ? (WHICH ? exp_cw13(v) : exp(v)) : acc;
}
static long long ulps(double a, double b) {
    long long ia, ib;
    memcpy(&ia, &a, 8); memcpy(&ib, &b, 8);
    return ia > ib ? ia - ib : ib - ia;
}
int main() {
    const size_t n = 10000000;
    std::vector<double> x(n), y(n), z(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-60.0, 0.0);
    for (size_t i = 0; i < n; ++i) x[i] = i < n - 100000 ? u(g) : -800.0 * (double)(i - (n - 100000)) / 100000.0;
    double *dx, *dy;
    hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    for (int which = 0; which < 2; ++which) {
        if (which) hipLaunchKernelGGL(k_exp<1>, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dy, n, 1);
        else hipLaunchKernelGGL(k_exp<0>, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dy, n, 1);
        hipMemcpy((which ? z : y).data(), dy, n * 8, hipMemcpyDeviceToHost);
    }
    long long worst_cw = 0, worst_ocml = 0; size_t at = 0;
    for (size_t i = 0; i < n; ++i) {
        const double ref = std::exp(x[i]);
        const long long a = ulps(z[i], ref), b = ulps(y[i], ref);
        if (a > worst_cw) { worst_cw = a; at = i; }
        if (b > worst_ocml) worst_ocml = b;
    }
    printf("exp_cw13 vs host exp: max %lld ulp (at x = %.17g: %.17g vs %.17g); OCML exp vs host exp: max %lld ulp; %zu arguments\n", worst_cw, x[at], z[at],
           std::exp(x[at]), worst_ocml, n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        float best = 1e9f;
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0);
            if (which) hipLaunchKernelGGL(k_exp<1>, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dy, n, 64);
            else hipLaunchKernelGGL(k_exp<0>, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dy, n, 64);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("%s: %.3f ms for %zu x 64 evaluations = %.1f G/s\n", which ? "exp_cw13" : "OCML exp", best, n, n * 64.0 / best / 1e6);
    }
    return 0;
}
