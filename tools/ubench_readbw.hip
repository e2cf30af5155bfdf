// ubench_readbw.hip -- how fast can MI355X READ a buffer of the size of W (36 MB at N = 3000: Infinity-Cache resident across
// repeated launches) or of C4's W (400 MB) with a plain streaming kernel?  The floor under the small-batch triangular products.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_readbw.hip -o tools/ubench_readbw && tools/ubench_readbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int U, typename T>
__global__ __launch_bounds__(256) void k_read(const T* __restrict__ p, size_t n, double* __restrict__ out) {
    // grid-stride, U independent loads in flight per thread
    double s = 0.0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (sizeof(T) == 16) s += v[u].x + v[u].y; else s += v[u];
        }
    }
    for (; i < n; i += stride) { if constexpr (sizeof(T) == 16) s += p[i].x + p[i].y; else s += p[i]; }
    if (s == 1.2345e300) out[0] = s;
}
// contiguous block per workgroup (each CU walks its own slab), 16-byte loads, U in flight
template <int U>
__global__ __launch_bounds__(256) void k_read_slab(const double2* __restrict__ p, size_t n, double* __restrict__ out) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x, b = per * blockIdx.x, e = b + per < n ? b + per : n;
    double s = 0.0;
    size_t i = b + threadIdx.x;
    for (; i + (U - 1) * 256 < e; i += U * 256) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
    }
    for (; i < e; i += 256) s += p[i].x + p[i].y;
    if (s == 1.2345e300) out[0] = s;
}

int main() {
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    double* out;
    CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t mb : {36ul, 72ul, 400ul, 2048ul}) {
        const size_t bytes = mb << 20;
        double* p;
        CK(hipMalloc(&p, bytes));
        CK(hipMemset(p, 0, bytes));
        auto run = [&](const char* name, auto launch) {
            for (int i = 0; i < 5; ++i) launch();
            CK(hipDeviceSynchronize());
            const int reps = mb > 1000 ? 20 : 100;
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%5zu MB  %-46s %8.2f us  %5.2f TB/s\n", mb, name, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12);
        };
        for (int wpc : {2, 4, 8}) {
            char nm[96];
            snprintf(nm, sizeof nm, "grid-stride 8 B x4 in flight, %d wg/CU", wpc);
            run(nm, [&] { hipLaunchKernelGGL((k_read<4, double>), dim3(cus * wpc), dim3(256), 0, 0, p, bytes / 8, out); });
            snprintf(nm, sizeof nm, "grid-stride 16 B x4 in flight, %d wg/CU", wpc);
            run(nm, [&] { hipLaunchKernelGGL((k_read<4, double2>), dim3(cus * wpc), dim3(256), 0, 0, (const double2*)p, bytes / 16, out); });
            snprintf(nm, sizeof nm, "grid-stride 16 B x8 in flight, %d wg/CU", wpc);
            run(nm, [&] { hipLaunchKernelGGL((k_read<8, double2>), dim3(cus * wpc), dim3(256), 0, 0, (const double2*)p, bytes / 16, out); });
            snprintf(nm, sizeof nm, "slab per wg 16 B x8 in flight, %d wg/CU", wpc);
            run(nm, [&] { hipLaunchKernelGGL((k_read_slab<8>), dim3(cus * wpc), dim3(256), 0, 0, (const double2*)p, bytes / 16, out); });
        }
        CK(hipFree(p));
    }
    return 0;
}
