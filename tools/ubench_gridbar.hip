// Cost of a device-wide barrier inside one persistent kernel on MI355X (8 XCDs, one L2 each), for the single-kernel ascent:
//   variant 0: atomic counter + __threadfence() on both sides (agent-scope release = L2 write-back, acquire = invalidate)
//   variant 2: agent-scope stores + hierarchical counters (one per 32 workgroups, then one global) + ACQUIRE-only fence, plain loads
//   variant 3: as 1 with hierarchical counters
//   variant 1: data exchanged with agent-scope (sc1) stores / loads only; the barrier is s_waitcnt vmcnt(0) + the atomic counter
// Every workgroup writes 1 KB per phase and reads 1 KB written by another workgroup (on another XCD) in the previous phase; the
// result is checked.  usage: ubench_gridbar [workgroups] [phases]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int VAR>
__global__ __launch_bounds__(256) void k_bar(double* buf, unsigned* counter, int phases, unsigned long long* bad, unsigned long long spin_ticks) {
    const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    __shared__ int abort_s;
    if (t == 0) abort_s = 0;
    __syncthreads();
    for (int p = 0; p < phases; ++p) {
        double* mine = buf + ((size_t)(p & 1) * nb + b) * 128;
        if (t < 128) {
            const double v = (double)p * 1000.0 + b + t * 1e-3;
            if (VAR == 0) mine[t] = v;
            else __hip_atomic_store(mine + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (VAR == 0) __threadfence();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            const unsigned long long t0 = wall_clock64();
            if (VAR >= 2) {
                // hierarchical: groups of 32 workgroups count on their own word (64-byte apart); the last of a group counts on the global one
                const int grp = b >> 5, gsz = min(32, nb - (grp << 5)), ngrp = (nb + 31) >> 5;
                const unsigned a = atomicAdd(counter + 16 * (1 + grp), 1u);
                if (a == (unsigned)gsz * (unsigned)(p + 1) - 1u) atomicAdd(counter, 1u);
                const unsigned want = (unsigned)ngrp * (unsigned)(p + 1);
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t0 > spin_ticks) { abort_s = 1; break; }
                }
            } else {
                atomicAdd(counter, 1u);
                const unsigned want = (unsigned)nb * (unsigned)(p + 1);
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t0 > spin_ticks) { abort_s = 1; break; }
                }
            }
            if (VAR == 0) __threadfence();
            if (VAR == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (abort_s) return;
        const int ob = (b + nb / 2 + 1) % nb;   // a workgroup on another XCD
        const double* other = buf + ((size_t)(p & 1) * nb + ob) * 128;
        if (t < 128) {
            const double got = (VAR == 0 || VAR == 2) ? other[t] : __hip_atomic_load(other + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got != (double)p * 1000.0 + ob + t * 1e-3) atomicAdd(bad, 1ull);
        }
    }
}

int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 512, phases = argc > 2 ? atoi(argv[2]) : 2000;
    double* buf; unsigned* cnt; unsigned long long* bad;
    CHECK(hipMalloc(&buf, (size_t)2 * nb * 128 * 8));
    CHECK(hipMalloc(&cnt, 4096)); CHECK(hipMalloc(&bad, 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int var = 0; var < 4; ++var) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipMemset(cnt, 0, 4096)); CHECK(hipMemset(bad, 0, 8)); CHECK(hipMemset(buf, 0, (size_t)2 * nb * 128 * 8));
            CHECK(hipEventRecord(e0));
            if (var == 0) hipLaunchKernelGGL(k_bar<0>, dim3(nb), dim3(256), 0, 0, buf, cnt, phases, bad, 200000000ull);
            else if (var == 1) hipLaunchKernelGGL(k_bar<1>, dim3(nb), dim3(256), 0, 0, buf, cnt, phases, bad, 200000000ull);
            else if (var == 2) hipLaunchKernelGGL(k_bar<2>, dim3(nb), dim3(256), 0, 0, buf, cnt, phases, bad, 200000000ull);
            else hipLaunchKernelGGL(k_bar<3>, dim3(nb), dim3(256), 0, 0, buf, cnt, phases, bad, 200000000ull);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long hb; CHECK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
            if (rep == 1) printf("variant %d (%s), %d workgroups: %.2f us per phase (write 1 KB, barrier, read 1 KB), %llu wrong values\n", var,
                                 var == 0 ? "__threadfence" : var == 1 ? "sc1 accesses" : var == 2 ? "sc1 stores, hierarchical counters, acquire fence + plain loads" : "sc1 accesses, hierarchical counters", nb, ms * 1e3 / phases, hb);
        }
    }
    return 0;
}
