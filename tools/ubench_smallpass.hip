// ubench_smallpass.hip -- round 5: the triangular product of the small-batch path as a 16-right-hand-side MFMA contraction over
// K-MAJOR tiles (out[c][r] = sum_k A[k][c] rhs[k][r], A = W' for V' = K*' W', A = W for U' = V' W), split along the contraction index
// across workgroups with the partial tiles added by the LAST ARRIVER of a column block in a fixed order.
// Question it answers before the library is touched: does a barrier-free stream of plain 16-byte register loads feeding
// v_mfma_f64_16x16x4 reach the plain-read time of W (5.5 us for 36 MB at N = 3000) where the LDS-DMA ring of k_trimv_stream takes 15.6?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_smallpass.hip -o tools/ubench_smallpass && tools/ubench_smallpass [N] [P] [m]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

struct Tile { int cb, kc0, kc1, t0, nseg, pad0, pad1, pad2; };   // column block, contraction chunks [kc0, kc1) of 128, first tile of the column block, its tiles

__device__ __forceinline__ void st_agent2(double* p, double x, double y) {
    d2 v; v.x = x; v.y = y;
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ d2 ld_agent2(const double* p) {
    d2 out;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(out) : "v"(p) : "memory");
    return out;
}

// MODE 0: full (partials + last-arriver combine); 1: no combine (partials written, nobody adds); 2: no partial stores either (one store per lane)
template <int UPPER, int MODE>
__global__ __launch_bounds__(256) void k_tri16(const double* __restrict__ A, int64_t ld, int N, const double* __restrict__ rhs16,
                                               const Tile* __restrict__ tiles, double* __restrict__ part, unsigned* __restrict__ counters,
                                               double* __restrict__ out16) {
    __shared__ double rt[128 * 16];
    __shared__ int is_last;
    const Tile tl = tiles[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, p = lane & 15;
    const int c0 = tl.cb * 128 + wave * 32;
    const double* ap = A + (int64_t)(tl.kc0 * 128 + q) * ld + c0 + 2 * p;
    d2 w[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) w[s] = *(const d2*)(ap + (int64_t)(4 * s) * ld);
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int kc = tl.kc0; kc < tl.kc1; ++kc) {
        if (kc > tl.kc0) __syncthreads();
        {   // right-hand-side tile of this chunk -> LDS: rows k of [k][16], 8 doubles per thread
            const int k = kc * 128 + (tid >> 1);
            const double* src = rhs16 + (int64_t)k * 16 + (tid & 1) * 8;
            d2 v0 = {0, 0}, v1 = {0, 0}, v2 = {0, 0}, v3 = {0, 0};
            if (k < N) { v0 = *(const d2*)src; v1 = *(const d2*)(src + 2); v2 = *(const d2*)(src + 4); v3 = *(const d2*)(src + 6); }
            double* dst = rt + tid * 8;
            *(d2*)dst = v0; *(d2*)(dst + 2) = v1; *(d2*)(dst + 4) = v2; *(d2*)(dst + 6) = v3;
        }
        __syncthreads();
        const bool diag = kc == tl.cb, last_k = kc * 128 + 128 > N, more = kc + 1 < tl.kc1;
        const double* apn = ap + (int64_t)(kc + 1 - tl.kc0) * 128 * ld;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const double b = rt[(4 * s + q) * 16 + p];
            d2 wv = w[s];
            if (more) w[s] = *(const d2*)(apn + (int64_t)(4 * s) * ld);
            if (diag || last_k) {
                const int k = kc * 128 + 4 * s + q, c = c0 + 2 * p;
                const bool okx = k < N && (UPPER ? k >= c : k <= c), oky = k < N && (UPPER ? k >= c + 1 : k <= c + 1);
                if (!okx) wv.x = 0.0;
                if (!oky) wv.y = 0.0;
            }
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wv.x, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wv.y, b, acc1, 0, 0, 0);
        }
    }
    // lane (q, p) holds out[c0 + 2 (q + 4 v) + e][r = p] in acc_e[v].  Pair the lanes p, p ^ 1: the even one keeps column e = 0 with
    // r = p, p + 1, the odd one column e = 1 with r = p - 1, p: 16-byte pieces of the [c][16] layout
    double* pt = part + (int64_t)blockIdx.x * 2048;
    if (MODE == 2) {
        if (acc0[0] + acc1[0] + acc0[1] + acc1[1] + acc0[2] + acc1[2] + acc0[3] + acc1[3] == 12345.678) pt[tid] = 1.0;
        return;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const double mine = (p & 1) ? acc0[v] : acc1[v];          // what the neighbour wants
        const double got = __shfl_xor(mine, 1);
        const int cl = wave * 32 + 2 * (q + 4 * v) + (p & 1);
        double x, y;
        if (p & 1) { x = got; y = acc1[v]; } else { x = acc0[v]; y = got; }
        st_agent2(pt + cl * 16 + (p & ~1), x, y);
    }
    if (MODE == 1) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) is_last = atomicAdd(&counters[tl.cb], 1u) == (unsigned)(tl.nseg - 1);
    __syncthreads();
    if (!is_last) return;
    if (tid == 0) counters[tl.cb] = 0u;
    const double* p0 = part + (int64_t)tl.t0 * 2048;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = 2 * (tid + 256 * i);
        d2 sum = {0.0, 0.0};
        for (int si = 0; si < tl.nseg; ++si) {
            const d2 v = ld_agent2(p0 + (int64_t)si * 2048 + e);
            sum.x += v.x; sum.y += v.y;
        }
        const int c = tl.cb * 128 + (e >> 4);
        if (c < N) *(d2*)(out16 + (int64_t)c * 16 + (e & 15)) = sum;
    }
}

template <int UPPER>
static void make_tiles(int N, int m, std::vector<Tile>& tiles) {
    const int T = (N + 127) / 128;
    tiles.clear();
    // heaviest column blocks first
    for (int o = 0; o < T; ++o) {
        const int cb = UPPER ? o : T - 1 - o;
        const int klo = UPPER ? cb : 0, khi = UPPER ? T : cb + 1;
        const int nseg = (khi - klo + m - 1) / m, t0 = (int)tiles.size();
        for (int si = 0; si < nseg; ++si) {
            Tile t{};
            t.cb = cb; t.kc0 = klo + si * m; t.kc1 = std::min(khi, t.kc0 + m); t.t0 = t0; t.nseg = nseg;
            tiles.push_back(t);
        }
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 3000;
    const int P = argc > 2 ? atoi(argv[2]) : 10;
    const int m = argc > 3 ? atoi(argv[3]) : 1;
    const int64_t ld = (N + 1 + 127) / 128 * 128 + 16;
    std::vector<double> hA((size_t)ld * ld), hR((size_t)ld * 16, 0.0), ref((size_t)ld * 16, 0.0), ref_u((size_t)ld * 16, 0.0);
    srand(1);
    for (auto& x : hA) x = rand() / (double)RAND_MAX - 0.5;      // BOTH triangles non-zero: the kernel must mask
    for (int k = 0; k < N; ++k) for (int r = 0; r < P; ++r) hR[(size_t)k * 16 + r] = rand() / (double)RAND_MAX - 0.5;
    for (int r = 0; r < P; ++r)
        for (int c = 0; c < N; ++c) {
            double s = 0.0, u = 0.0;
            for (int k = 0; k <= c; ++k) s += hA[(size_t)k * ld + c] * hR[(size_t)k * 16 + r];
            for (int k = c; k < N; ++k) u += hA[(size_t)k * ld + c] * hR[(size_t)k * 16 + r];
            ref[(size_t)c * 16 + r] = s; ref_u[(size_t)c * 16 + r] = u;
        }
    std::vector<Tile> tl[2];
    make_tiles<0>(N, m, tl[0]);
    make_tiles<1>(N, m, tl[1]);
    double *dA, *dR, *dO, *dP;
    Tile* dT[2];
    unsigned* dC;
    CK(hipMalloc(&dA, hA.size() * 8)); CK(hipMalloc(&dR, hR.size() * 8)); CK(hipMalloc(&dO, hR.size() * 8));
    CK(hipMalloc(&dP, std::max(tl[0].size(), tl[1].size()) * 2048 * 8));
    CK(hipMalloc(&dC, 4096 * 4)); CK(hipMemset(dC, 0, 4096 * 4));
    for (int u = 0; u < 2; ++u) { CK(hipMalloc(&dT[u], tl[u].size() * sizeof(Tile))); CK(hipMemcpy(dT[u], tl[u].data(), tl[u].size() * sizeof(Tile), hipMemcpyHostToDevice)); }
    CK(hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dR, hR.data(), hR.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> hO(hR.size());
    printf("N = %d, P = %d, m = %d: %zu / %zu workgroups (lower / upper), %.1f MB of A\n", N, P, m, tl[0].size(), tl[1].size(), 8.0 * N * (N + 1) / 2 / 1e6);
    auto run = [&](const char* name, int upper, bool check, auto launch) {
        CK(hipMemset(dO, 0, hR.size() * 8));
        for (int i = 0; i < 20; ++i) launch(upper);
        CK(hipDeviceSynchronize());
        const int reps = 200;
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch(upper);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double err = -1.0;
        if (check) {
            CK(hipMemcpy(hO.data(), dO, hR.size() * 8, hipMemcpyDeviceToHost));
            err = 0.0;
            const std::vector<double>& rf = upper ? ref_u : ref;
            for (int r = 0; r < P; ++r)
                for (int c = 0; c < N; ++c) err = std::max(err, std::fabs(hO[(size_t)c * 16 + r] - rf[(size_t)c * 16 + r]));
        }
        const double bytes = 8.0 * N * (N + 1) / 2;
        printf("%-52s %s  %7.2f us per launch  %5.2f TB/s of A  max err %.1e\n", name, upper ? "upper" : "lower", ms * 1e3 / reps,
               bytes / (ms * 1e-3 / reps) / 1e12, err);
    };
    for (int upper = 0; upper < 2; ++upper) {
        const unsigned nt = (unsigned)tl[upper].size();
        run("mfma 16x16x4 tiles, partials + last-arriver combine", upper, true, [&](int up) {
            if (up) hipLaunchKernelGGL((k_tri16<1, 0>), dim3(nt), dim3(256), 0, 0, dA, ld, N, dR, dT[1], dP, dC, dO);
            else hipLaunchKernelGGL((k_tri16<0, 0>), dim3(nt), dim3(256), 0, 0, dA, ld, N, dR, dT[0], dP, dC, dO);
        });
        run("  ablation: partials stored, nobody combines", upper, false, [&](int up) {
            if (up) hipLaunchKernelGGL((k_tri16<1, 1>), dim3(nt), dim3(256), 0, 0, dA, ld, N, dR, dT[1], dP, dC, dO);
            else hipLaunchKernelGGL((k_tri16<0, 1>), dim3(nt), dim3(256), 0, 0, dA, ld, N, dR, dT[0], dP, dC, dO);
        });
        run("  ablation: stream + MFMA only (no stores)", upper, false, [&](int up) {
            if (up) hipLaunchKernelGGL((k_tri16<1, 2>), dim3(nt), dim3(256), 0, 0, dA, ld, N, dR, dT[1], dP, dC, dO);
            else hipLaunchKernelGGL((k_tri16<0, 2>), dim3(nt), dim3(256), 0, 0, dA, ld, N, dR, dT[0], dP, dC, dO);
        });
    }
    return 0;
}
