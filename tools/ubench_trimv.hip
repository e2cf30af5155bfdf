// ubench_trimv.hip -- the row-wise triangular product of the small-batch path (V'[r][j] = sum_{k<=j} W[j][k] K*'[r][k], and its
// upper twin on W'), alone: the shipped form (one workgroup per 8 rows x 8 right-hand sides, dispatched longest rows first)
// against a BALANCED form (one workgroup per CU, every workgroup walks a list of row blocks of equal total length).
// Hypothesis behind it: a CU pulls ~10 B/clk from HBM/MALL (MI355X_MICROARCH.md), so the product's time is the bytes of the
// MOST LOADED CU over that rate -- with ~1.5 workgroups per CU and the longest rows dispatched together that is 2.7x the mean.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_trimv.hip -o tools/ubench_trimv && tools/ubench_trimv [N] [P]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int RT_ROWS = 8, RT_THREADS = 256;
// ---- the shipped kernel (csrc/kernels_linalg.hip k_rows_trimv), verbatim --------------------------------------------
template <int PV>
__global__ __launch_bounds__(RT_THREADS) void k_rows_trimv(const double* __restrict__ W, int64_t ld, int64_t N0,
                                                    const double* __restrict__ rows, int64_t ldr, int P_total,
                                                    double* __restrict__ out, int64_t ldo, int upper) {
    __shared__ double red[RT_THREADS / 64][RT_ROWS * PV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = (P_total + PV - 1) / PV;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int64_t tiles = (N0 + RT_ROWS - 1) / RT_ROWS;
    const int64_t tsel = (int64_t)(idx / G) * 8 + xcd;
    const int64_t j0 = (upper ? tsel : tiles - 1 - tsel) * RT_ROWS;
    const int r0 = (idx % G) * PV;
    const int P = min(PV, P_total - r0);
    if (j0 < 0 || j0 >= N0 || P <= 0) return;
    rows += (int64_t)r0 * ldr;
    out += (int64_t)r0 * ldo;
    double a[RT_ROWS * PV];
#pragma unroll
    for (int t = 0; t < RT_ROWS * PV; ++t) a[t] = 0.0;
    const int64_t k_lo = upper ? j0 : 0, k_hi = upper ? N0 : min(N0, j0 + RT_ROWS);
    constexpr int UNROLL_K = PV == 1 ? 4 : 2;
#pragma unroll UNROLL_K
    for (int64_t k = k_lo + threadIdx.x; k < k_hi; k += RT_THREADS) {
        double w[RT_ROWS], v[PV];
#pragma unroll
        for (int i = 0; i < RT_ROWS; ++i) {
            const int64_t j = j0 + i;
            const bool valid = j < N0 && (upper ? k >= j : k <= j);
            w[i] = valid ? W[j * ld + k] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < PV; ++r) v[r] = r < P ? rows[r * ldr + k] : 0.0;
#pragma unroll
        for (int i = 0; i < RT_ROWS; ++i)
#pragma unroll
            for (int r = 0; r < PV; ++r) a[PV * i + r] += w[i] * v[r];
    }
    if constexpr (PV == 8) {
#pragma unroll
        for (int o = 32, n = 64; o >= 1; o >>= 1, n >>= 1) {
            const bool up = (lane & o) != 0;
#pragma unroll
            for (int t = 0; t < n / 2; ++t) {
                const double send = up ? a[t] : a[t + n / 2];
                const double keep = up ? a[t + n / 2] : a[t];
                a[t] = keep + __shfl_xor(send, o);
            }
        }
        red[wave][lane] = a[0];
    } else {
#pragma unroll
        for (int t = 0; t < RT_ROWS; ++t) {
            double x = a[t];
            for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
            if (lane == 0) red[wave][t] = x;
        }
    }
    __syncthreads();
    if (threadIdx.x < RT_ROWS * PV) {
        const int i = threadIdx.x / PV, r = threadIdx.x % PV;
        const int t = threadIdx.x;
        const double sum = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        if (r < P && j0 + i < N0) out[(int64_t)r * ldo + j0 + i] = sum;
    }
}

// ---- balanced form -------------------------------------------------------------------------------------------------
// G workgroups of NT threads (one per CU when G = #CUs).  Row blocks of RB rows, sorted by cost (their contraction length); workgroup w
// takes the blocks at sorted positions w, 2G-1-w, 2G+w, 4G-1-w, ... (a snake: every workgroup gets the same number of bytes to a
// few per cent).  Thread t owns the contraction indices {2t, 2t+1} + 2 NT n of every row (16-byte loads); RB x PV running sums
// per thread; reduction: recursive halving inside the wave (RB PV = 64 sums -> lane l keeps sum l), then the waves in index order.
template <int PV, int RB, int NT>
__global__ __launch_bounds__(NT) void k_trimv_bal(const double* __restrict__ W, int64_t ld, int64_t N0,
                                                  const double* __restrict__ rows, int64_t ldr, int P,
                                                  double* __restrict__ out, int64_t ldo, int upper) {
    static_assert(RB * PV == 64, "the halving tree below reduces exactly 64 sums per lane");
    constexpr int NW = NT / 64;
    __shared__ double red[NW][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x, w = blockIdx.x;
    const int64_t nblk = (N0 + RB - 1) / RB;
    for (int q = 0;; ++q) {
        const int64_t pos = (int64_t)q * G + ((q & 1) ? G - 1 - w : w);   // position in the cost-descending order
        if (pos >= nblk) break;
        const int64_t b = upper ? pos : nblk - 1 - pos;                     // lower: the last rows are the longest; upper: the first
        const int64_t j0 = b * RB;
        double a[RB * PV];
#pragma unroll
        for (int t = 0; t < RB * PV; ++t) a[t] = 0.0;
        const int64_t k_lo = upper ? (j0 & ~(int64_t)1) : 0, k_hi = upper ? N0 : min(N0, j0 + RB);
        for (int64_t k = k_lo + 2 * tid; k < k_hi; k += 2 * NT) {
            double2 wv[RB], vv[PV];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int64_t j = j0 + i;
                wv[i] = j < N0 ? *reinterpret_cast<const double2*>(W + j * ld + k) : make_double2(0.0, 0.0);
                // the triangle: lower keeps k <= j, upper keeps k >= j (the other half of the buffer is not zero in general)
                if (upper ? k < j : k > j) wv[i].x = 0.0;
                if (upper ? k + 1 < j : k + 1 > j) wv[i].y = 0.0;
            }
#pragma unroll
            for (int r = 0; r < PV; ++r)
                vv[r] = r < P ? *reinterpret_cast<const double2*>(rows + (int64_t)r * ldr + k) : make_double2(0.0, 0.0);
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int r = 0; r < PV; ++r) {
                    a[PV * i + r] += wv[i].x * vv[r].x;
                    a[PV * i + r] += wv[i].y * vv[r].y;
                }
        }
#pragma unroll
        for (int o = 32, n = 64; o >= 1; o >>= 1, n >>= 1) {
            const bool up = (lane & o) != 0;
#pragma unroll
            for (int t = 0; t < n / 2; ++t) {
                const double send = up ? a[t] : a[t + n / 2];
                const double keep = up ? a[t + n / 2] : a[t];
                a[t] = keep + __shfl_xor(send, o);
            }
        }
        red[wave][lane] = a[0];
        __syncthreads();
        if (tid < 64) {
            double s = red[0][tid];
#pragma unroll
            for (int x = 1; x < NW; ++x) s += red[x][tid];
            const int i = tid / PV, r = tid % PV;
            if (r < P && j0 + i < N0) out[(int64_t)r * ldo + j0 + i] = s;
        }
        __syncthreads();
    }
}


// ---- streaming form: W through an LDS ring by LDS-DMA, D chunks ahead -------------------------------------------------
// Workgroup = 512 threads = 8 waves = 2 contraction halves (kh) x 4 right-hand-side groups (rg) of 4; row blocks of 4 rows,
// contraction chunks of 256 (one ring slot = 4 rows x 2 KiB = 8 KiB = one 1-KiB DMA piece per wave).  A workgroup walks its
// row blocks (snake over the cost-sorted list) chunk by chunk; the DMA of step s + D is issued when step s starts, so
// D x 8 KiB per workgroup are in flight whatever the consumers do.  The right-hand-side values of a step (4 x 16 B per lane)
// are prefetched into registers with the same distance -- every step issues exactly five VMEM instructions per wave, which
// is what makes the counted s_waitcnt vmcnt(5 (D - 1)) exact.
typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;
__device__ __forceinline__ d2 gload128(const double* p) {
    d2 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
template <int OFF>
__device__ __forceinline__ d2 lds_read128(uint32_t a) {
    d2 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF));
    return r;
}
struct Walker {   // the (row block, chunk) steps of one workgroup, in order (all fields wave-uniform)
    int q, c, c_first, c_last, G, w, upper, nblk, N0, j0, RB;
    bool valid;
    __device__ __forceinline__ void load() {
        const int pos = q * G + ((q & 1) ? G - 1 - w : w);
        valid = pos < nblk;
        if (!valid) return;
        const int b = upper ? pos : nblk - 1 - pos;
        j0 = b * RB;
        if (upper) { c_first = j0 >> 8; c_last = (N0 - 1) >> 8; }
        else { c_first = 0; c_last = min(N0 - 1, j0 + RB - 1) >> 8; }
        c = c_first;
    }
    __device__ __forceinline__ void init(int G_, int w_, int upper_, int N0_, int RB_) {
        G = G_; w = w_; upper = upper_; N0 = N0_; RB = RB_; nblk = (N0_ + RB_ - 1) / RB_; q = 0; load();
    }
    __device__ __forceinline__ void advance() { if (!valid) return; if (++c > c_last) { ++q; load(); } }
};
// Workgroup = 512 threads = 8 waves = 2 contraction halves (kh) x 4 right-hand-side groups (rg) of 4; row blocks of 8 rows,
// chunks of 256 contraction indices.  One ring slot = the 8 x 256 tile of W (16 KiB) + the 16 x 256 tile of the right-hand sides
// (32 KiB), all of it brought by LDS-DMA: every step is exactly SIX 1-KiB pieces per wave (2 of W, 4 of its own right-hand
// sides), issued D steps ahead, so s_waitcnt vmcnt(6 (D - 1)) is exact and nothing async ever lands in a register.
template <int D, int ABL = 0>   // ABL (diagnostics, results wrong): 1 no FMAs, 2 no LDS reads, 4 no block-end reduction, 8 no right-hand-side DMA
__global__ __launch_bounds__(512) void k_trimv_dma(const double* __restrict__ W, int64_t ld, int64_t N0_,
                                                   const double* __restrict__ rows, int64_t ldr, int P,
                                                   double* __restrict__ out, int64_t ldo, int upper) {
    constexpr int NS = D + 1, WT = 8 * 256, SLOT = 24 * 256, VM = 6;   // doubles per W tile / per ring slot; VMEM instructions per step and wave
    extern __shared__ __attribute__((aligned(16))) double ring[];   // [NS][8 + 16][256] + red[8][32]
    double* red = ring + NS * SLOT;
    const int N0 = (int)N0_;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave & 1, rg = wave >> 1;
    const int r_base = blockIdx.y * 16 + rg * 4;                                        // this wave's four right-hand sides
    const bool live = r_base < P;            // a wave whose right-hand sides do not exist brings W only (2 pieces per step, not 6)
    Walker pw, cw;
    pw.init(gridDim.x, blockIdx.x, upper, N0, 8);
    cw.init(gridDim.x, blockIdx.x, upper, N0, 8);
    const uint32_t ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)ring;
    const double* rsrc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rsrc[r] = rows + (int64_t)min(r_base + r, max(P - 1, 0)) * ldr + kh * 128 + lane * 2;
    auto issue = [&](const Walker& x, int slot) {
        const int row = min(x.j0 + wave, N0 - 1);                   // W: wave w brings row w of the tile (two 1-KiB halves)
        const double* src = W + (int64_t)row * ld + (x.c << 8) + lane * 2;
        double* dst = ring + slot * SLOT + wave * 256;
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)src, (lds_void_ptr)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src + 128), (lds_void_ptr)(dst + 128), 16, 0, 0);
        if (live && (ABL & 8) == 0) {
            double* rdst = ring + slot * SLOT + WT + (rg * 4) * 256 + kh * 128;   // its own four right-hand sides, its own contraction half
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)(rsrc[r] + (x.c << 8)), (lds_void_ptr)(rdst + r * 256), 16, 0, 0);
        }
    };
    int issued = 0;          // steps issued and not yet consumed
#pragma unroll
    for (int u = 0; u < D; ++u) {
        if (pw.valid) { issue(pw, u); ++issued; }
        pw.advance();
    }
    double acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.0;
    int slot = 0;            // ring slot of the consumer's step
    while (cw.valid) {
        // step s has landed when only the steps issued after it are outstanding: VM instructions each
        const int later = issued - 1;
        if (live && (ABL & 8) == 0) {
            if (later >= D - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM * (D - 1)) : "memory");
            else if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM * 2) : "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM * 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (later >= D - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (D - 1)) : "memory");
            else if (later == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();          // every wave's pieces of step s are in; every wave is done with step s - 1
        const uint32_t a = ring_addr + (uint32_t)(slot * SLOT + kh * 128 + lane * 2) * 8u;
        const uint32_t ar = a + (uint32_t)(WT + rg * 4 * 256) * 8u;
        d2 wr[8], rcur[4];
        if constexpr ((ABL & 2) != 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wr[i] = d2{1.0 + lane, 2.0};
            rcur[0] = rcur[1] = rcur[2] = rcur[3] = d2{0.5, 0.25 * lane};
        } else {
        if (live) { rcur[0] = lds_read128<0>(ar); rcur[1] = lds_read128<2048>(ar); rcur[2] = lds_read128<4096>(ar); rcur[3] = lds_read128<6144>(ar); }
        else { rcur[0] = rcur[1] = rcur[2] = rcur[3] = d2{0.0, 0.0}; }
        wr[0] = lds_read128<0>(a); wr[1] = lds_read128<2048>(a); wr[2] = lds_read128<4096>(a); wr[3] = lds_read128<6144>(a);
        wr[4] = lds_read128<8192>(a); wr[5] = lds_read128<10240>(a); wr[6] = lds_read128<12288>(a); wr[7] = lds_read128<14336>(a);
        }
        const int cj0 = cw.j0, cc = cw.c, c_first = cw.c_first, c_last = cw.c_last;
        // the slot of step s - 1 is free now (everybody passed the barrier): issue step s + D into it
        --issued;
        if (pw.valid) { issue(pw, slot == 0 ? NS - 1 : slot - 1); ++issued; }
        pw.advance();
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wr[0]), "+v"(wr[1]), "+v"(wr[2]), "+v"(wr[3]), "+v"(wr[4]), "+v"(wr[5]), "+v"(wr[6]), "+v"(wr[7]),
                                              "+v"(rcur[0]), "+v"(rcur[1]), "+v"(rcur[2]), "+v"(rcur[3]));
        const int k = (cc << 8) + kh * 128 + lane * 2;
        const bool edge = upper ? (cc == c_first || cc == c_last) : cc == c_last;   // wave-uniform: chunks that need the triangle / N0 mask
        if (edge && live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (k >= N0) rcur[r].x = 0.0;
                if (k + 1 >= N0) rcur[r].y = 0.0;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = cj0 + i;
                const bool okx = k < N0 && (upper ? k >= j : k <= j), oky = k + 1 < N0 && (upper ? k + 1 >= j : k + 1 <= j);
                if (!okx) wr[i].x = 0.0;
                if (!oky) wr[i].y = 0.0;
            }
        }
        if (live && (ABL & 1) == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[i][r] += wr[i].x * rcur[r].x;
                acc[i][r] += wr[i].y * rcur[r].y;
            }
        } else if (live) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][0] += wr[i].x + rcur[i & 3].y;
        }
        slot = slot == NS - 1 ? 0 : slot + 1;
        const bool block_ends = cc == c_last;
        cw.advance();
        if (block_ends && (ABL & 4) != 0) {
            if (tid < 128 && cj0 + (tid & 7) < N0 && (tid >> 3) < P) out[(int64_t)(tid >> 3) * ldo + cj0 + (tid & 7)] = acc[tid & 7][0];
        } else if (block_ends) {
            // 32 sums per lane -> after five halving levels every lane pair holds one; sum id = lane bits 5..1
            double a32[32];
#pragma unroll
            for (int t = 0; t < 32; ++t) a32[t] = acc[t >> 2][t & 3];
#pragma unroll
            for (int o = 32, n = 32; o >= 2; o >>= 1, n >>= 1) {
                const bool up = (lane & o) != 0;
#pragma unroll
                for (int t = 0; t < n / 2; ++t) {
                    const double send = up ? a32[t] : a32[t + n / 2];
                    const double keep = up ? a32[t + n / 2] : a32[t];
                    a32[t] = keep + __shfl_xor(send, o);
                }
            }
            double v = a32[0];
            v += __shfl_xor(v, 1);
            const int id = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
            if ((lane & 1) == 0) red[(kh * 4 + rg) * 32 + id] = v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid < 128) {      // 8 rows x 16 right-hand sides: contraction half 0 + half 1
                const int g = tid >> 5, id2 = tid & 31, i = id2 >> 2, r = id2 & 3;
                const int rr = blockIdx.y * 16 + g * 4 + r;
                const double sum = red[g * 32 + id2] + red[(4 + g) * 32 + id2];
                if (rr < P && cj0 + i < N0) out[(int64_t)rr * ldo + cj0 + i] = sum;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][r] = 0.0;
            // (red is rewritten only at the NEXT block's end, behind at least one step barrier that waves 0-1 reach after their reads)
        }
    }
}


// The same with row blocks of SIXTEEN rows (acc 16 x 4 per lane): one right-hand-side tile serves twice the rows (the rhs pieces are two thirds
// of the LDS-DMA traffic of the 8-row form), half the steps and barriers.  Slot = (16 + 16) x 256 doubles = 64 KiB: ring depth 1.
// Same summation order per (row, right-hand side) as the 8-row form: bit-identical results.
template <int D>
__global__ __launch_bounds__(512) void k_trimv_dma16(const double* __restrict__ W, int64_t ld, int64_t N0_,
                                                     const double* __restrict__ rows, int64_t ldr, int P,
                                                     double* __restrict__ out, int64_t ldo, int upper) {
    constexpr int NS = D + 1, WT = 16 * 256, SLOT = 32 * 256, VM = 8;
    extern __shared__ __attribute__((aligned(16))) double ring[];   // [NS][16 + 16][256] + red[8][64]
    double* red = ring + NS * SLOT;
    const int N0 = (int)N0_;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave & 1, rg = wave >> 1;
    const int r_base = blockIdx.y * 16 + rg * 4;
    const bool live = r_base < P;
    Walker pw, cw;
    pw.init(gridDim.x, blockIdx.x, upper, N0, 16);
    cw.init(gridDim.x, blockIdx.x, upper, N0, 16);
    const uint32_t ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)ring;
    const double* rsrc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rsrc[r] = rows + (int64_t)min(r_base + r, max(P - 1, 0)) * ldr + kh * 128 + lane * 2;
    auto issue = [&](const Walker& x, int slot) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {                               // W: wave w brings rows w and w + 8 of the tile (two 1-KiB halves each)
            const int row = min(x.j0 + wave + 8 * h, N0 - 1);
            const double* src = W + (int64_t)row * ld + (x.c << 8) + lane * 2;
            double* dst = ring + slot * SLOT + (wave + 8 * h) * 256;
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)src, (lds_void_ptr)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src + 128), (lds_void_ptr)(dst + 128), 16, 0, 0);
        }
        if (live) {
            double* rdst = ring + slot * SLOT + WT + (rg * 4) * 256 + kh * 128;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)(rsrc[r] + (x.c << 8)), (lds_void_ptr)(rdst + r * 256), 16, 0, 0);
        }
    };
    int issued = 0;
#pragma unroll
    for (int u = 0; u < D; ++u) {
        if (pw.valid) { issue(pw, u); ++issued; }
        pw.advance();
    }
    double acc[16][4];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.0;
    int slot = 0;
    while (cw.valid) {
        const int later = issued - 1;
        if (live) {
            if (later >= D - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM * (D - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (later >= D - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (D - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const uint32_t a = ring_addr + (uint32_t)(slot * SLOT + kh * 128 + lane * 2) * 8u;
        const uint32_t ar = a + (uint32_t)(WT + rg * 4 * 256) * 8u;
        d2 wr[16], rcur[4];
        if (live) { rcur[0] = lds_read128<0>(ar); rcur[1] = lds_read128<2048>(ar); rcur[2] = lds_read128<4096>(ar); rcur[3] = lds_read128<6144>(ar); }
        else { rcur[0] = rcur[1] = rcur[2] = rcur[3] = d2{0.0, 0.0}; }
        wr[0] = lds_read128<0>(a); wr[1] = lds_read128<2048>(a); wr[2] = lds_read128<4096>(a); wr[3] = lds_read128<6144>(a);
        wr[4] = lds_read128<8192>(a); wr[5] = lds_read128<10240>(a); wr[6] = lds_read128<12288>(a); wr[7] = lds_read128<14336>(a);
        wr[8] = lds_read128<16384>(a); wr[9] = lds_read128<18432>(a); wr[10] = lds_read128<20480>(a); wr[11] = lds_read128<22528>(a);
        wr[12] = lds_read128<24576>(a); wr[13] = lds_read128<26624>(a); wr[14] = lds_read128<28672>(a); wr[15] = lds_read128<30720>(a);
        const int cj0 = cw.j0, cc = cw.c, c_first = cw.c_first, c_last = cw.c_last;
        --issued;
        if (pw.valid) { issue(pw, slot == 0 ? NS - 1 : slot - 1); ++issued; }
        pw.advance();
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wr[0]), "+v"(wr[1]), "+v"(wr[2]), "+v"(wr[3]), "+v"(wr[4]), "+v"(wr[5]), "+v"(wr[6]), "+v"(wr[7]),
                                              "+v"(wr[8]), "+v"(wr[9]), "+v"(wr[10]), "+v"(wr[11]), "+v"(wr[12]), "+v"(wr[13]), "+v"(wr[14]), "+v"(wr[15]),
                                              "+v"(rcur[0]), "+v"(rcur[1]), "+v"(rcur[2]), "+v"(rcur[3]));
        const int k = (cc << 8) + kh * 128 + lane * 2;
        const bool edge = upper ? (cc == c_first || cc == c_last) : cc == c_last;
        if (edge && live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (k >= N0) rcur[r].x = 0.0;
                if (k + 1 >= N0) rcur[r].y = 0.0;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int j = cj0 + i;
                const bool okx = k < N0 && (upper ? k >= j : k <= j), oky = k + 1 < N0 && (upper ? k + 1 >= j : k + 1 <= j);
                if (!okx) wr[i].x = 0.0;
                if (!oky) wr[i].y = 0.0;
            }
        }
        if (live) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[i][r] += wr[i].x * rcur[r].x;
                    acc[i][r] += wr[i].y * rcur[r].y;
                }
        }
        slot = slot == NS - 1 ? 0 : slot + 1;
        const bool block_ends = cc == c_last;
        cw.advance();
        if (block_ends) {
            double a64[64];
#pragma unroll
            for (int t = 0; t < 64; ++t) a64[t] = acc[t >> 2][t & 3];
#pragma unroll
            for (int o = 32, n = 64; o >= 1; o >>= 1, n >>= 1) {
                const bool up = (lane & o) != 0;
#pragma unroll
                for (int t = 0; t < n / 2; ++t) {
                    const double send = up ? a64[t] : a64[t + n / 2];
                    const double keep = up ? a64[t + n / 2] : a64[t];
                    a64[t] = keep + __shfl_xor(send, o);
                }
            }
            const int id = ((lane >> 5) & 1) * 32 + ((lane >> 4) & 1) * 16 + ((lane >> 3) & 1) * 8 + ((lane >> 2) & 1) * 4 + ((lane >> 1) & 1) * 2 + (lane & 1);
            red[(kh * 4 + rg) * 64 + id] = a64[0];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid < 256) {      // 16 rows x 16 right-hand sides: contraction half 0 + half 1
                const int g = tid >> 6, id2 = tid & 63, i = id2 >> 2, r = id2 & 3;
                const int rr = blockIdx.y * 16 + g * 4 + r;
                const double sum = red[g * 64 + id2] + red[(4 + g) * 64 + id2];
                if (rr < P && cj0 + i < N0) out[(int64_t)rr * ldo + cj0 + i] = sum;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][r] = 0.0;
        }
    }
}

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 3000;
    const int P = argc > 2 ? atoi(argv[2]) : 10;
    const int64_t ld = (N + 1 + 127) / 128 * 128 + 16;
    const int PR = (P + 15) / 16 * 16;
    std::vector<double> hW((size_t)ld * ld), hR((size_t)PR * ld), ref((size_t)PR * ld, 0.0), ref_u((size_t)PR * ld, 0.0);
    srand(1);
    for (auto& x : hW) x = rand() / (double)RAND_MAX - 0.5;      // BOTH triangles non-zero: the kernels must mask
    for (auto& x : hR) x = rand() / (double)RAND_MAX - 0.5;
    for (int r = 0; r < P; ++r)
        for (int64_t j = 0; j < N; ++j) {
            double s = 0.0, u = 0.0;
            for (int64_t k = 0; k <= j; ++k) s += hW[j * ld + k] * hR[r * ld + k];
            for (int64_t k = j; k < N; ++k) u += hW[j * ld + k] * hR[r * ld + k];
            ref[r * ld + j] = s; ref_u[r * ld + j] = u;
        }
    double *dW, *dR, *dO;
    CK(hipMalloc(&dW, hW.size() * 8)); CK(hipMalloc(&dR, hR.size() * 8)); CK(hipMalloc(&dO, hR.size() * 8));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dR, hR.data(), hR.size() * 8, hipMemcpyHostToDevice));
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)k_trimv_dma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_trimv_dma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_trimv_dma<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_trimv_dma<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_trimv_dma<2, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_trimv_dma<2, 15>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_trimv_dma<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_trimv_dma<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_trimv_dma16<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<double> hO(hR.size()), hO8(hR.size());
    auto run = [&](const char* name, int upper, auto launch) {
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
        CK(hipMemcpy(hO.data(), dO, hR.size() * 8, hipMemcpyDeviceToHost));
        double err = 0.0;
        const std::vector<double>& rf = upper ? ref_u : ref;
        for (int r = 0; r < P; ++r)
            for (int64_t j = 0; j < N; ++j) err = std::max(err, std::fabs(hO[r * ld + j] - rf[r * ld + j]));
        const double bytes = 8.0 * N * (N + 1) / 2;
        printf("%-44s %s  %7.2f us per launch  %5.2f TB/s of W  max err %.1e\n", name, upper ? "upper" : "lower", ms * 1e3 / reps,
               bytes / (ms * 1e-3 / reps) / 1e12, err);
    };
    for (int upper = 0; upper < 2; ++upper) {
        run("shipped: 8 rows x 8 rhs per workgroup", upper, [&](int up) {
            const int64_t tiles = (N + RT_ROWS - 1) / RT_ROWS;
            if (P == 1) hipLaunchKernelGGL(k_rows_trimv<1>, dim3((unsigned)(8 * ((tiles + 7) / 8))), dim3(RT_THREADS), 0, 0, dW, ld, N, dR, ld, P, dO, ld, up);
            else { const int64_t G = (P + 7) / 8;
                hipLaunchKernelGGL(k_rows_trimv<8>, dim3((unsigned)(8 * ((tiles + 7) / 8) * G)), dim3(RT_THREADS), 0, 0, dW, ld, N, dR, ld, P, dO, ld, up); }
        });
        if (P <= 8) {
            run("balanced: 8 rows x 8 rhs, 512 thr, 1 wg/CU", upper, [&](int up) {
                hipLaunchKernelGGL((k_trimv_bal<8, 8, 512>), dim3(cus), dim3(512), 0, 0, dW, ld, N, dR, ld, P, dO, ld, up); });
            run("balanced: 8 rows x 8 rhs, 256 thr, 2 wg/CU", upper, [&](int up) {
                hipLaunchKernelGGL((k_trimv_bal<8, 8, 256>), dim3(2 * cus), dim3(256), 0, 0, dW, ld, N, dR, ld, P, dO, ld, up); });
        }
        for (int wpc : {1, 2}) {
            char nm[96];
            snprintf(nm, sizeof nm, "LDS-DMA ring D=1, %d wg/CU", wpc);
            run(nm, upper, [&](int up) { hipLaunchKernelGGL((k_trimv_dma<1>), dim3(wpc * cus, (P + 15) / 16), dim3(512), (2 * 6144 + 256) * 8, 0, dW, ld, N, dR, ld, P, dO, ld, up); });
            snprintf(nm, sizeof nm, "LDS-DMA ring D=2, %d wg/CU", wpc);
            run(nm, upper, [&](int up) { hipLaunchKernelGGL((k_trimv_dma<2>), dim3(wpc * cus, (P + 15) / 16), dim3(512), (3 * 6144 + 256) * 8, 0, dW, ld, N, dR, ld, P, dO, ld, up); });
        }
        hO8 = hO;   // (the D = 2 ring's result, one workgroup per CU ... two per CU: the same bits)
        run("LDS-DMA ring D=1, 16-ROW blocks, 1 wg/CU", upper, [&](int up) { hipLaunchKernelGGL((k_trimv_dma16<1>), dim3(cus, (P + 15) / 16), dim3(512), (2 * 8192 + 512) * 8, 0, dW, ld, N, dR, ld, P, dO, ld, up); });
        { size_t diff = 0; for (int r = 0; r < P; ++r) for (int64_t j = 0; j < N; ++j) diff += hO[r * ld + j] != hO8[r * ld + j];
          printf("    16-row form against the 8-row ring: %zu of %lld entries differ in some bit\n", diff, (long long)(P * N)); }
        if (upper == 0) {
            run("  ablation: ring D=2 without the FMAs", upper, [&](int up) { hipLaunchKernelGGL((k_trimv_dma<2, 1>), dim3(cus, (P + 15) / 16), dim3(512), (3 * 6144 + 256) * 8, 0, dW, ld, N, dR, ld, P, dO, ld, up); });
            run("  ablation: ... without FMAs and LDS reads", upper, [&](int up) { hipLaunchKernelGGL((k_trimv_dma<2, 3>), dim3(cus, (P + 15) / 16), dim3(512), (3 * 6144 + 256) * 8, 0, dW, ld, N, dR, ld, P, dO, ld, up); });
            run("  ablation: ... and without the block-end tree", upper, [&](int up) { hipLaunchKernelGGL((k_trimv_dma<2, 7>), dim3(cus, (P + 15) / 16), dim3(512), (3 * 6144 + 256) * 8, 0, dW, ld, N, dR, ld, P, dO, ld, up); });
            run("  ablation: ... and without the rhs DMA", upper, [&](int up) { hipLaunchKernelGGL((k_trimv_dma<2, 15>), dim3(cus, (P + 15) / 16), dim3(512), (3 * 6144 + 256) * 8, 0, dW, ld, N, dR, ld, P, dO, ld, up); });
            run("  ablation: only the block-end tree removed", upper, [&](int up) { hipLaunchKernelGGL((k_trimv_dma<2, 4>), dim3(cus, (P + 15) / 16), dim3(512), (3 * 6144 + 256) * 8, 0, dW, ld, N, dR, ld, P, dO, ld, up); });
            run("  ablation: only the rhs DMA removed", upper, [&](int up) { hipLaunchKernelGGL((k_trimv_dma<2, 8>), dim3(cus, (P + 15) / 16), dim3(512), (3 * 6144 + 256) * 8, 0, dW, ld, N, dR, ld, P, dO, ld, up); });
        }
        run("balanced: 4 rows x 16 rhs, 512 thr, 1 wg/CU", upper, [&](int up) {
            hipLaunchKernelGGL((k_trimv_bal<16, 4, 512>), dim3(cus), dim3(512), 0, 0, dW, ld, N, dR, ld, P, dO, ld, up); });
        run("balanced: 4 rows x 16 rhs, 256 thr, 2 wg/CU", upper, [&](int up) {
            hipLaunchKernelGGL((k_trimv_bal<16, 4, 256>), dim3(2 * cus), dim3(256), 0, 0, dW, ld, N, dR, ld, P, dO, ld, up); });
        run("balanced: 4 rows x 16 rhs, 256 thr, 4 wg/CU", upper, [&](int up) {
            hipLaunchKernelGGL((k_trimv_bal<16, 4, 256>), dim3(4 * cus), dim3(256), 0, 0, dW, ld, N, dR, ld, P, dO, ld, up); });
    }
    return 0;
}
