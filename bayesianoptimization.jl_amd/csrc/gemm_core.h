// gemm_core.h -- the FP64 contraction engine every dense kernel of libbohip shares.
//
// Why v_mfma_f64_4x4x4_4b_f64 and not v_mfma_f64_16x16x4_f64: measured on MI355X
// (tools/ubench_fp64b.hip, profiles/r01_ubench_fp64.txt) the 16x16x4 form sustains 36 TF/s
// (1 wave/SIMD) to 48 TF/s (>=2 waves/SIMD) while the 4x4x4 four-block form sustains
// 68-73 TF/s, i.e. ~93 % of the 78.6 TF/s FP64 peak.  Lane layout (probed on hardware,
// tools/probe_mfma444.hip):  A[b][i][k] in lane 16k+4b+i,  B[b][k][j] in lane 16k+4b+j,
// D[b][i][j] in lane 16i+4b+j, with 4 independent blocks b.  We arrange the four blocks as a
// 2 x 2 grid (bi = b>>1, bj = b&1) so ONE instruction computes an 8 x 8 x 4 product:
//     rows 4*bi + i, cols 4*bj + j, A replicated over bj, B replicated over bi
// which balances the operand footprint (32 + 32 distinct doubles per instruction).
//
// Workgroup tile 128 rows x 64 columns, 256 threads = 4 waves in 2 x 2, wave tile 64 x 32 = 8 x 4 groups
// -> 32 accumulator doubles per lane.  Both operand tiles are K-major ([rows][16 contraction indices]),
// brought into LDS by global_load_lds (LDS-DMA), three buffers deep.
#pragma once
#ifndef BOHIP_TRACE
#define BOHIP_TRACE 0
#endif
#ifndef BOHIP_ABL
#define BOHIP_ABL 0   // ablation of k_trigemm_sq's loop (tools only, see gemm_tile_loop_glds3_ks)
#endif
#include <type_traits>
#include "common.h"

namespace bohip {

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double mfma444(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// ================================================================================================
// Staging: global_load_lds_dwordx4 (LDS-DMA) writes the tiles straight into LDS, so staging costs no VGPRs,
// no ds_write instructions and no wait inside the MFMA stream.  The DMA destination is wave-uniform base +
// lane*16 B, i.e. rows are exactly 128 B with no padding, so bank conflicts are removed by an XOR swizzle
// of the 16-B segment index with (row & 7), applied on the SOURCE address (which segment a lane fetches)
// and again on the fragment reads:   LDS[row][p] holds global segment p ^ (row & 7).
// Fragment read of (row, col): bank pair = 32*(row&1) + 4*((col>>1) ^ (row&7)) + 2*(col&1) (mod 64) is
// distinct over the 8 rows x 2 columns a 32-lane half touches -> conflict-free.
// ================================================================================================
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;
constexpr int GL_ROW = KC;  // 16 doubles = 128 B, unpadded

template <int NQ>  // NQ * 32 rows
__device__ __forceinline__ void glds_tile(const double* __restrict__ src_lane, int64_t ld, double* dst_wave) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src_lane + (int64_t)q * 32 * ld),
                                         (lds_void_ptr)(dst_wave + q * 32 * GL_ROW), 16, 0, 0);
}

// Fragment reads as ds_read_b128: lane-group kq = lane>>4 reads the 16-B segment 4S + kq of its row, i.e.
// the contraction-index PAIR (8S + 2kq, 8S + 2kq + 1); the first MFMA of a pair consumes the even member,
// the second the odd one (A and B agree, and any permutation of the contraction index is legal).
// b128 moves 256 B/clk against 128 B/clk for the ds_read2_b64 hipcc fuses two b64 reads into, halving the
// LDS time of the fragment traffic that competes with the DMA writes.  Swizzled slot = (8*(row&1) +
// (seg ^ (row&7))) mod 16 is distinct over each 16-lane service group -> conflict-free.
template <int NJ>
__device__ __forceinline__ void load_frags_swz(const double* ap, const double* bp, const int (&oa)[2], const int (&ob)[2],
                                               int S, d2 (&av)[8], d2 (&bv)[NJ]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) av[i] = *reinterpret_cast<const d2*>(ap + i * 8 * GL_ROW + oa[S]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[j] = *reinterpret_cast<const d2*>(bp + j * 8 * GL_ROW + ob[S]);
}
template <int NJ>
__device__ __forceinline__ void mma_pair(const d2 (&av)[8], const d2 (&bv)[NJ], double (&acc)[8][NJ]) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mfma444(av[i].x, bv[j].x, acc[i][j]);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mfma444(av[i].y, bv[j].y, acc[i][j]);
}

// LDS fragment read whose completion is tracked by hand (see gemm_tile_loop_glds3_ks)
template <int OFF>
__device__ __forceinline__ d2 ds_read128(uint32_t lds_byte_addr) {
    d2 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "n"(OFF));
    return r;
}
template <int NJ, int XY = 0>   // XY: 0 both members of the contraction-index pair, 1 the even one only, 2 the odd one only
__device__ __forceinline__ void mma_row(const d2& a, const d2 (&bv)[NJ], double (&acc)[NJ]) {
    if constexpr (XY != 2) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = mfma444(a.x, bv[j].x, acc[j]);
    }
    if constexpr (XY != 1) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = mfma444(a.y, bv[j].y, acc[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int NJ>
__device__ __forceinline__ void mma_chunk_swz(const double* ap, const double* bp, const int (&oa)[2], const int (&ob)[2],
                                              double (&acc)[8][NJ]) {
    d2 a0[8], b0[NJ], a1[8], b1[NJ];
    load_frags_swz<NJ>(ap, bp, oa, ob, 0, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    load_frags_swz<NJ>(ap, bp, oa, ob, 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mma_pair<NJ>(a0, b0, acc);
    __builtin_amdgcn_sched_barrier(0);
    mma_pair<NJ>(a1, b1, acc);
}

// ---- LDS-DMA with THREE buffers: two chunks in flight ------------------------------------------
// With two buffers the DMA of chunk c+1 has exactly one chunk of MFMA time (~1 us) to land, and
// __syncthreads() drains it with vmcnt(0).  Measured on this kernel: removing the staging altogether is
// 15 % faster while removing 4 % of the MFMAs changes nothing -> the loop is bound by load latency x
// bytes in flight per CU, not by the matrix pipe.  Three buffers + a COUNTED s_waitcnt vmcnt(n) + a raw
// s_barrier keep the DMA of chunk c+2 in flight across the barrier (2 chunks of latency tolerance).
template <int NJ>
__device__ __forceinline__ void wait_all_but_one_chunk() {
    static_assert(NJ == 4 || NJ == 6 || NJ == 8, "glds count per chunk = 4 + NJ/2");
    if constexpr (NJ == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (NJ == 6) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    if constexpr (NJ == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
template <int NJ>
constexpr int glds3_lds_bytes() { return 3 * (TILE + 16 * NJ) * GL_ROW * 8; }

template <int NJ>
__device__ __forceinline__ void gemm_tile_loop_glds3(const double* __restrict__ A, int64_t lda,
                                                     const double* __restrict__ B, int64_t ldb, int kc_begin,
                                                     int kc_end, double* smem, double (&acc)[8][NJ],
                                                     int active_rows = TILE) {
    constexpr int BQ = NJ / 2;
    constexpr int AT = TILE * GL_ROW, BT = 16 * NJ * GL_ROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    double* As = smem;             // [3][128][16]
    double* Bs = smem + 3 * AT;    // [3][16*NJ][16]
    if (kc_begin >= kc_end) return;
    const int srow = wave * 8 + (lane >> 3), sseg = (lane & 7) ^ (lane >> 3);
    const double* a_src = A + (int64_t)srow * lda + sseg * 2;
    const double* b_src = B + (int64_t)srow * ldb + sseg * 2;
    double* a_dst = As + wave * 8 * GL_ROW;
    double* b_dst = Bs + wave * 8 * GL_ROW;
    const int k = lane >> 4, b = (lane >> 2) & 3, t = lane & 3;
    const int r7a = 4 * (b >> 1) + t, r7b = 4 * (b & 1) + t;
    int oa[2], ob[2];
#pragma unroll
    for (int S = 0; S < 2; ++S) {
        oa[S] = ((4 * S + k) ^ r7a) << 1;
        ob[S] = ((4 * S + k) ^ r7b) << 1;
    }
    const int a_frag = (wr * 64 + r7a) * GL_ROW, b_frag = (wc * 8 * NJ + r7b) * GL_ROW;
    const bool wave_active = wr * 64 < active_rows;
    glds_tile<4>(a_src + (int64_t)kc_begin * KC, lda, a_dst);
    glds_tile<BQ>(b_src + (int64_t)kc_begin * KC, ldb, b_dst);
    if (kc_begin + 1 < kc_end) {
        glds_tile<4>(a_src + (int64_t)(kc_begin + 1) * KC, lda, a_dst + AT);
        glds_tile<BQ>(b_src + (int64_t)(kc_begin + 1) * KC, ldb, b_dst + BT);
        wait_all_but_one_chunk<NJ>();
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int kc = kc_begin; kc < kc_end; ++kc) {
        const int nxt2 = cur == 0 ? 2 : cur - 1;  // (cur + 2) % 3
        const bool more2 = kc + 2 < kc_end;
        if (more2) {
            glds_tile<4>(a_src + (int64_t)(kc + 2) * KC, lda, a_dst + nxt2 * AT);
            glds_tile<BQ>(b_src + (int64_t)(kc + 2) * KC, ldb, b_dst + nxt2 * BT);
        }
        if (wave_active) mma_chunk_swz<NJ>(As + cur * AT + a_frag, Bs + cur * BT + b_frag, oa, ob, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (more2) wait_all_but_one_chunk<NJ>();           // chunk kc+1 has landed, chunk kc+2 stays in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        cur = cur == 2 ? 0 : cur + 1;
    }
}

// One chunk's fragment reads + MFMAs of one wave of the 8-wave loop (below).  B fragments first, then A row by row, issued as
// inline asm so that the waits are OURS: hipcc answers an LDS read that follows an LDS-DMA with s_waitcnt lgkmcnt(0) (all 12
// reads) before the first MFMA; here row i starts as soon as its own fragment has landed (LDS returns in order): lgkmcnt(7 - i).
// skip: row groups below it are all-zero in a triangular block (wave-uniform).
template <int NJ, int ABL, int XY>
__device__ __forceinline__ void chunk_mma(const double* ap, const double* bp, int skip, double (&acc)[8][NJ]) {
    d2 av[8], bv[NJ];
    const uint32_t pa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const double*)ap;
    const uint32_t pb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const double*)bp;
    if constexpr ((ABL & 2) != 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "=v"(bv[j]) : "v"(pb));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "=v"(av[i]) : "v"(pa));
    } else {
        bv[0] = ds_read128<0>(pb); bv[1] = ds_read128<1024>(pb); bv[2] = ds_read128<2048>(pb); bv[3] = ds_read128<3072>(pb);
        av[0] = ds_read128<0>(pa); av[1] = ds_read128<1024>(pa); av[2] = ds_read128<2048>(pa); av[3] = ds_read128<3072>(pa);
        av[4] = ds_read128<4096>(pa); av[5] = ds_read128<5120>(pa); av[6] = ds_read128<6144>(pa); av[7] = ds_read128<7168>(pa);
    }
    asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(av[0]));
    if (skip <= 0) mma_row<NJ, XY>(av[0], bv, acc[0]);
    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(av[1]));
    if (skip <= 1) mma_row<NJ, XY>(av[1], bv, acc[1]);
    asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(av[2]));
    if (skip <= 2) mma_row<NJ, XY>(av[2], bv, acc[2]);
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(av[3]));
    if (skip <= 3) mma_row<NJ, XY>(av[3], bv, acc[3]);
    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(av[4]));
    if (skip <= 4) mma_row<NJ, XY>(av[4], bv, acc[4]);
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(av[5]));
    if (skip <= 5) mma_row<NJ, XY>(av[5], bv, acc[5]);
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(av[6]));
    if (skip <= 6) mma_row<NJ, XY>(av[6], bv, acc[6]);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[7]));
    if (skip <= 7) mma_row<NJ, XY>(av[7], bv, acc[7]);
}

// ---- 8-wave variant: intra-workgroup split of the contraction index ------------------------------
// Waves 0-3 take the first half (8 of 16 contraction indices) of every chunk, waves 4-7 the second half,
// same 128 x 64 output tile; partial accumulators are added through LDS at the end.  Per SIMD this puts
// 4 waves (2 workgroups x 2) instead of 2 behind the matrix pipe, so a wave stalled in VMEM issue /
// barrier / LDS latency is far more likely to be covered.  Each wave issues 3 of the chunk's 24 DMA pieces.
// Measured on k_trigemm_sq (C2): 0.69 ms vs 0.71 ms for the 4-wave loop; a 16-wave 4-way split drops to one
// workgroup per CU (102 VGPRs x 16 waves) and is slower (0.77 ms).
constexpr int GEMM_THREADS_8 = 512;
#if BOHIP_TRACE
__device__ unsigned long long g_phase[8192 * 8 * 4];   // [block][wave]{issue+mfma, vmcnt wait, barrier wait, iterations}
#endif
// SWAVE: keep the wave index in an SGPR (readfirstlane).  k_trigemm_sq: 128 VGPRs without a spill and scalar branches on the
// wave's role, +2 %.  k_gemm_nt keeps the vector form: with it it needs 129 VGPRs = ONE workgroup per CU, and the launches
// that run beside the factorisation's chain (and the inverse's small levels) are faster that way than with two.
// FOLD = false: leave the two partial accumulator sets as they are (a caller that runs the loop in two pieces -- with a
// wait for the second piece's operands in between, kernels_exec.hip -- folds once, after the last piece)
template <int NJ, int ABL = 0, bool SWAVE = false, bool FOLD = true>  // ABL: ablation switches (tools only): 1 no DMA, 2 no LDS reads, 4 no barriers
__device__ __forceinline__ void gemm_tile_loop_glds3_ks(const double* __restrict__ A, int64_t lda,
                                                        const double* __restrict__ B, int64_t ldb, int kc_begin,
                                                        int kc_end, double* smem, double (&acc)[8][NJ],
                                                        int active_rows = TILE, int tri_kc = 1 << 30, int tid_in = -1) {
    // tri_kc: first chunk of a 128 x 128 block of A that is LOWER-TRIANGULAR (row r has zeros at k > r): from there on
    // an 8-row group whose rows all lie above this wave's 8 contraction indices contributes nothing and is skipped
    // (47 % of that block's MFMAs).
    static_assert(NJ == 4, "piece distribution below assumes 16 + 8 DMA pieces per chunk");
    constexpr int AT = TILE * GL_ROW, BT = 16 * NJ * GL_ROW;
    // tid_in: the caller's copy of threadIdx.x (a persistent kernel passes one the compiler cannot see through, so that
    // nothing lane-dependent is hoisted out of its job loop and kept alive across the whole job)
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wave = SWAVE ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    const int khalf = wave >> 2, w4 = wave & 3, wr = w4 >> 1, wc = w4 & 1;
    double* As = smem;             // [3][128][16]
    double* Bs = smem + 3 * AT;    // [3][64][16]
    if (kc_begin >= kc_end) return;
    // DMA pieces (8 rows x 128 B each): A has 16, B has 8; wave w issues pieces w, w + 8 (A) and 16 + w (B)
    const int prow = lane >> 3, sseg = (lane & 7) ^ (lane >> 3);
    const double* a_src0 = A + (int64_t)(wave * 8 + prow) * lda + sseg * 2;
    const double* a_src1 = A + (int64_t)((wave + 8) * 8 + prow) * lda + sseg * 2;
    const double* b_src = B + (int64_t)(wave * 8 + prow) * ldb + sseg * 2;
    double* a_dst0 = As + wave * 8 * GL_ROW;
    double* a_dst1 = As + (wave + 8) * 8 * GL_ROW;
    double* b_dst = Bs + wave * 8 * GL_ROW;
    auto issue = [&](int kc, int buf) {
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(a_src0 + (int64_t)kc * KC), (lds_void_ptr)(a_dst0 + buf * AT), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(a_src1 + (int64_t)kc * KC), (lds_void_ptr)(a_dst1 + buf * AT), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(b_src + (int64_t)kc * KC), (lds_void_ptr)(b_dst + buf * BT), 16, 0, 0);
    };
    // HALF mode reads rows 0..63 of the A tile only: pieces 8..15 (rows 64..127, a_src1) are not fetched -- two DMA pieces per
    // wave and chunk instead of three (and the rows behind a 64-row half job need not exist)
    auto issue_h = [&](int kc, int buf) {
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(a_src0 + (int64_t)kc * KC), (lds_void_ptr)(a_dst0 + buf * AT), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(b_src + (int64_t)kc * KC), (lds_void_ptr)(b_dst + buf * BT), 16, 0, 0);
    };
    const int k = lane >> 4, b = (lane >> 2) & 3, t = lane & 3;
    const int r7a = 4 * (b >> 1) + t, r7b = 4 * (b & 1) + t;
    const int oa = ((4 * khalf + k) ^ r7a) << 1, ob = ((4 * khalf + k) ^ r7b) << 1;
    // HALF mode (active_rows <= 64, e.g. the last row tile of W at N = 3000: 57 rows): instead of leaving the waves of the lower
    // 64 rows idle for the whole job, both row halves of the wave grid work on rows 0..63 -- wr = 0 takes the even member of every
    // contraction-index pair, wr = 1 the odd one -- and the partial sums are added at the end.  Half the MFMAs per wave and
    // chunk: the job takes about half as long (those jobs are the LONGEST ones, K = N).
    const bool half = active_rows <= TILE / 2;
    const int wre = half ? 0 : wr;
    const int xy = __builtin_amdgcn_readfirstlane(half ? 1 + wr : 0);
    const int a_frag = (wre * 64 + r7a) * GL_ROW + oa, b_frag = (wc * 8 * NJ + r7b) * GL_ROW + ob;
    const bool wave_active = half || wr * 64 < active_rows;
    const int skip0 = __builtin_amdgcn_readfirstlane(khalf - 8 * wre);   // wave-uniform: keep it in an SGPR
    if (half) {
        issue_h(kc_begin, 0);
        if (kc_begin + 1 < kc_end) {
            issue_h(kc_begin + 1, 1);
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        issue(kc_begin, 0);
        if (kc_begin + 1 < kc_end) {
            issue(kc_begin + 1, 1);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __builtin_amdgcn_s_barrier();
    // the chunk loop exists three times (both / even / odd members): a choice INSIDE the loop costs 40 VGPRs of copies
    auto run = [&](auto xy_tag) {
        int cur = 0;
#if BOHIP_TRACE
        unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, phn = 0;
#endif
        for (int kc = kc_begin; kc < kc_end; ++kc) {
#if BOHIP_TRACE
            const unsigned long long tA = __builtin_amdgcn_s_memtime();
#endif
            const int nxt2 = cur == 0 ? 2 : cur - 1;
            const bool more2 = kc + 2 < kc_end;
            constexpr bool HALF = decltype(xy_tag)::value != 0;
            if (more2 && !(ABL & 1)) {
                if constexpr (HALF) issue_h(kc + 2, nxt2);
                else issue(kc + 2, nxt2);
            }
            if (wave_active) {
                const double* ap = As + cur * AT + a_frag;
                const double* bp = Bs + cur * BT + b_frag;
                const int skip = kc >= tri_kc ? 2 * (kc - tri_kc) + skip0 : 0;   // all-zero row groups of a triangular block
                chunk_mma<NJ, ABL, decltype(xy_tag)::value>(ap, bp, skip, acc);
            }
            __builtin_amdgcn_sched_barrier(0);
#if BOHIP_TRACE
            const unsigned long long tB = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (more2) {
                if constexpr (HALF) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if BOHIP_TRACE
            const unsigned long long tC = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
#endif
            if constexpr ((ABL & 4) == 0) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#if BOHIP_TRACE
            const unsigned long long tD = __builtin_amdgcn_s_memtime();
            ph0 += tB - tA; ph1 += tC - tB; ph2 += tD - tC; phn += 1;
#endif
            cur = cur == 2 ? 0 : cur + 1;
        }
#if BOHIP_TRACE
        if (lane == 0 && blockIdx.x < 8192) {
            unsigned long long* o = g_phase + ((size_t)blockIdx.x * 8 + wave) * 4;
            o[0] = ph0; o[1] = ph1; o[2] = ph2; o[3] = phn;
        }
#endif
    };
    if (xy == 0) run(std::integral_constant<int, 0>{});
    else if (xy == 1) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
    if constexpr (!FOLD) return;
    // add the second half's partial accumulators into the first half's (through LDS: 32 doubles per lane)
    double* xch = smem;  // 256 lanes x 32 doubles = 64 KB <= staging area
    if (khalf == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) xch[(i * NJ + j) * 256 + (tid & 255)] = acc[i][j];
    }
    __syncthreads();
    if (khalf == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] += xch[(i * NJ + j) * 256 + tid];
    }
    __syncthreads();
    if (half) {   // rows 0..63: waves 2, 3 (odd members) into waves 0, 1 (even members); the lower row half ends up zero
        if (khalf == 0 && wr == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    xch[(i * NJ + j) * 256 + (tid - 128)] = acc[i][j];
                    acc[i][j] = 0.0;
                }
        }
        __syncthreads();
        if (khalf == 0 && wr == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] += xch[(i * NJ + j) * 256 + tid];
        }
        __syncthreads();
    }
}

// Where this lane's accumulator acc[mi][nj] lives inside the 128 x (16 NJ) workgroup tile.
__device__ __forceinline__ int acc_row(int lane, int wr, int mi) {
    return wr * 64 + 8 * mi + 4 * (((lane >> 2) & 3) >> 1) + (lane >> 4);
}
template <int NJ>
__device__ __forceinline__ int acc_col(int lane, int wc, int nj) {
    return wc * 8 * NJ + 8 * nj + 4 * (((lane >> 2) & 3) & 1) + (lane & 3);
}

}  // namespace bohip
