// kernels_linalg.hip -- model-update kernels (rows A1-A3 of SURVEY.md section 8):
//   k_build_cov      SEArd/SEIso/Mat52Ard kernel-matrix assembly, lower 128-tiles (HBM-write bound)
//   k_potf2_inv      128x128 diagonal block: Cholesky + triangular inverse in LDS (latency bound)
//   k_gemm_nt        FP64 MFMA contraction C = alpha*A*B' + beta*C, both operands K-major (panel solve,
//                    trailing updates, recursive triangular inverse W = L^-1, U' = V'W of the gradient path)
//   k_sub_mean, k_rows_trimv, k_mll  -- alpha = W'(W(y - beta)) and the marginal likelihood
// Reference call sites replaced: update!/append!/fit! in src/models/gp.jl:11-18 (GaussianProcesses.jl
// update_cK! + ElasticPDMats Cholesky behind them).
#include "gemm_core.h"
#include <utility>
#ifndef BOHIP_FACTOR16_ROWDPP
#define BOHIP_FACTOR16_ROWDPP 1   // factor16: row J's entries reach the lanes of a 16-row by DPP row_newbcast (1) or ds_swizzle (0)
#endif

namespace bohip {

// ------------------------------------------------------------------------------------------------
// A1: cK = K + (exp(2 logNoise) + eps) I on the lower 128-tiles of an Npad x Npad row-major buffer.
// Rows/cols >= N are identity padding so that the blocked factorisation needs no edge cases.
// X is [N][d] row-major (= Julia's d x N column-major).  64 x 64 outputs per workgroup.
// Written with contraction off so entries are bit-identical to the oracle's up to exp().
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double cov_from_r(int kern, double sigma2, double r) {
    if (kern == KERN_MAT52ARD) {
        const double R = sqrt(r), s = sqrt(5.0) * R;
        return sigma2 * (1.0 + s + 5.0 / 3.0 * r) * exp(-s);
    }
    return sigma2 * exp(-0.5 * r);
}

template <int DT>
__global__ __launch_bounds__(256) void k_build_cov(const double* __restrict__ X, int64_t N, int64_t Npad,
                                                   KernelHyper hp, double noise, double* __restrict__ K,
                                                   int64_t ld, int rows_per_block) {
#pragma clang fp contract(off)
    // thread = column j (its coordinates live in registers); the block walks rows_per_block (<= 32) rows whose
    // coordinates are staged in LDS by one coalesced load: read as wave-uniform scalars straight from memory, hipcc
    // emits one s_load + s_waitcnt lgkmcnt(0) PER DIMENSION inside `if (k < d)` branches and the kernel is bound by
    // that latency chain (1.0 TB/s).  Dimensions d <= k < DT carry zero weight instead of a branch.
    __shared__ double xi_l[32 * DT];
    const int d = hp.d;
    const int64_t j = blockIdx.x * 256 + threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.y * rows_per_block;
    if (blockIdx.x * 256 / TILE > (i0 + rows_per_block - 1) / TILE) return;  // whole block above the diagonal tiles
    for (int t = threadIdx.x; t < rows_per_block * DT; t += 256) {
        const int64_t i = i0 + t / DT;
        const int k = t % DT;
        xi_l[t] = (i < N && k < d) ? X[i * d + k] : 0.0;
    }
    double xj[DT], w[DT];
#pragma unroll
    for (int k = 0; k < DT; ++k) {
        xj[k] = (k < d && j < N) ? X[j * d + k] : 0.0;
        w[k] = k < d ? hp.il2[k] : 0.0;
    }
    __syncthreads();
    for (int c = 0; c < rows_per_block; ++c) {
        const int64_t i = i0 + c;
        if (i >= Npad) break;
        if (j >= Npad || j > i) continue;  // only the lower triangle is ever written (the rest stays zero)
        double v;
        if (i < N && j < N) {
            double r = 0.0;
#pragma unroll
            for (int k = 0; k < DT; ++k) {
                const double t = xi_l[c * DT + k] - xj[k];
                r += w[k] * (t * t);
            }
            v = cov_from_r(hp.kern, hp.sigma2, r);
            if (i == j) v += noise;
        } else {
            v = (i == j) ? 1.0 : 0.0;
        }
        K[i * ld + j] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// A2 (diagonal block): in-LDS Cholesky of one 128 x 128 block followed by its triangular inverse,
// one workgroup.  This kernel sits on the critical path of the blocked factorisation (it cannot
// overlap the trailing update that feeds it), so it is organised to minimise barriers:
//
//  Cholesky, 8 panels of 16 columns (3 barriers per panel):
//    A1  wave 0 factors the 16 x 16 diagonal block wave-synchronously (no workgroup barrier);
//        finished columns go to the MIRROR position (strict upper triangle, a[c][r]) and the
//        diagonal to dl[], so the lower triangle keeps serving as workspace.
//    A2  one thread per row below solves x L16' = a_row (16 steps in registers) -> mirror.
//    A3  rank-16 trailing update with 7 x 7 cyclic register tiles per thread.
//  Inverse by recursive doubling over 16-blocks (4 barriers per level):
//    B0  thread c inverts its column of a 16 x 16 diagonal block in registers.
//    B1  for h = 16, 32, 64:  S = L21 W11,  W21 = -W22 S  with (h/16) x 4 register tiles.
//  W (lower, incl. diagonal) is built in the lower triangle while L stays in the mirror.
// LDS row stride 129 doubles: column walks (stride 129 doubles = 258 dwords = 2 mod 64 banks) are
// conflict-free for 16 consecutive rows.
// info: first failing pivot (1-based global index) if the block is not positive definite.
// ------------------------------------------------------------------------------------------------
constexpr int PF_LD = TILE + 1;
constexpr int POTF2_LDS_BYTES = (TILE * PF_LD + 2 * TILE) * 8;

constexpr int PF_THREADS = 256;  // measured: 512 threads is slower (254k vs 228k cycles per block)

// One level of the in-LDS recursive inverse on the matrix pipe: pairs of H-blocks [W11 0; W21 W22],
//     S = L21 W11 (8x8 output blocks, k >= column block),  W21 = -W22 S (k <= row block).
// Each wave owns whole 8 x 8 output blocks and runs v_mfma_f64_4x4x4 (2 x 2 block arrangement = 8 x 8 x 4 per
// instruction, see gemm_core.h) with operands read straight from the LDS image; the triangular operands are
// masked by index because the other triangle of the image holds L (mirror) / workspace.
//   lane l: kq = l>>4, b = (l>>2)&3, t = l&3;  A element (row 4(b>>1)+t, k kq), B element (k kq, col 4(b&1)+t),
//           D element (row 4(b>>1) + (l>>4), col 4(b&1) + (l&3)).
template <int H>
__device__ __forceinline__ void inv_level(double* a, int tid) {
    constexpr int NB = H / 8;                 // 8x8 blocks per side of an H-block
    // each wave owns 2 block-rows x NB block-columns of ONE pair: 2 NB independent accumulator chains per k-step,
    // 2 + NB operand reads for 2 NB MFMAs.
    const int lane = tid & 63, wave = tid >> 6;
    const int pair = wave / (NB / 2), ib0 = 2 * (wave % (NB / 2));
    const int q0 = 2 * H * pair, q1 = q0 + H;
    const int kq = lane >> 4, bb = (lane >> 2) & 3, t = lane & 3;
    const int ar = 4 * (bb >> 1) + t, bc = 4 * (bb & 1) + t;
    const int dr = 4 * (bb >> 1) + (lane >> 4), dc = 4 * (bb & 1) + (lane & 3);
    double acc[2][NB];
    // ---- S = L21 W11;  L21[i][k] = mirror a[q0+k][q1+i],  W11[k][j] = a[q0+k][q0+j] for k >= j
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[r][j] = 0.0;
#pragma unroll 2
    for (int k0 = 0; k0 < H; k0 += 4) {
        const int k = k0 + kq;
        const double* row = a + (q0 + k) * PF_LD;
        double av[2], wv[NB];
#pragma unroll
        for (int r = 0; r < 2; ++r) av[r] = row[q1 + 8 * (ib0 + r) + ar];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const double w = row[q0 + 8 * j + bc];
            wv[j] = (k >= 8 * j + bc) ? w : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[r][j] = mfma444(av[r], wv[j], acc[r][j]);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < NB; ++j) a[(q1 + 8 * (ib0 + r) + dr) * PF_LD + q0 + 8 * j + dc] = acc[r][j];   // S into the W21 slot
    __syncthreads();
    // ---- W21 = -W22 S;  W22[i][k] = a[q1+i][q1+k] for k <= i,  S[k][j] = a[q1+k][q0+j]
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[r][j] = 0.0;
    for (int k0 = 0; k0 < 8 * ib0 + 16; k0 += 4) {   // W22 lower-triangular: nothing beyond the second block-row's diagonal
        const int k = k0 + kq;
        double wv[2], sv[NB];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = 8 * (ib0 + r) + ar;
            const double w = a[(q1 + i) * PF_LD + q1 + k];
            wv[r] = (k <= i) ? w : 0.0;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) sv[j] = a[(q1 + k) * PF_LD + q0 + 8 * j + bc];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[r][j] = mfma444(wv[r], sv[j], acc[r][j]);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < NB; ++j) a[(q1 + 8 * (ib0 + r) + dr) * PF_LD + q0 + 8 * j + dc] = -acc[r][j];
    __syncthreads();
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
// 1/sqrt(x) to ~1 ulp: hardware v_rsq_f64 seed + two Newton steps (the libm sqrt + divide pair is a
// ~150-cycle dependent chain, and this sits on the critical path of every column).
#ifndef BOHIP_RSQRT_NEWTON2
#define BOHIP_RSQRT_NEWTON2 1   // 1 (default): two Newton steps; 0: one Halley step (round-4 experiment: same speed, different last bits)
#endif
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
#if BOHIP_RSQRT_NEWTON2
    double h = 0.5 * x;
    y = y * (1.5 - h * y * y);
    y = y * (1.5 - h * y * y);
    return y;
#else
    // ONE third-order (Halley) step instead of two Newton steps: v_rsq_f64 is good to ~2^-24, its residual e = 1 - x y^2 cubed is below
    // 2^-70, and the chain  x y -> e -> (t, y e) -> y + y e t  is four dependent operations instead of eight.  This function sits on the
    // pivot chain of every 16 x 16 block (factor16_step: 16 times per block, nothing overlaps it).  MEASURED: refit 1.648 against 1.665 ms
    // at N = 3000, 9.80 against 9.93 at N = 10^4 (profiles/r04_rsqrt_halley_ab.txt): inside the noise -- the depth of this chain is not
    // what a pivot step waits for either.  Not the default (the factor's last bits would change for nothing).
    const double xy = x * y;
    const double e = __builtin_fma(-xy, y, 1.0);
    const double t = __builtin_fma(0.375, e, 0.5);
    return __builtin_fma(y * e, t, y);
#endif
}

// A3 for a trailing size of NB 16-blocks: cyclic NB x NB register tile per thread (lower half only), FP64 VALU.
// (An MFMA formulation was measured slower here: 96k vs 32k cycles per block -- the panels are only 16 deep, so
// each 8x8 block is 4 dependent MFMAs behind ~10 LDS reads, while the register-tiled VALU form reuses every
// operand NB times.)
// (ty, tx): the element of every 16 x 16 sub-block this thread owns.  SKIP00: leave out sub-block (0, 0), the next
// diagonal block, which the look-ahead updates and factors separately.
template <int NB, bool SKIP00, int AMOD = -1, int LD = TILE + 1>   // AMOD >= 0: only the sub-block rows ai with ai % 3 == AMOD;  LD: row stride of the image
__device__ __forceinline__ void trailing_update(double* a, int P, int ty, int tx) {
    const int base = P + 16;
    double acc[NB][NB];
#pragma unroll
    for (int ai = 0; ai < NB; ++ai)
#pragma unroll
        for (int ki = 0; ki < NB; ++ki) acc[ai][ki] = 0.0;
#pragma unroll 4
    for (int c = 0; c < 16; ++c) {
        const double* col = a + (P + c) * LD + base;
        double li[NB], lk[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            li[q] = col[ty + 16 * q];
            lk[q] = col[tx + 16 * q];
        }
#pragma unroll
        for (int ai = 0; ai < NB; ++ai) {
            if (AMOD >= 0 && ai % 3 != AMOD) continue;
#pragma unroll
            for (int ki = 0; ki <= ai; ++ki) acc[ai][ki] += li[ai] * lk[ki];
        }
    }
#pragma unroll
    for (int ai = 0; ai < NB; ++ai)
#pragma unroll
        for (int ki = 0; ki <= ai; ++ki) {
            if (AMOD >= 0 && ai % 3 != AMOD) continue;
            if (SKIP00 && ai == 0 && ki == 0) continue;
            const int i = base + ty + 16 * ai, k = base + tx + 16 * ki;
            if (k <= i) a[i * LD + k] -= acc[ai][ki];
        }
}
template <bool SKIP00, int AMOD, int LD = TILE + 1>
__device__ __forceinline__ void trailing_dispatch(double* a, int P, int nb, int ty, int tx) {
    switch (nb) {  // tile size known at compile time per panel
        case 7: trailing_update<7, SKIP00, AMOD, LD>(a, P, ty, tx); break;
        case 6: trailing_update<6, SKIP00, AMOD, LD>(a, P, ty, tx); break;
        case 5: trailing_update<5, SKIP00, AMOD, LD>(a, P, ty, tx); break;
        case 4: trailing_update<4, SKIP00, AMOD, LD>(a, P, ty, tx); break;
        case 3: trailing_update<3, SKIP00, AMOD, LD>(a, P, ty, tx); break;
        case 2: trailing_update<2, SKIP00, AMOD, LD>(a, P, ty, tx); break;
        case 1: trailing_update<1, SKIP00, AMOD, LD>(a, P, ty, tx); break;
        default: break;
    }
}
// A1: the 16 x 16 diagonal block at (P, P), factored by ONE wave with all 64 lanes: lane (r = lane & 15, q = lane >> 4) holds
// A[r][4q .. 4q+3] of the FULL symmetric block.  Step j needs three things from other lanes, none of them through LDS
// memory and none on the reciprocal-square-root chain (they move the un-scaled entries; the scaling by 1/L_jj follows):
//     a_jj            v_readlane from lane (j, j >> 2)                          (wave-uniform)
//     a[r][j]         ds_bpermute from lane (r, j >> 2): the row's entry in column j
//     a[j][4q + e]    ds_swizzle "lane j of my 16-row": row j's entries in my columns -- by symmetry these ARE the
//                     multipliers l_k of my columns, so no per-column broadcast is needed
// The first version kept a row per lane (16 of 64 lanes busy) and fetched every l_k with its own v_readlane pair:
// 240 pairs per block, 8.7 k cycles per block on the critical path of every panel.  Same arithmetic, same operation order:
// the factor is bit-identical.  Finished columns go to the mirror position, the diagonal to dl / idl.
template <int J>
__device__ __forceinline__ double row_bcast16(double v) {   // value of lane J of this lane's row of 16
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_swizzle(lo, (J << 5) | 0x10);
    hi = __builtin_amdgcn_ds_swizzle(hi, (J << 5) | 0x10);
    return __hiloint2double(hi, lo);
}
template <int J>
__device__ __forceinline__ double row_bcast16_dpp(double v) {   // the same value by DPP row_newbcast: a VALU move, no trip through the LDS crossbar
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + J, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + J, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_fetch(double v, int byte_addr) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_bpermute(byte_addr, lo);
    hi = __builtin_amdgcn_ds_bpermute(byte_addr, hi);
    return __hiloint2double(hi, lo);
}
template <int J>
__device__ __forceinline__ void factor16_step(double (&v)[4], int r, int q, double* dl, double* idl, int P, int lane, int* info,
                                              int row0) {
    constexpr int QJ = J >> 2, EJ = J & 3;
    double ajj = readlane_f64(v[EJ], J + 16 * QJ);
    const double arj = lane_fetch(v[EJ], 4 * (r + 16 * QJ));
    double ajk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ajk[e] = row_bcast16<J>(v[e]);
    if (!(ajj > 0.0)) {
        if (lane == 0) atomicCAS(info, 0, row0 + P + J + 1);
        ajj = 1.0;
    }
    const double inv = fast_rsqrt(ajj);
    const double lr = arj * inv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double lk = ajk[e] * inv;
        if (4 * q + e > J) v[e] -= lr * lk;
    }
    if (q == QJ) v[EJ] = lr;                       // column J is final: L[r][J] for r > J (rows <= J: dead entries)
    if (lane == J + 16 * QJ) { dl[P + J] = ajj * inv; idl[P + J] = inv; }
}
// The same step with nothing but arithmetic and lane exchanges in it: the diagonal and its reciprocal (lane-uniform) stay in registers
// until the block is done, a non-positive pivot is remembered and reported once at the end -- no divergent branch (two per step before:
// `if (lane == ...)` around two LDS writes, `if (!(ajj > 0))` around the atomic) between a pivot and the next.
template <int J>
__device__ __forceinline__ void factor16_step_nb(double (&v)[4], int r, int q, double (&dj)[16], double (&ij)[16], int& bad) {
    constexpr int QJ = J >> 2, EJ = J & 3;
    double ajj = readlane_f64(v[EJ], J + 16 * QJ);
    const double arj = lane_fetch(v[EJ], 4 * (r + 16 * QJ));
    double ajk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ajk[e] = BOHIP_FACTOR16_ROWDPP ? row_bcast16_dpp<J>(v[e]) : row_bcast16<J>(v[e]);
    const bool ok = ajj > 0.0;
    bad = (!ok && bad == 0) ? J + 1 : bad;
    ajj = ok ? ajj : 1.0;
    const double inv = fast_rsqrt(ajj);
    const double lr = arj * inv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double lk = ajk[e] * inv;
        if (4 * q + e > J) v[e] -= lr * lk;   // (forced into a select -- every lane computes, the lanes right of column J keep -- it was slower: 6500 against 5500 cycles per block)
    }
    if (q == QJ) v[EJ] = lr;
    dj[J] = ajj * inv;
    ij[J] = inv;
}
template <int... Js>
__device__ __forceinline__ void factor16_steps_nb(double (&v)[4], int r, int q, double (&dj)[16], double (&ij)[16], int& bad,
                                                  std::integer_sequence<int, Js...>) {
    (factor16_step_nb<Js>(v, r, q, dj, ij, bad), ...);
}
#ifndef BOHIP_FACTOR16_NOBRANCH
#define BOHIP_FACTOR16_NOBRANCH 1
#endif
#ifndef BOHIP_FACTOR16_SWIZZLE
#define BOHIP_FACTOR16_SWIZZLE 1   // 1 (default): the 64-lane form above; 0: the 16-lane DPP form below (round-4 experiment, same speed)
#endif
// Round 4: the same factorisation with ONE ROW PER LANE on 16 lanes and every broadcast a DPP move (row_newbcast: "lane J of my row of
// 16 to the whole row", a VALU instruction, gfx90a+) instead of a trip through the LDS crossbar.  The 64-lane form above needs, per pivot,
// a[r][J] from another 16-lane row (ds_bpermute) and row J's entries in the lane's columns (ds_swizzle): ~120 cycles of LDS-pipe
// latency on the chain  update -> broadcast -> scale -> update, 16 times per block: 3.3 us per 16 x 16 block, a third of every panel of
// the pivot chain (profiles/r03_chol_form1_chain_trace_N3000.txt).  With a whole row in the lane, l_r = a[r][J] / L_JJ needs nothing
// from anybody, and the 15 - J multipliers l_k reach the lanes by DPP (two 32-bit moves each, issued back to back, the one the next
// pivot needs first).  Same operands, same operations, same order per entry: the factor is bit-identical to the 64-lane form's
// (tools/chol_ab.py: identical factor and alpha at N = 500 ... 10^4).  MEASURED: no faster -- 3.24 us per 16 x 16 block in the chain's
// trace either way, refit 1.667 against 1.693 ms at N = 3000 (profiles/r04_factor16_dpp_ab.txt).  The broadcasts were never the chain:
// a pivot step is readlane/DPP -> v_rsq_f64 -> two Newton steps (eight dependent FP64 operations at ~16 cycles each beside the trailing
// update's waves on the same SIMDs) -> scale -> update, ~440 cycles whichever way the multipliers travel.  Kept as the alternative form.
template <int J>
__device__ __forceinline__ double row_bcast_dpp(double v) {   // value of lane J of this lane's row of 16 (DPP row_newbcast)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + J, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + J, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int J>
__device__ __forceinline__ void factor16_row_step(double (&v)[16], int r, double* dl, double* idl, int P, int* info, int row0) {
    double ajj = row_bcast_dpp<J>(v[J]);                 // the pivot: lane J's diagonal entry
    if (!(ajj > 0.0)) {
        if (r == 0) atomicCAS(info, 0, row0 + P + J + 1);
        ajj = 1.0;
    }
    const double inv = fast_rsqrt(ajj);
    const double lr = v[J] * inv;                          // L[r][J] (rows r > J; row J itself: L_JJ)
    // the multipliers of the columns right of J, nearest first (the next pivot needs column J + 1 only)
#define BOHIP_F16_COL(K)                                                     \
    if constexpr (K > J) {                                                   \
        const double lk_ = row_bcast_dpp<K>(lr);                             \
        v[K] -= lr * lk_;                                                    \
    }
    BOHIP_F16_COL(1) BOHIP_F16_COL(2) BOHIP_F16_COL(3) BOHIP_F16_COL(4) BOHIP_F16_COL(5) BOHIP_F16_COL(6) BOHIP_F16_COL(7) BOHIP_F16_COL(8)
    BOHIP_F16_COL(9) BOHIP_F16_COL(10) BOHIP_F16_COL(11) BOHIP_F16_COL(12) BOHIP_F16_COL(13) BOHIP_F16_COL(14) BOHIP_F16_COL(15)
#undef BOHIP_F16_COL
    v[J] = lr;                                             // column J is final
    if (r == J) { dl[P + J] = ajj * inv; idl[P + J] = inv; }
}
__device__ __forceinline__ void factor16(double* a, double* dl, double* idl, int P, int lane, int* info, int row0) {
#if BOHIP_FACTOR16_SWIZZLE
    const int r = lane & 15, q = lane >> 4;
    double v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 4 * q + e;
        v[e] = (k <= r) ? a[(P + r) * PF_LD + P + k] : a[(P + k) * PF_LD + P + r];   // lower triangle, mirrored into the upper
    }
#if BOHIP_FACTOR16_NOBRANCH
    {
        double dj[16], ij[16];
        int bad = 0;
        factor16_steps_nb(v, r, q, dj, ij, bad, std::make_integer_sequence<int, 16>{});
        double dmine = 0.0, imine = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            dmine = lane == j ? dj[j] : dmine;
            imine = lane == j ? ij[j] : imine;
        }
        if (lane < 16) { dl[P + lane] = dmine; idl[P + lane] = imine; }
        if (bad != 0 && lane == 0) atomicCAS(info, 0, row0 + P + bad);
    }
#else
    factor16_step<0>(v, r, q, dl, idl, P, lane, info, row0);   factor16_step<1>(v, r, q, dl, idl, P, lane, info, row0);
    factor16_step<2>(v, r, q, dl, idl, P, lane, info, row0);   factor16_step<3>(v, r, q, dl, idl, P, lane, info, row0);
    factor16_step<4>(v, r, q, dl, idl, P, lane, info, row0);   factor16_step<5>(v, r, q, dl, idl, P, lane, info, row0);
    factor16_step<6>(v, r, q, dl, idl, P, lane, info, row0);   factor16_step<7>(v, r, q, dl, idl, P, lane, info, row0);
    factor16_step<8>(v, r, q, dl, idl, P, lane, info, row0);   factor16_step<9>(v, r, q, dl, idl, P, lane, info, row0);
    factor16_step<10>(v, r, q, dl, idl, P, lane, info, row0);  factor16_step<11>(v, r, q, dl, idl, P, lane, info, row0);
    factor16_step<12>(v, r, q, dl, idl, P, lane, info, row0);  factor16_step<13>(v, r, q, dl, idl, P, lane, info, row0);
    factor16_step<14>(v, r, q, dl, idl, P, lane, info, row0);  factor16_step<15>(v, r, q, dl, idl, P, lane, info, row0);
#endif
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 4 * q + e;
        if (k < r) a[(P + k) * PF_LD + P + r] = v[e];   // mirror
    }
#else
    if (lane >= 16) return;                                // (the caller's wave continues after the call: no barrier inside)
    const int r = lane;
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (k <= r) ? a[(P + r) * PF_LD + P + k] : 0.0;   // row r of the lower triangle
    factor16_row_step<0>(v, r, dl, idl, P, info, row0);   factor16_row_step<1>(v, r, dl, idl, P, info, row0);
    factor16_row_step<2>(v, r, dl, idl, P, info, row0);   factor16_row_step<3>(v, r, dl, idl, P, info, row0);
    factor16_row_step<4>(v, r, dl, idl, P, info, row0);   factor16_row_step<5>(v, r, dl, idl, P, info, row0);
    factor16_row_step<6>(v, r, dl, idl, P, info, row0);   factor16_row_step<7>(v, r, dl, idl, P, info, row0);
    factor16_row_step<8>(v, r, dl, idl, P, info, row0);   factor16_row_step<9>(v, r, dl, idl, P, info, row0);
    factor16_row_step<10>(v, r, dl, idl, P, info, row0);  factor16_row_step<11>(v, r, dl, idl, P, info, row0);
    factor16_row_step<12>(v, r, dl, idl, P, info, row0);  factor16_row_step<13>(v, r, dl, idl, P, info, row0);
    factor16_row_step<14>(v, r, dl, idl, P, info, row0);  factor16_row_step<15>(v, r, dl, idl, P, info, row0);
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < r) a[(P + k) * PF_LD + P + r] = v[k];     // finished columns to the mirror position
#endif
}

// Round 6: the same 16 x 16 factorisation with W16 = L16^-1 GROWN INSIDE it (the chain's pivot_block, kernels_chol.hip).  Until now the
// inverse of a pivot block was a second 16-step chain on another wave (group_fsub_w16, ~2 us) beside a 16-step substitution per row
// below the block (~1.6 us), both BEHIND factor16 on the critical path of every panel.  The two triangles of the block are complementary:
// at step J the Schur complement lives in the columns > J of the rows > J, and the columns <= J of those rows are free -- exactly where
// the partial sums of the inverse live,
//     T[i][c] = delta_ic - sum_{k < J} L[i][k] W[k][c]      (rows i >= J, columns c < J),      W[J][c] = T[J][c] / L_JJ,
// and their update at step J,  T[i][c] -= L[i][J] W[J][c],  is the SAME rank-1 formula  M[i][c] -= l_i (M[J][c] / L_JJ)  the Schur
// complement gets (row J's entries in my columns, scaled, times my row's multiplier): the step's four fused multiply-adds per lane
// simply lose their column predicate.  What a step adds: the rows <= J stop (their multiplier is zero), column J starts its life as
// T[i][J] = -l_i / L_JJ, and the finished column of L moves to a register of its own.  Row r keeps T[r][:] unscaled until the end (nobody
// reads a finished row again), so the scaling by 1 / L_rr is four multiplications after the last step.
// Out: L (strict lower) to the mirror position of the image, diagonal to dl, reciprocals to idl, W16 (lower incl. diagonal, ZEROS above:
// every consumer sums all 16 terms) to w16out[16][WLS].
template <int J>
__device__ __forceinline__ void factor16w_step(double (&v)[4], double (&lc)[4], int r, int q, double (&dj)[16], double (&ij)[16], int& bad) {
    constexpr int QJ = J >> 2, EJ = J & 3;
    double ajj = readlane_f64(v[EJ], J + 16 * QJ);
    double arj = lane_fetch(v[EJ], 4 * (r + 16 * QJ));
    arj = (r > J) ? arj : 0.0;                       // rows <= J are final: their multiplier is zero
    double ajk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ajk[e] = row_bcast16_dpp<J>(v[e]);
    const bool ok = ajj > 0.0;
    bad = (!ok && bad == 0) ? J + 1 : bad;
    ajj = ok ? ajj : 1.0;
    const double inv = fast_rsqrt(ajj);
    const double lr = arj * inv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double lk = ajk[e] * inv;              // columns > J: L[c][J];  columns < J: W[J][c]
        v[e] -= lr * lk;
    }
    if (q == QJ) {
        lc[EJ] = lr;                                 // column J of L (rows > J; zero above)
        v[EJ] = (r > J) ? -(lr * inv) : v[EJ];       // T[r][J] = -L[r][J] W[J][J]
    }
    dj[J] = ajj * inv;
    ij[J] = inv;
}
template <int... Js>
__device__ __forceinline__ void factor16w_steps(double (&v)[4], double (&lc)[4], int r, int q, double (&dj)[16], double (&ij)[16], int& bad,
                                                std::integer_sequence<int, Js...>) {
    (factor16w_step<Js>(v, lc, r, q, dj, ij, bad), ...);
}
template <int LD, int WLS>   // LD: row stride of the image;  WLS: row stride of w16out
__device__ __forceinline__ void factor16w(double* a, double* dl, double* idl, double* w16out, int P, int lane, int* info, int row0) {
    const int r = lane & 15, q = lane >> 4;
    double v[4], lc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 4 * q + e;
        v[e] = (k <= r) ? a[(P + r) * LD + P + k] : a[(P + k) * LD + P + r];   // lower triangle, mirrored into the upper
    }
    double dj[16], ij[16];
    int bad = 0;
    factor16w_steps(v, lc, r, q, dj, ij, bad, std::make_integer_sequence<int, 16>{});
    double dmine = 0.0, imine = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        dmine = r == j ? dj[j] : dmine;
        imine = r == j ? ij[j] : imine;
    }
    if (lane < 16) { dl[P + lane] = dmine; idl[P + lane] = imine; }
    if (bad != 0 && lane == 0) atomicCAS(info, 0, row0 + P + bad);
    d2 w01, w23;
    {
        double w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 4 * q + e;
            if (k < r) a[(P + k) * LD + P + r] = lc[e];   // mirror
            w[e] = k < r ? v[e] * imine : (k == r ? imine : 0.0);
        }
        w01.x = w[0]; w01.y = w[1]; w23.x = w[2]; w23.y = w[3];
    }
    *reinterpret_cast<d2*>(w16out + r * WLS + 4 * q) = w01;
    *reinterpret_cast<d2*>(w16out + r * WLS + 4 * q + 2) = w23;
}

// B0 + B1 of the header above on an image whose strict upper triangle (mirror) holds L and whose idl[] holds 1 / L_ii:
// W = L^-1 is built in the lower triangle and stored to Wblk (lower) and WTblk (upper).  Shared by k_potf2_inv and k_inv128.
template <bool SC1 = false>   // SC1: the result is read by kernels that are already running (agent-scope stores)
__device__ __forceinline__ void inverse_phase(double* a, const double* idl, int tid, double* __restrict__ Wblk,
                                              double* __restrict__ WTblk, int64_t ldw) {
    if (tid < TILE) {
        const int q = tid & ~15, cc = tid & 15;
        double w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k) sacc += a[(q + k) * PF_LD + q + i] * w[k];  // w[k] = 0 for k < cc
            w[i] = (i < cc) ? 0.0 : (i == cc ? idl[q + i] : -sacc * idl[q + i]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i >= cc) a[(q + i) * PF_LD + q + cc] = w[i];
    }
    __syncthreads();
    inv_level<16>(a, tid);
    inv_level<32>(a, tid);
    inv_level<64>(a, tid);
    for (int e = tid; e < TILE * TILE / 2; e += PF_THREADS) {  // W (lower) and W' (upper), 16-B stores
        const int i = e >> 6, c = (e & 63) * 2;
        if (c <= i) {
            d2 v;
            v.x = a[i * PF_LD + c];
            v.y = (c + 1 <= i) ? a[i * PF_LD + c + 1] : 0.0;
            if constexpr (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(Wblk + (int64_t)i * ldw + c), "v"(v) : "memory");
            else *reinterpret_cast<d2*>(Wblk + (int64_t)i * ldw + c) = v;
        }
        // W'[r][q] = W[q][r] for q >= r: row r = i, columns q = c, c + 1
        if (c + 1 >= i) {
            d2 v;
            v.x = (c >= i) ? a[c * PF_LD + i] : 0.0;
            v.y = a[(c + 1) * PF_LD + i];
            if constexpr (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(WTblk + (int64_t)i * ldw + c), "v"(v) : "memory");
            else *reinterpret_cast<d2*>(WTblk + (int64_t)i * ldw + c) = v;
        }
    }
}

#ifdef BOHIP_POTF2_CLOCKS
#define PF_CLK(slot) do { if (threadIdx.x == 0) pf_clocks[slot] += clock64() - pf_t0; pf_t0 = clock64(); } while (0)
__device__ long long pf_clocks[16];
#else
#define PF_CLK(slot) do {} while (0)
#endif

__global__ __launch_bounds__(PF_THREADS) void k_potf2_inv(double* __restrict__ Lblk, int64_t ld, double* __restrict__ Wblk,
                                                   double* __restrict__ WTblk, int64_t ldw, int* __restrict__ info,
                                                   int row0) {
    extern __shared__ double sm[];
#ifdef BOHIP_POTF2_CLOCKS
    long long pf_t0 = clock64();
#endif
    double* a = sm;
    double* dl = sm + TILE * PF_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* idl = sm + TILE * PF_LD + TILE;  // 1/L_ii (the hot loops multiply, never divide)
    {   // 16 independent loads in flight per thread; each half-wave reads one full 128-B... row segment
        const int j = tid & 127, ih = tid >> 7;  // 2 row phases
#pragma unroll 1
        for (int i0 = 0; i0 < TILE; i0 += 32) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = Lblk[(int64_t)(i0 + 2 * u + ih) * ld + j];
#pragma unroll
            for (int u = 0; u < 16; ++u) a[(i0 + 2 * u + ih) * PF_LD + j] = (j <= i0 + 2 * u + ih) ? v[u] : 0.0;
        }
    }
    __syncthreads();
    PF_CLK(0);
    // Look-ahead over the 8 panels of 16 columns: the pivot block of panel j+1 (wave 0, a serial chain of 16 pivots) runs
    // beside the trailing update of panel j (waves 1-3) instead of before it:
    //   A1(0);  for each panel:  A2 row solves | A3a: the next diagonal block only (one element per thread) |
    //                            wave 0: A1(next)  ||  waves 1-3: the rest of the trailing update (plus, split three ways,
    //                            the rows wave 0 would have owned)
    if (wave == 0) factor16(a, dl, idl, 0, lane, info, row0);
    __syncthreads();
    PF_CLK(1);
    for (int jb = 0; jb < 8; ++jb) {
        const int P = 16 * jb;
        const int base = P + 16, m = TILE - base;
        if (tid < m) {  // A2: rows below the diagonal block
            const int i = base + tid;
            double x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                double sacc = a[i * PF_LD + P + c];
#pragma unroll
                for (int k = 0; k < c; ++k) sacc -= x[k] * a[(P + k) * PF_LD + P + c];
                x[c] = sacc * idl[P + c];   // (a right-looking form -- 16 x (mul, fma) on the chain -- measured 1 % slower)
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) a[(P + c) * PF_LD + i] = x[c];
        }
        __syncthreads();
        PF_CLK(2);
        if (m == 0) break;
        {   // A3a: rank-16 update of the next diagonal block, element (ty, tx) per thread (done by wave 0 alone, 4 elements
            // per lane, it costs more than this extra barrier: 175k vs 163k cycles per block)
            const int ty = tid >> 4, tx = tid & 15;
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const double* col = a + (P + c) * PF_LD + base;
                acc += col[ty] * col[tx];
            }
            if (tx <= ty) a[(base + ty) * PF_LD + base + tx] -= acc;
        }
        __syncthreads();
#ifdef BOHIP_POTF2_CLOCKS
        const long long tq0 = clock64();
#endif
        if (wave == 0) {
            factor16(a, dl, idl, base, lane, info, row0);
        } else {
            const int u = tid - 64;   // 0..191: rows 4..15 of every sub-block
            trailing_dispatch<true, -1>(a, P, m >> 4, 4 + (u >> 4), u & 15);
            // rows 0..3 (wave 0 is busy): each of the three waves takes every third row of sub-blocks
            const int ty0 = (u >> 4) & 3, tx0 = u & 15;
            if (wave == 1) trailing_dispatch<true, 0>(a, P, m >> 4, ty0, tx0);
            else if (wave == 2) trailing_dispatch<true, 1>(a, P, m >> 4, ty0, tx0);
            else trailing_dispatch<true, 2>(a, P, m >> 4, ty0, tx0);
        }
#ifdef BOHIP_POTF2_CLOCKS
        if (tid == 0) pf_clocks[10] += clock64() - tq0;     // factor16 alone (wave 0)
        if (tid == 64) pf_clocks[11] += clock64() - tq0;    // trailing update alone (wave 1)
        if (tid == 192) pf_clocks[12] += clock64() - tq0;   // trailing update alone (wave 3)
#endif
        __syncthreads();
        PF_CLK(3);
    }
    // L out (coalesced along j): strict lower from the mirror, diagonal from dl, zeros above
    // (the strict upper triangle of the L block and of W, and the strict lower triangle of W', are zero from
    //  allocation and never written by anyone: k_build_cov leaves the upper triangle of diagonal tiles alone)
    for (int e = tid; e < TILE * TILE / 2; e += PF_THREADS) {  // 16-B stores; only the lower triangle
        const int i = e >> 6, j = (e & 63) * 2;
        if (j > i) continue;
        d2 v;
        v.x = (j < i) ? a[j * PF_LD + i] : dl[i];
        v.y = (j + 1 < i) ? a[(j + 1) * PF_LD + i] : (j + 1 == i ? dl[i] : 0.0);
        *reinterpret_cast<d2*>(Lblk + (int64_t)i * ld + j) = v;
    }
    PF_CLK(4);
    inverse_phase(a, idl, tid, Wblk, WTblk, ldw);
    PF_CLK(9);
}

// ------------------------------------------------------------------------------------------------
// C = alpha * A * B' + beta * C with both operands K-major, on the LDS-DMA engine of the scoring kernel
// (128 x 64 tiles, 3 LDS buffers, counted vmcnt).  Used by the factorisation: panel solve (B = inverse
// diagonal block) and trailing updates.  `diag_skip`: tiles strictly above the block diagonal of the
// GLOBAL matrix are skipped: tile (ti, tj64) is kept iff col0 + 64 tj64 < row0 + 128 (ti + 1)  (row0/col0 =
// global offsets of C in elements), so rectangular panels that straddle the diagonal need no special grid.
// ------------------------------------------------------------------------------------------------
struct GemmNTParams {
    const double* A;      // [M][K]  K-major, row stride lda
    const double* B;      // [N][K]  K-major, row stride ldb
    double* C;            // [M][N]
    double* CT;           // optional: also store C' here (CT[n][m], row stride ldct)
    int64_t lda, ldb, ldc, ldct;
    int64_t zA, zB, zC, zCT;  // element strides per batch (blockIdx.y)
    int mt, nt64, kc;         // 128-row tiles, 64-column tiles, 16-deep chunks
    double alpha, beta;
    int klo_from_m;   // A upper-triangular in (m, k): contraction starts at k = 128 ti
    int khi_from_m;   // A lower-triangular: contraction ends at k = 128 (ti + 1)
    int klo_from_n;   // B upper-triangular in (n, k): contraction starts at k = 128 (tj64 / 2)
    int diag_skip;    // skip tiles strictly above the global block diagonal (row0/col0 = element offsets of C)
    int ktot;         // with kz: total number of contraction chunks of the problem (the last slice is cut there)
    int kz;           // split-K batches: batch z covers contraction chunks [z kz, z kz + kc) of the whole problem; the
                      // triangular limits above are taken in those absolute chunk numbers (zA/zB carry the operand offsets)
    int64_t row0, col0;
    // ragged batches: tile (ti, tj64) of batch z exists iff  row_t0 + z row_ts + ti < total_t  and
    // col_t0 + z col_ts + tj64/2 < total_t  (all in 128-tiles); total_t <= 0 disables the check
    int row_t0, row_ts, col_t0, col_ts, total_t;
    // dataflow gating (kernels_chol.hip): every workgroup waits until *wait_flag >= wait_val before it touches an operand,
    // and each of its eight waves adds 1 to *signal when its part of the tile is in memory (a workgroup without a
    // tile adds 8), so a consumer can wait for 8 x gridDim.x x gridDim.y.  abort: see flag_wait_ge.
    const unsigned* wait_flag;
    unsigned wait_val;
    const unsigned* wait_flag2;   // optional second flag, same rule
    unsigned wait_val2;
    int wait_stride_ti, wait_stride_tj2;   // this workgroup waits on wait_flag[ti * stride_ti] and wait_flag2[(tj / 2) * stride_tj2]
    int wait2_tj2_max;            // > 0: the second flag only exists for tj / 2 < this; beyond, the B rows are rows of the A range:
    int wait2_rows;               //      wait2_rows = 1: wait on wait_flag[(tj / 2 - wait2_tj2_max) * wait_stride_ti], 0: no wait
    unsigned* signal_rows;        // per row tile: waves of the column tiles tj < signal_rows_ntj count into signal_rows[ti * signal_rows_stride]
    int signal_rows_ntj, signal_rows_stride;
    int coalesced;                // store C through LDS as 16-byte pieces by all eight waves (plain stores) instead of from the MFMA layout
    const unsigned* wait_flag3;   // optional third flag, the same for every workgroup (a whole earlier launch on another stream)
    unsigned wait_val3;
    unsigned* signal;
    unsigned* signal_row0;        // the workgroups of the first row tile (ti == 0, scheduled first) also count here
    unsigned* signal_col0;        // the workgroups of the first 128 columns (tj < 2) also count here
    int first_row_col;            // dispatch order: first row tile, then the first 128 columns of the other rows, then the rest
                                  // (the chain waits for exactly those tiles; the bulk of the launch follows)
    unsigned* abort_flag;
    unsigned long long spin_ticks;   // bound of the in-kernel wait (wall_clock64 ticks)
};

template <bool SWAVE = false>   // SWAVE: see gemm_tile_loop_glds3_ks (the throughput kernels below use it)
__device__ __forceinline__ void gemm_nt_body(const GemmNTParams& p, const int bid, const int z, double* smem) {
    // klo_from_n: the contraction of column tile tj starts at 128 (tj/2) -> low tj = long jobs: issue them first
    int ti = p.klo_from_n ? bid % p.mt : bid / p.nt64;
    int tj = p.klo_from_n ? bid / p.mt : bid % p.nt64;
    if (p.first_row_col && p.nt64 > 2) {
        const int b = bid, edge = p.nt64 + 2 * (p.mt - 1);
        if (b < p.nt64) { ti = 0; tj = b; }
        else if (b < edge) { ti = 1 + (b - p.nt64) / 2; tj = (b - p.nt64) & 1; }
        else { ti = 1 + (b - edge) / (p.nt64 - 2); tj = 2 + (b - edge) % (p.nt64 - 2); }
    }
    const bool no_tile = (p.diag_skip && p.col0 + CTILE * (int64_t)tj >= p.row0 + (int64_t)TILE * (ti + 1)) ||
                         (p.total_t > 0 && (p.row_t0 + z * p.row_ts + ti >= p.total_t || p.col_t0 + z * p.col_ts + (tj >> 1) >= p.total_t));
    if (no_tile) {
        if (p.signal && threadIdx.x == 0) atomicAdd(p.signal, 8u);
        if (p.signal_row0 && ti == 0 && threadIdx.x == 0) atomicAdd(p.signal_row0, 8u);
        if (p.signal_col0 && tj < 2 && threadIdx.x == 0) atomicAdd(p.signal_col0, 8u);
        if (p.signal_rows && tj < p.signal_rows_ntj && threadIdx.x == 0) atomicAdd(p.signal_rows + (int64_t)ti * p.signal_rows_stride, 8u);
        return;
    }
    if (p.wait_flag || p.wait_flag3) {
        if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                bool ok = !p.wait_flag ||
                          __hip_atomic_load(p.wait_flag + (int64_t)ti * p.wait_stride_ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= p.wait_val;
                if (ok && p.wait_flag2) {
                    const int tj2 = tj >> 1;
                    if (p.wait2_tj2_max > 0 && tj2 >= p.wait2_tj2_max)
                        ok = !p.wait2_rows || __hip_atomic_load(p.wait_flag + (int64_t)(tj2 - p.wait2_tj2_max) * p.wait_stride_ti, __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT) >= p.wait_val;
                    else
                        ok = __hip_atomic_load(p.wait_flag2 + (int64_t)tj2 * p.wait_stride_tj2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= p.wait_val2;
                }
                if (ok && p.wait_flag3) ok = __hip_atomic_load(p.wait_flag3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= p.wait_val3;
                if (ok) break;
                if (p.abort_flag && __hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                __builtin_amdgcn_s_sleep(4);
                if (wall_clock64() - t0 > (p.spin_ticks ? p.spin_ticks : CH_SPIN_TICKS_DEFAULT)) { if (p.abort_flag) atomicCAS(p.abort_flag, 0u, 1u); break; }   // see flag_wait_ge (the FIRST waiter to give up names the cause)
            }
        }
        __syncthreads();   // what the flag guards was written with agent-scope stores (kernels_chol.hip): no cache maintenance here
    }
    int kb = 0, ke = p.kc;
    const int koff = z * p.kz;
    if (p.klo_from_m) kb = max(kb, ti * (TILE / KC) - koff);
    if (p.klo_from_n) kb = max(kb, (tj >> 1) * (TILE / KC) - koff);
    if (p.khi_from_m) ke = min(ke, (ti + 1) * (TILE / KC) - koff);
    if (p.kz > 0) {
        if (p.ktot > 0) ke = min(ke, p.ktot - koff);
        if (kb >= ke) return;   // this slice does not touch the tile: its plane entry is never read
    }
    double acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    gemm_tile_loop_glds3_ks<4, 0, SWAVE>(p.A + z * p.zA + (int64_t)ti * TILE * p.lda, p.lda, p.B + z * p.zB + (int64_t)tj * CTILE * p.ldb,
                                         p.ldb, kb, ke, smem, acc);
    const int lane = threadIdx.x & 63, wave = SWAVE ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6), wr = wave >> 1, wc = wave & 1;
    double* C = p.C ? p.C + z * p.zC + (int64_t)ti * TILE * p.ldc + (int64_t)tj * CTILE : nullptr;
    if ((p.signal || p.coalesced) && C && (!p.CT || (p.coalesced && !p.signal))) {
        // A running kernel reads this tile: the stores are agent-scope (write-through).  Issued straight from the MFMA
        // accumulator layout they are 8-byte pieces in 32-byte runs -- 15-25 us per tile, longer than the contraction.  So
        // the tile takes a turn through LDS and leaves as 16-byte pieces, 1 KB contiguous per wave instruction, by all 8 waves.
        constexpr int TS = CTILE + 2;
        double* Tl = smem;   // [128][66]: the staging buffers are free (the loop ended on a barrier)
        if (wave < 4) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) Tl[acc_row(lane, wr, mi) * TS + acc_col<4>(lane, wc, nj)] = p.alpha * acc[mi][nj];
        }
        __syncthreads();
        const int64_t grow0 = p.row0 + (int64_t)ti * TILE, gcol0 = p.col0 + (int64_t)tj * CTILE;
#pragma unroll
        for (int u = 0; u < (TILE * CTILE / 2) / GEMM_THREADS_8; ++u) {
            const int piece = threadIdx.x + GEMM_THREADS_8 * u, r = piece >> 5, c = (piece & 31) * 2;
            double* dst = C + (int64_t)r * p.ldc + c;
            const bool k0 = !(p.diag_skip && gcol0 + c > grow0 + r), k1 = !(p.diag_skip && gcol0 + c + 1 > grow0 + r);
            if (!k0) continue;   // (k1 implies k0)
            d2 v = *reinterpret_cast<const d2*>(Tl + r * TS + c);
            if (p.beta != 0.0) {
                const d2 old = *reinterpret_cast<const d2*>(dst);
                v.x += p.beta * old.x;
                v.y += p.beta * old.y;
            }
            if (!p.signal) {   // (coalesced only: nobody reads the tile before this kernel ends)
                if (k1) *reinterpret_cast<d2*>(dst) = v;
                else *dst = v.x;
            } else if (k1) {
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
            } else {
                __hip_atomic_store(dst, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (!p.signal) {
            if (p.CT) {   // the transpose from the same LDS tile: CT[c][r], 16-byte pieces along r (no diag_skip / beta users)
                double* CTt = p.CT + z * p.zCT + (int64_t)tj * CTILE * p.ldct + (int64_t)ti * TILE;
#pragma unroll
                for (int u = 0; u < (TILE * CTILE / 2) / GEMM_THREADS_8; ++u) {
                    const int piece = threadIdx.x + GEMM_THREADS_8 * u, c = piece >> 6, r = (piece & 63) * 2;
                    d2 v;
                    v.x = Tl[r * TS + c];
                    v.y = Tl[(r + 1) * TS + c];
                    *reinterpret_cast<d2*>(CTt + (int64_t)c * p.ldct + r) = v;
                }
            }
            return;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // landed (a workgroup-scope fence emits no such wait)
        if (lane == 0) {
            atomicAdd(p.signal, 1u);
            if (p.signal_row0 && ti == 0) atomicAdd(p.signal_row0, 1u);
            if (p.signal_col0 && tj < 2) atomicAdd(p.signal_col0, 1u);
            if (p.signal_rows && tj < p.signal_rows_ntj) atomicAdd(p.signal_rows + (int64_t)ti * p.signal_rows_stride, 1u);
        }
        return;
    }
    if (wave >= 4) return;  // waves 4-7 only contributed partial sums (already folded into waves 0-3)
    double* CT = p.CT ? p.CT + z * p.zCT + (int64_t)tj * CTILE * p.ldct + (int64_t)ti * TILE : nullptr;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int r = acc_row(lane, wr, mi);
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
            const int c = acc_col<4>(lane, wc, nj);
            // symmetric updates never touch the strict upper triangle of the global matrix (it must stay zero)
            if (p.diag_skip && p.col0 + (int64_t)tj * CTILE + c > p.row0 + (int64_t)ti * TILE + r) continue;
            double v = p.alpha * acc[mi][nj];
            if (C) {
                double* dst = C + (int64_t)r * p.ldc + c;
                if (p.beta != 0.0) v += p.beta * *dst;
                *dst = v;
            }
            if (CT) CT[(int64_t)c * p.ldct + r] = v;
        }
    }
}

__global__ __launch_bounds__(GEMM_THREADS_8, 2) void k_gemm_nt(GemmNTParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    gemm_nt_body(p, (int)blockIdx.x, (int)blockIdx.y, smem);
}
// Two launches in one grid: blocks [0, na) run `a`, the rest run `b`.  cholesky_dataflow2 puts block k's update and block k+1's
// row solve into one launch: the solve's workgroups wait (in-kernel) for counters the update's first workgroups raise -- those
// have lower block numbers, are dispatched first and wait for nothing in this launch -- so both are resident when the inverse
// of the next diagonal block arrives, and ONE in-order stream carries every flagged launch.
// (these two are compiled for FOUR waves per SIMD = 128 VGPRs = two workgroups per CU: throughput launches.  k_gemm_nt itself
// stays at 129 VGPRs / one workgroup per CU, which suits the small launches beside the first dataflow form's chain better.)
__global__ __launch_bounds__(GEMM_THREADS_8, 4) void k_gemm_nt_hi(GemmNTParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    gemm_nt_body<true>(p, (int)blockIdx.x, (int)blockIdx.y, smem);
}
__global__ __launch_bounds__(GEMM_THREADS_8, 4) void k_gemm_nt_pair(GemmNTParams a, GemmNTParams b, int na) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if ((int)blockIdx.x < na) gemm_nt_body<true>(a, (int)blockIdx.x, 0, smem);
    else gemm_nt_body<true>(b, (int)blockIdx.x - na, 0, smem);
}
// the same with four parameter sets: blocks [0, n0) run p0, [n0, n0+n1) p1, [.., +n2) p2, the rest p3
__global__ __launch_bounds__(GEMM_THREADS_8, 4) void k_gemm_nt_quad(GemmNTParams p0, GemmNTParams p1, GemmNTParams p2, GemmNTParams p3,
                                                                 int n0, int n1, int n2) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x;
    if (b < n0) gemm_nt_body<true>(p0, b, 0, smem);
    else if (b < n0 + n1) gemm_nt_body<true>(p1, b - n0, 0, smem);
    else if (b < n0 + n1 + n2) gemm_nt_body<true>(p2, b - n0 - n1, 0, smem);
    else gemm_nt_body<true>(p3, b - n0 - n1 - n2, 0, smem);
}

// dst[i][j] = src[i][j] on every 128-tile strictly below the diagonal tiles (the factorisation parks the solved
// panels in the scratch matrix so the panel solve never runs in place; this moves them home in one pass).
__global__ __launch_bounds__(256) void k_copy_offdiag_tiles(const double* __restrict__ src, double* __restrict__ dst,
                                                            int64_t ld, int T) {
    const int b = blockIdx.x;  // strictly-lower tile index: (ti, tj), ti > tj
    int ti = (int)((sqrt(8.0 * b + 1.0) + 1.0) * 0.5);
    while (ti * (ti - 1) / 2 > b) --ti;
    while ((ti + 1) * ti / 2 <= b) ++ti;
    const int tj = b - ti * (ti - 1) / 2;
    if (ti >= T) return;
    const d2* s2 = reinterpret_cast<const d2*>(src + (int64_t)ti * TILE * ld + (int64_t)tj * TILE);
    d2* t2 = reinterpret_cast<d2*>(dst + (int64_t)ti * TILE * ld + (int64_t)tj * TILE);
    for (int e = threadIdx.x; e < TILE * TILE / 2; e += 256) {
        const int r = e >> 6, c = e & 63;
        t2[(int64_t)r * (ld / 2) + c] = s2[(int64_t)r * (ld / 2) + c];
    }
}


// ------------------------------------------------------------------------------------------------
// A3: alpha = W' (W (y - beta)): k_sub_mean, then two passes of k_rows_trimv (below) on W and on W'.
// ------------------------------------------------------------------------------------------------
__global__ void k_sub_mean(const double* __restrict__ y, double beta, int64_t N, double* __restrict__ r) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < N) r[i] = y[i] - beta;
}
// mll = -0.5 r'alpha - sum log L_ii - N/2 log(2 pi)   (single workgroup, fixed order)
__global__ __launch_bounds__(256) void k_mll(const double* __restrict__ L, int64_t ld, int64_t N,
                                             const double* __restrict__ r, const double* __restrict__ alpha,
                                             double* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += 256) s += -0.5 * r[i] * alpha[i] - log(L[i * ld + i]);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] - 0.5 * (double)N * log(2.0 * M_PI);
}

// ------------------------------------------------------------------------------------------------
// N2: gradient of the marginal likelihood (role of GaussianProcesses.jl update_target_and_dtarget!, called from
// optimizemodel!, reference src/models/gp.jl:59-64):   d mll / d theta = 1/2 tr((alpha alpha' - cK^-1) dcK/dtheta).
// cK^-1 = W'W is one lower-triangular k_gemm_nt on the resident W' (N^3/3 flops on the MFMA engine); this kernel
// is the single pass over it: thread = column j (coordinates in registers), the block walks rows_per_block rows,
// entries of K and dK/dll_k are recomputed from X (cheaper than a second N x N buffer).  Per-block partial sums
// [block][NP], NP = d + 3: {logNoise, beta, ll_0..ll_{d-1}, logsig}; k_dmll_final adds them in a fixed order.
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void k_dmll_parts(const double* __restrict__ X, int64_t N, KernelHyper hp,
                                                    double noise_var, const double* __restrict__ Kinv, int64_t ld,
                                                    const double* __restrict__ alpha, int rows_per_block,
                                                    double* __restrict__ parts) {
    const int d = hp.d, NP = d + 3;
    const int64_t j = blockIdx.x * 256 + threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t i1 = min(N, i0 + rows_per_block);
    double* out = parts + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * NP;
    if ((int64_t)blockIdx.x * 256 > i1 - 1) {   // block entirely above the diagonal
        for (int k = threadIdx.x; k < NP; k += 256) out[k] = 0.0;
        return;
    }
    double xj[DT], acc[DT], a_sig = 0.0, a_noise = 0.0, a_mean = 0.0;
#pragma unroll
    for (int k = 0; k < DT; ++k) {
        xj[k] = (j < N && k < d) ? X[j * d + k] : 0.0;
        acc[k] = 0.0;
    }
    const double aj = j < N ? alpha[j] : 0.0;
    for (int64_t i = i0; i < i1; ++i) {
        if (j > i) continue;   // (j <= i < N)
        const double ai = alpha[i];
        const double G = (ai * aj - Kinv[i * ld + j]) * (i == j ? 0.5 : 1.0);
        double t[DT], r = 0.0;
#pragma unroll
        for (int k = 0; k < DT; ++k) {
            const double dx = (k < d) ? X[i * d + k] - xj[k] : 0.0;
            t[k] = (k < d) ? hp.il2[k] * (dx * dx) : 0.0;
            r += t[k];
        }
        double Kij, fac;
        if (hp.kern == KERN_MAT52ARD) {
            const double sq = sqrt(5.0) * sqrt(r), e = exp(-sq);
            Kij = hp.sigma2 * (1.0 + sq + 5.0 / 3.0 * r) * e;
            fac = 5.0 / 3.0 * hp.sigma2 * (1.0 + sq) * e;
        } else {
            Kij = hp.sigma2 * exp(-0.5 * r);
            fac = Kij;
        }
        const double gf = G * fac;
#pragma unroll
        for (int k = 0; k < DT; ++k) acc[k] += gf * t[k];
        a_sig += G * 2.0 * Kij;
        if (i == j) {
            a_noise += G * 2.0 * noise_var;   // G carries the 1/2
            a_mean += ai;
        }
    }
    // block reduction of NP accumulators: wave shuffles, then 4 wave leaders through LDS
    __shared__ double red[4][DMAX + 3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto wsum = [](double v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    a_noise = wsum(a_noise); a_mean = wsum(a_mean); a_sig = wsum(a_sig);
#pragma unroll
    for (int k = 0; k < DT; ++k) acc[k] = wsum(acc[k]);
    if (lane == 0) {
        red[wave][0] = a_noise; red[wave][1] = a_mean;
#pragma unroll
        for (int k = 0; k < DT; ++k) if (k < d) red[wave][2 + k] = acc[k];
        red[wave][2 + d] = a_sig;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < NP; k += 256) out[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
}
// out[k] = sum over blocks of parts[b][k] in a fixed strided order + tree; SEIso folds the d length entries into one.
// One workgroup per output parameter.
__global__ __launch_bounds__(256) void k_dmll_final(const double* __restrict__ parts, int64_t nblocks, int NP, int d,
                                                    int iso, double* __restrict__ out) {
    __shared__ double red[256];
    const int k = blockIdx.x;   // output slot
    // iso: slots are {noise, mean, ll, logsig}: slot 2 gathers part columns 2 .. 2+d-1, slot 3 reads column 2+d
    int c0 = k, c1 = k + 1;
    if (iso) { if (k == 2) { c0 = 2; c1 = 2 + d; } else if (k == 3) { c0 = 2 + d; c1 = 3 + d; } }
    double s = 0.0;
    for (int64_t b = threadIdx.x; b < nblocks; b += 256)
        for (int c = c0; c < c1; ++c) s += parts[b * NP + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = red[0];
}

}  // namespace bohip

// ================================================================================================
// A2': incremental extension of the factor by p <= APPEND_PMAX observations (role of ElasticPDMats
// append!: U12 = U11' \ K12, U22 = chol(K22 - U12'U12), reference src/models/gp.jl:11).
// With W = L^-1 resident the new rows are products, not solves:
//     L21 = K21 W11'                      (k_rows_trimv  : one pass over W, HBM-bound)
//     L22 = chol(K22 - L21 L21'), W22     (k_schur_chol  : one workgroup)
//     W21 = -W22 (L21 W11)                (k_rows_trimv on W': one pass, then k_apply_w22)
// ================================================================================================
namespace bohip {

constexpr int APPEND_PMAX = 32;   // larger batches take the full refit
constexpr int APPEND_CHUNK = 8;   // right-hand sides per pass over W

// New rows [N0, Npad1) of cK into L (cols 0..i) and identity padding rows into L and W.
__global__ __launch_bounds__(256) void k_cov_rows(const double* __restrict__ X, int64_t N0, int64_t N1, int64_t Npad1,
                                                  KernelHyper hp, double noise, double* __restrict__ L,
                                                  double* __restrict__ W, double* __restrict__ WT, int64_t ld) {
#pragma clang fp contract(off)
    const int64_t i = N0 + blockIdx.y;
    const int64_t j = blockIdx.x * 256 + threadIdx.x;
    if (i >= Npad1 || j >= Npad1) return;
    const int d = hp.d;
    if (i < N1) {
        if (j > i) { if (j < N1) L[i * ld + j] = 0.0; return; }
        double r = 0.0;
        for (int k = 0; k < d; ++k) {
            const double t = X[i * d + k] - X[j * d + k];
            r += hp.il2[k] * (t * t);
        }
        double v = cov_from_r(hp.kern, hp.sigma2, r);
        if (i == j) v += noise;
        L[i * ld + j] = v;
    } else {  // identity padding (also clears a stale alpha row)
        const double v = (i == j) ? 1.0 : 0.0;
        L[i * ld + j] = v;
        W[i * ld + j] = v;
        WT[j * ld + i] = v;
    }
}

// out[r][j] = sum_{k<=j} W[j][k] * rows[r][k]   for j < N0, r < P        (upper != 0: W is an UPPER-triangular K-major matrix, rows of
// W', and the sum runs over k in [j, N0):  out[r][j] = sum_{k>=j} W'[j][k] * rows[r][k] = (rows W)[r][j] for the lower-triangular W).
// The row-wise triangular product of the small-batch path (the reference's default: 10 L-BFGS restarts per acquire_max, V' = K*' W'
// and U' = V' W per evaluation), of alpha = W'(W(y - beta)) and of the incremental append.  STREAMING form (round 4): the product
// is one pass over 8 N^2 / 2 bytes of W and nothing else, yet the round-3 kernel (one workgroup per 8 rows x 8 right-hand sides,
// k-strided loads into 64 running sums per thread, 205 VGPRs) took 24 us for 36 MB at N = 3000 with ten right-hand sides: two
// groups of right-hand sides = two passes, and every workgroup a chain of dependent load -> use round trips
// (tools/ubench_trimv.hip; a plain read of the same 36 MB takes 5.5 us, tools/ubench_readbw.hip).  Here:
//   * ONE workgroup per CU (512 threads = 8 waves = 2 contraction halves x 4 groups of 4 right-hand sides) walks a list of 8-row
//     blocks of W, the blocks dealt out in a snake over the cost-sorted list so that every workgroup streams about the same bytes;
//   * a block is consumed in chunks of 256 contraction indices; a chunk's 8 x 256 tile of W and 16 x 256 tile of right-hand sides
//     arrive by LDS-DMA into a ring of D + 1 slots, D steps ahead of their use: every step is exactly six 1-KiB pieces per wave
//     (two for a wave whose right-hand sides do not exist), so the counted s_waitcnt vmcnt(6 (D - 1)) is exact, nothing asynchronous
//     lands in a register, and the fragment reads are inline asm (hipcc would answer them with vmcnt(0), see gemm_core.h);
//   * up to 16 right-hand sides ride on ONE pass over W (blockIdx.y: further groups of 16).
// Measured (N = 3000, lower / upper): 10 right-hand sides 15.6 / 16.7 us against 24.1 / 22.9; 8: 13.0 / 14.4 against 17.4 / 16.7;
// 1: 12.0 / 13.4 against 13.0 / 12.7; N = 10^4, 10 right-hand sides: 104 / 111 against 142 / 137 (profiles/r04_ubench_trimv.txt).
// Summation order of one (row, right-hand side) pair: lane l of contraction half h adds its two products of every chunk in chunk
// order; the 64 lanes are combined by recursive halving (lane bit 5 first), then half 0 + half 1 -- fixed whatever else is in the
// batch and whichever workgroup gets the block (reference property test/acquisitionfunctions.jl:8-11: batch == single, bit for bit).
// Needs: ld even, W / rows 16-byte aligned, and readable memory up to the next multiple of 256 columns behind N0 in every row read
// (the buffers are sized for it: a row's overshoot is the start of the next row; the values are masked by index, not by zeroes).
constexpr int TRIMV_THREADS = 512, TRIMV_D = 2;
constexpr int trimv_lds_bytes(int D) { return ((D + 1) * 24 * 256 + 8 * 32) * 8; }
struct TrimvWalker {   // the (row block, chunk) steps of one workgroup, in order (all fields wave-uniform)
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
template <int D>
__global__ __launch_bounds__(TRIMV_THREADS) void k_trimv_stream(const double* __restrict__ W, int64_t ld, int64_t N0_,
                                                   const double* __restrict__ rows, int64_t ldr, int P,
                                                   double* __restrict__ out, int64_t ldo, int upper, const unsigned* __restrict__ go = nullptr) {
    if (go && *go == 0u) return;   // (the free-running ascent: a pass that was queued before the host knew that every start point had stopped)
    constexpr int NS = D + 1, WT = 8 * 256, SLOT = 24 * 256, VM = 6;   // doubles per W tile / per ring slot; VMEM instructions per step and wave
    extern __shared__ __attribute__((aligned(16))) double ring[];   // [NS][8 + 16][256] + red[8][32]
    double* red = ring + NS * SLOT;
    const int N0 = (int)N0_;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave & 1, rg = wave >> 1;
    const int r_base = blockIdx.y * 16 + rg * 4;                                        // this wave's four right-hand sides
    const bool live = r_base < P;            // a wave whose right-hand sides do not exist brings W only (2 pieces per step, not 6)
    TrimvWalker pw, cw;
    pw.init(gridDim.x, blockIdx.x, upper, N0, 8);
    cw.init(gridDim.x, blockIdx.x, upper, N0, 8);
    const uint32_t ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)ring;
    const double* rsrc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rsrc[r] = rows + (int64_t)min(r_base + r, max(P - 1, 0)) * ldr + kh * 128 + lane * 2;
    auto issue = [&](const TrimvWalker& x, int slot) {
        const int row = min(x.j0 + wave, N0 - 1);                   // W: wave w brings row w of the tile (two 1-KiB halves)
        const double* src = W + (int64_t)row * ld + (x.c << 8) + lane * 2;
        double* dst = ring + slot * SLOT + wave * 256;
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)src, (lds_void_ptr)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src + 128), (lds_void_ptr)(dst + 128), 16, 0, 0);
        if (live) {
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
        if (live) {
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
        if (live) { rcur[0] = ds_read128<0>(ar); rcur[1] = ds_read128<2048>(ar); rcur[2] = ds_read128<4096>(ar); rcur[3] = ds_read128<6144>(ar); }
        else { rcur[0] = rcur[1] = rcur[2] = rcur[3] = d2{0.0, 0.0}; }
        wr[0] = ds_read128<0>(a); wr[1] = ds_read128<2048>(a); wr[2] = ds_read128<4096>(a); wr[3] = ds_read128<6144>(a);
        wr[4] = ds_read128<8192>(a); wr[5] = ds_read128<10240>(a); wr[6] = ds_read128<12288>(a); wr[7] = ds_read128<14336>(a);
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
        if (live) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
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

// One workgroup: S = K22 - L21 L21' (p x p), L22 = chol(S), W22 = L22^-1.
// K22 sits in L[N0+r][N0+s]; L21 in L[N0+r][0..N0).  Results overwrite L22 in place and go to W22.
// The p (p + 1) / 2 inner products of the Schur complement, ONE WORKGROUP EACH (k_schur_chol's own loop takes them one after the other,
// a block reduction and ~2.7 us apiece: 44 us at p = 5, 375 us at p = 16 -- the README's `repetitions = 5` appends five rows per
// iteration).  Same threads, same strides, same reduction tree as there: the same bits.  Sout[r][s], row stride APPEND_PMAX.
__global__ __launch_bounds__(256) void k_schur_dots(const double* __restrict__ L, int64_t ld, int64_t N0, int p, double* __restrict__ Sout) {
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = blockIdx.x;
    int r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= e) ++r;
    while (r * (r + 1) / 2 > e) --r;
    const int sidx = e - r * (r + 1) / 2;
    const double* a = L + (N0 + r) * ld;
    const double* b = L + (N0 + sidx) * ld;
    double acc = 0.0;
    for (int64_t k = tid; k < N0; k += 256) acc += a[k] * b[k];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) Sout[r * APPEND_PMAX + sidx] = a[N0 + sidx] - ((red[0] + red[1]) + (red[2] + red[3]));
}
// Sin != nullptr: the Schur complement comes from k_schur_dots.
__global__ __launch_bounds__(256) void k_schur_chol(double* __restrict__ L, double* __restrict__ W, double* __restrict__ WT,
                                                    int64_t ld, int64_t N0, int p, int* __restrict__ info, const double* __restrict__ Sin) {
    __shared__ double S[APPEND_PMAX][APPEND_PMAX + 1];
    __shared__ double Winv[APPEND_PMAX][APPEND_PMAX + 1];
    __shared__ double red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (Sin) {
        for (int e = tid; e < p * p; e += 256) {
            const int r = e / p, c = e % p;
            if (c <= r) S[r][c] = Sin[r * APPEND_PMAX + c];
        }
        __syncthreads();
    }
    for (int e = 0; e < (Sin ? 0 : p * (p + 1) / 2); ++e) {  // (r, s), s <= r; every thread strides the dot product
        int r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while ((r + 1) * (r + 2) / 2 <= e) ++r;
        while (r * (r + 1) / 2 > e) --r;
        const int sidx = e - r * (r + 1) / 2;
        const double* a = L + (N0 + r) * ld;
        const double* b = L + (N0 + sidx) * ld;
        double acc = 0.0;
        for (int64_t k = tid; k < N0; k += 256) acc += a[k] * b[k];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (tid == 0) S[r][sidx] = a[N0 + sidx] - ((red[0] + red[1]) + (red[2] + red[3]));
        __syncthreads();
    }
    // Cholesky of the p x p block, column by column: the diagonal on thread 0, the column's entries one thread per row; then the inverse, one
    // thread per column.  Every entry is computed by the expression, in the order, the one-thread version used (same bits) -- that one took
    // p^3 / 3 dependent LDS round trips (89 us at p = 16), this takes ~p^2.
    for (int j = 0; j < p; ++j) {
        if (tid == 0) {
            double ajj = S[j][j];
            for (int k = 0; k < j; ++k) ajj -= S[j][k] * S[j][k];
            if (!(ajj > 0.0)) { atomicCAS(info, 0, (int)(N0 + j + 1)); ajj = 1.0; }
            S[j][j] = sqrt(ajj);
        }
        __syncthreads();
        if (tid > j && tid < p) {
            const int i = tid;
            const double dd = S[j][j];
            double v = S[i][j];
            for (int k = 0; k < j; ++k) v -= S[i][k] * S[j][k];
            S[i][j] = v / dd;
        }
        __syncthreads();
    }
    if (tid < p) {
        const int c = tid;
        Winv[c][c] = 1.0 / S[c][c];
        for (int i = c + 1; i < p; ++i) {
            double v = 0.0;
            for (int k = c; k < i; ++k) v += S[i][k] * Winv[k][c];
            Winv[i][c] = -v / S[i][i];
        }
    }
    __syncthreads();
    for (int e = tid; e < p * p; e += 256) {
        const int r = e / p, c = e % p;
        L[(N0 + r) * ld + N0 + c] = (c <= r) ? S[r][c] : 0.0;
        W[(N0 + r) * ld + N0 + c] = (c <= r) ? Winv[r][c] : 0.0;
        WT[(N0 + c) * ld + N0 + r] = (c <= r) ? Winv[r][c] : 0.0;
    }
}

// W21[r][c] = -sum_{s<=r} W22[r][s] * T[s][c]   with T = L21 W11 (rows of T at Tm, row stride ldt); W21' goes to W' too.
__global__ __launch_bounds__(256) void k_apply_w22(double* __restrict__ W, double* __restrict__ WT, int64_t ld, int64_t N0, int p,
                                                   const double* __restrict__ Tm, int64_t ldt) {
    __shared__ double w22[APPEND_PMAX * APPEND_PMAX];    // (read p (p + 1) / 2 times by every thread: from LDS, not through the vector cache)
    for (int e = threadIdx.x; e < p * p; e += 256) w22[e] = W[(N0 + e / p) * ld + N0 + e % p];
    __syncthreads();
    const int64_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N0) return;
    double T[APPEND_PMAX];
    for (int s = 0; s < p; ++s) T[s] = Tm[(int64_t)s * ldt + c];
    for (int r = 0; r < p; ++r) {
        double v = 0.0;
        for (int s = 0; s <= r; ++s) v += w22[r * p + s] * T[s];
        W[(N0 + r) * ld + c] = -v;
        WT[c * ld + N0 + r] = -v;
    }
}

// alpha after an append, incrementally.  With W_new = [W11 0; W21 W22] and the residual r = y - beta (old entries unchanged):
//     u = W_new r = [u_old; u2],   u2 = W21 r1 + W22 r2            (u_old = W11 r1 is what compute_alpha left in dt)
//     alpha_new = W_new' u = [alpha_old + W21' u2; W22' u2]
// O(N p) instead of the two passes over W that the full product takes (N = 3000: 36 of an append's 107 us, N = 10^4: 175 of 450).
// Fixed summation orders (u2: one workgroup per new row, thread-strided partial sums, waves by butterfly, waves in order; alpha: r ascending).
// k_alpha_append_u: r2 -> dr, u2 -> dt.   k_alpha_append_apply: alpha (and its copy in W's padding row N1).
__global__ __launch_bounds__(1024) void k_alpha_append_u(const double* __restrict__ W, int64_t ld, int64_t N0, int p,
                                                         const double* __restrict__ y, double beta, double* __restrict__ r,
                                                         double* __restrict__ t) {
    // workgroup rr: u2[rr] (one new row each: p rows in parallel)
    __shared__ double red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, rr = blockIdx.x;
    double a = 0.0;
    for (int64_t c = tid; c < N0; c += 1024) a += W[(N0 + rr) * ld + c] * r[c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off);
    if (lane == 0) red[wave] = a;
    __syncthreads();
    if (tid == 0) {
        double v = 0.0;
        for (int w = 0; w < 16; ++w) v += red[w];
        for (int s2 = 0; s2 <= rr; ++s2) v += W[(N0 + rr) * ld + N0 + s2] * (y[N0 + s2] - beta);
        t[N0 + rr] = v;
        r[N0 + rr] = y[N0 + rr] - beta;      // (nobody reads the new residuals in this kernel: the old ones end at N0)
    }
}
__global__ __launch_bounds__(256) void k_alpha_append_apply(double* __restrict__ W, int64_t ld, int64_t N0, int p,
                                                            const double* __restrict__ t, double* __restrict__ alpha) {
    const int64_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N0 + p) return;
    double a;
    if (c < N0) {
        a = alpha[c];
        for (int rr = 0; rr < p; ++rr) a += W[(N0 + rr) * ld + c] * t[N0 + rr];
    } else {
        const int s2 = (int)(c - N0);
        a = 0.0;
        for (int rr = s2; rr < p; ++rr) a += W[(N0 + rr) * ld + N0 + s2] * t[N0 + rr];
    }
    alpha[c] = a;
    W[(N0 + p) * ld + c] = a;     // alpha' in the first padding row of W (compute_alpha's copy)
}

// ---- small-batch posterior (chunks of SMALL_R, up to ~100 candidates): the default use of the reference (10 L-BFGS
// restarts) scores a handful of candidates per call.  A 128 x 64 MFMA tile would be almost empty and its single job per
// row tile latency-bound by the longest K loop; instead V' = K*' W' is computed row-wise like the incremental append
// (k_rows_trimv), k_small_finish (kernels_score.hip) turns V' into (sum v^2, mu - beta) and scores, and U' = V' W is the
// same row-wise kernel on W'.
constexpr int SMALL_R = 32;

}  // namespace bohip
