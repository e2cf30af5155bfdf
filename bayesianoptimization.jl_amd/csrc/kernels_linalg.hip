// kernels_linalg.hip -- model-update kernels (rows A1-A3 of SURVEY.md section 8):
//   k_build_cov      SEArd/SEIso/Mat52Ard kernel-matrix assembly, lower 128-tiles (HBM-write bound)
//   k_potf2_inv      128x128 diagonal block: Cholesky + triangular inverse in LDS (latency bound)
//   k_gemm           generic FP64 MFMA contraction C = alpha*A*op(B) + beta*C (panel solve, trailing
//                    update, recursive triangular inverse W = L^-1)
//   k_trimv / k_trimv_t, k_sub_mean, k_mll  -- alpha = W'(W(y - beta)) and the marginal likelihood
// Reference call sites replaced: update!/append!/fit! in src/models/gp.jl:11-18 (GaussianProcesses.jl
// update_cK! + ElasticPDMats Cholesky behind them).
#include "gemm_core.h"

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

__global__ __launch_bounds__(256) void k_build_cov(const double* __restrict__ X, int64_t N, int64_t Npad,
                                                   KernelHyper hp, double noise, double* __restrict__ K,
                                                   int64_t ld, int64_t row_begin) {
#pragma clang fp contract(off)
    extern __shared__ double sm[];  // xi[64][d], xj[64][d]
    const int d = hp.d;
    double* xi = sm;
    double* xj = sm + 64 * d;
    // tile enumeration: blockIdx.y = tile row (64-granular, offset by row_begin), blockIdx.x = tile col
    const int64_t i0 = row_begin + (int64_t)blockIdx.y * 64, j0 = (int64_t)blockIdx.x * 64;
    if (j0 / TILE > i0 / TILE) return;  // strictly upper 128-tile: never touched (stays zero)
    for (int e = threadIdx.x; e < 64 * d; e += 256) {
        const int64_t gi = i0 + e / d, gj = j0 + e / d;
        xi[e] = gi < N ? X[gi * d + e % d] : 0.0;
        xj[e] = gj < N ? X[gj * d + e % d] : 0.0;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // tx -> column (coalesced), ty -> 16 rows each
    const int64_t gj = j0 + tx;
    for (int rr = 0; rr < 16; ++rr) {
        const int li = ty * 16 + rr;
        const int64_t gi = i0 + li;
        double v;
        if (gi < N && gj < N) {
            double r = 0.0;
            for (int k = 0; k < d; ++k) {
                const double t = xi[li * d + k] - xj[tx * d + k];
                r += hp.il2[k] * (t * t);
            }
            v = cov_from_r(hp.kern, hp.sigma2, r);
            if (gi == gj) v += noise;
        } else {
            v = (gi == gj) ? 1.0 : 0.0;
        }
        if (gi < Npad && gj < Npad) K[gi * ld + gj] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// A2 (diagonal block): in-LDS Cholesky of one 128 x 128 block followed by its triangular inverse.
// Right-looking, ONE barrier per column: the scaled column j is written to the mirror position
// (row j of the strict upper triangle, never read by the trailing update) and the diagonal to dl[],
// so the unscaled column stays readable by every thread during the rank-1 update.
// Then thread c solves L w = e_c by forward substitution; w overwrites column c of the (now free)
// lower triangle.  Output: L block (in place, strict upper zeroed) and W block = L^-1.
// info: first failing pivot (1-based global index) if the block is not positive definite.
// ------------------------------------------------------------------------------------------------
constexpr int PF_LD = TILE + 1;
constexpr int POTF2_LDS_BYTES = (TILE * PF_LD + TILE) * 8;

__global__ __launch_bounds__(256) void k_potf2_inv(double* __restrict__ Lblk, int64_t ld, double* __restrict__ Wblk,
                                                   int64_t ldw, int* __restrict__ info, int row0) {
#pragma clang fp contract(off)
    extern __shared__ double sm[];
    double* a = sm;
    double* dl = sm + TILE * PF_LD;
    const int tid = threadIdx.x;
    for (int e = tid; e < TILE * TILE; e += 256) {
        const int i = e >> 7, j = e & 127;
        a[i * PF_LD + j] = (j <= i) ? Lblk[(int64_t)i * ld + j] : 0.0;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    for (int j = 0; j < TILE; ++j) {
        double ajj = a[j * PF_LD + j];
        if (!(ajj > 0.0)) {
            if (tid == 0) atomicCAS(info, 0, row0 + j + 1);
            ajj = 1.0;
        }
        const double dd = sqrt(ajj), inv = 1.0 / dd;
        if (tid == 0) dl[j] = dd;
        for (int i = j + 1 + ty; i < TILE; i += 16) {
            const double lij = a[i * PF_LD + j] * inv;
            for (int k = j + 1 + tx; k <= i; k += 16) a[i * PF_LD + k] -= lij * (a[k * PF_LD + j] * inv);
        }
        for (int i = j + 1 + tid; i < TILE; i += 256) a[j * PF_LD + i] = a[i * PF_LD + j] * inv;
        __syncthreads();
    }
    // write L (coalesced along j)
    for (int e = tid; e < TILE * TILE; e += 256) {
        const int i = e >> 7, j = e & 127;
        const double v = (j < i) ? a[j * PF_LD + i] : (j == i ? dl[i] : 0.0);
        Lblk[(int64_t)i * ld + j] = v;
    }
    __syncthreads();
    // inverse: thread c owns column c of W, stored at a[i][c], i >= c.  L[i][k] (i>k) is a[k][i].
    if (tid < TILE) {
        const int c = tid;
        a[c * PF_LD + c] = 1.0 / dl[c];
        for (int i = c + 1; i < TILE; ++i) {
            double s = 0.0;
            for (int k = c; k < i; ++k) s += a[k * PF_LD + i] * a[k * PF_LD + c];
            a[i * PF_LD + c] = -s / dl[i];
        }
    }
    __syncthreads();
    for (int e = tid; e < TILE * TILE; e += 256) {
        const int i = e >> 7, c = e & 127;
        Wblk[(int64_t)i * ldw + c] = (c <= i) ? a[i * PF_LD + c] : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------
// Generic contraction  C[z] = alpha * A[z] * op(B[z]) + beta * C[z]  on 128-tiles.
//   A K-major [M][K]; B K-major [N][K] (B_NMAJOR=false) or N-major [K][N] (true); C row-major.
//   lower_tiles : only tiles i >= j (symmetric trailing update)
//   klo_from_n  : contraction starts at k = 128*j  (B lower-triangular in N-major form)
//   khi_from_m  : contraction ends at  k = 128*(i+1) (A lower-triangular)
//   ragged M    : batch z covers tile rows [z_row0 + z*z_rstride, ...); tiles beyond total_rows exit
// ------------------------------------------------------------------------------------------------
struct GemmParams {
    const double* A;
    const double* B;
    double* C;
    int64_t lda, ldb, ldc;
    int64_t zA, zB, zC;
    int mt, nt, kc;
    double alpha, beta;
    int lower_tiles, klo_from_n, khi_from_m;
    int z_row0, z_rstride, total_rows;  // in tiles; total_rows <= 0 disables the check
};

template <bool B_NMAJOR>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_gemm(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int ti, tj;
    if (p.lower_tiles) {
        const int b = blockIdx.x;
        int i = (int)((sqrt(8.0 * b + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= b) ++i;
        while (i * (i + 1) / 2 > b) --i;
        ti = i;
        tj = b - i * (i + 1) / 2;
    } else {
        ti = blockIdx.x / p.nt;
        tj = blockIdx.x % p.nt;
    }
    const int z = blockIdx.y;
    if (p.total_rows > 0 && p.z_row0 + z * p.z_rstride + ti >= p.total_rows) return;
    const double* A = p.A + z * p.zA + (int64_t)ti * TILE * p.lda;
    const double* B = p.B + z * p.zB + (B_NMAJOR ? (int64_t)tj * TILE : (int64_t)tj * TILE * p.ldb);
    double* C = p.C + z * p.zC + (int64_t)ti * TILE * p.ldc + (int64_t)tj * TILE;
    int kb = 0, ke = p.kc;
    if (p.klo_from_n) kb = tj * (TILE / KC);
    if (p.khi_from_m) ke = min(ke, (ti + 1) * (TILE / KC));
    double acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.0;
    gemm_tile_loop<B_NMAJOR>(A, p.lda, B, p.ldb, kb, ke, smem, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int r = acc_row(lane, wr, mi);
#pragma unroll
        for (int nj = 0; nj < 8; ++nj) {
            const int c = acc_col(lane, wc, nj);
            double* dst = C + (int64_t)r * p.ldc + c;
            double v = p.alpha * acc[mi][nj];
            if (p.beta != 0.0) v += p.beta * *dst;
            *dst = v;
        }
    }
}

template __global__ void k_gemm<false>(GemmParams);
template __global__ void k_gemm<true>(GemmParams);

// ------------------------------------------------------------------------------------------------
// A3: alpha = W' (W (y - beta)).  W is lower-triangular row-major.
// ------------------------------------------------------------------------------------------------
__global__ void k_sub_mean(const double* __restrict__ y, double beta, int64_t N, double* __restrict__ r) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < N) r[i] = y[i] - beta;
}
// t[i] = sum_{j<=i} W[i][j] r[j] : one wave per row, lanes stride j (coalesced), fixed-order butterfly.
__global__ __launch_bounds__(256) void k_trimv(const double* __restrict__ W, int64_t ld, int64_t N,
                                               const double* __restrict__ r, double* __restrict__ t) {
    const int lane = threadIdx.x & 63;
    const int64_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    double s = 0.0;
    for (int64_t j = lane; j <= i; j += 64) s += W[i * ld + j] * r[j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) t[i] = s;
}
// a[j] = sum_{i>=j} W[i][j] t[i] : block = 64 columns x 4 row-strides, coalesced along j.
__global__ __launch_bounds__(256) void k_trimv_t(const double* __restrict__ W, int64_t ld, int64_t N,
                                                 const double* __restrict__ t, double* __restrict__ a) {
    __shared__ double red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t j = blockIdx.x * 64 + tx;
    double s = 0.0;
    if (j < N)
        for (int64_t i = j + ty; i < N; i += 4) s += W[i * ld + j] * t[i];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && j < N) a[j] = ((red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]));
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

}  // namespace bohip
