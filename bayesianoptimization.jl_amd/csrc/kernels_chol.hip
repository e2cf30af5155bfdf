// kernels_chol.hip -- the dataflow form of the blocked Cholesky factorisation (row A2 of SURVEY.md section 8; reference
// call sites: update!/append!/fit! at src/models/gp.jl:11-18, i.e. the ElasticPDMats / LAPACK potrf behind them).
//
// Why.  The launch-chained form (k_potf2_inv -> panel solve -> in-block update, one round per 128-block) is bound by its
// serial chain: ~66 us of single-workgroup diagonal-block work + two dependent launches (~2 x 20 us) per block, 24 blocks
// at N = 3000 = 2.75 ms for 9 GF.  Only three things are intrinsically serial: the pivot chain inside a diagonal block,
// the solve of the tile just below it, and the update of the NEXT diagonal block.  Here they are pipelined at 16-column
// granularity and never leave the chip's registers / LDS:
//
//   k_chol_chain   2 persistent workgroups ("row owners", rows r = w, w+2, ...).  The owner of row r
//                    during block r-1  holds tile (r, r-1) [128 x 128, registers] and tile (r, r) [lower, registers],
//                                      follows the 16-column panels the other owner publishes: row solve of its 128 rows
//                                      (substitution against the 16 x 16 pivot block), right-looking update of its
//                                      remaining columns, rank-16 update of (r, r);
//                    during block r    moves (r, r) -- fully updated -- into LDS and runs the pivot chain itself
//                                      (the body of k_potf2_inv without the inverse), publishing each finished panel of
//                                      L_rr to HBM/L2 with a release flag.
//                  The chain per block is therefore the pivot time + one flag hand-over, not pivot + 2 launches + 2 GEMMs.
//   k_chol_follow  one workgroup per tile (i, k), i >= k+2: the same panel follower without the diagonal tile.  No
//                  inverse W_kk is needed during the factorisation any more (k_inv128 builds all of them afterwards,
//                  in parallel, for the triangular inverse W = L^-1).
//   k_gemm_nt      trailing updates on the MFMA engine as before, gated by flags instead of host events: the row the
//                  chain needs next (k+2) is its own small launch whose completion counter the next owner waits on.
//
// Cross-workgroup protocol.  The chip has 8 XCDs with one L2 each; an agent-scope fence (__threadfence) is correct but is a
// whole-L2 write-back (release) / invalidate (acquire) of that XCD -- measured ~18 us per panel hand-over and it wrecks the
// L2 hit rate of the GEMMs running beside the chain.  So every datum that crosses workgroups INSIDE a running kernel is
// written and read with agent-scope (sc1) accesses, which go through to the memory-side coherence point, and ordering is
// a plain s_waitcnt:  producer  st_agent(data)... -> workgroup-scope release (vmcnt(0)) -> barrier -> one thread sets the
// flag (agent-scope store);  consumer  one thread spins on the flag -> barrier -> ld_agent(data).
// Every spin is bounded and honours a global abort word, so a logic error degrades into a wrong factor + error code,
// never into a hung GPU.
#include "gemm_core.h"

namespace bohip {

constexpr int CH_THREADS = 512;
constexpr int CH_PANELS = TILE / 16;          // 8 panels of 16 columns per 128-block
constexpr int WK_LPS = 16;                    // worker LDS: published panel LP[128][16]
constexpr int WK_XS = 17;                     //             solved rows   XB[128][17]
constexpr int WK_LDS_DOUBLES = 16 * (TILE + 2) + TILE * WK_XS + 16;   // LPt[16][130] | XB[128][17] | idl[16]
constexpr int CH_LDS_BYTES = (TILE * PF_LD + 2 * TILE) * 8;   // the potf2 image; the worker arrays alias its start
static_assert(WK_LDS_DOUBLES * 8 <= CH_LDS_BYTES, "worker arrays must fit inside the diagonal-block image");

struct CholFlags {
    unsigned* panel;      // [T * 8]   panel p of block k is published (columns of L_kk in L, 1/diag in idl_g)
    unsigned* solved;     // [T]       L(k+1, k) is complete in S (written by the owner of row k+1)
    unsigned* crit;       // [T]       counter: waves of the update launch for row k+2 of block k that have finished
    unsigned* abort;      // [1]       set on a spin time-out: every wait returns at once
    double* idl_g;        // [T * 128] 1 / L_ii, published with each panel
    unsigned crit_want;   // value of crit[k] when the whole row-(k+2) update launch of block k is in memory
    unsigned panel_want;  // value of panel[..] when every publishing wave has seen its stores land
};

__device__ __forceinline__ void flag_wait_ge(const unsigned* flag, unsigned want, unsigned* abort) {
    // one thread spins; callers put a barrier + __threadfence() behind it
    for (long it = 0;; ++it) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return;
        if (__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
        __builtin_amdgcn_s_sleep(4);
        if (it > 40000000L) {   // ~ seconds: something upstream never arrived
            __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
}
#ifndef BOHIP_CHOL_TRACE
#define BOHIP_CHOL_TRACE 0
#endif
#if BOHIP_CHOL_TRACE
__device__ unsigned long long g_chol_trace[4 * 1024];   // [0,1024): panel published; [1024,2048): owner saw panel; [2048,3072): owner finished panel; [3072..): block marks
#define CH_MARK(slot) g_chol_trace[(slot) & 4095] = wall_clock64()
#else
#define CH_MARK(slot) do {} while (0)
#endif
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void release_wg() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
__device__ __forceinline__ void flag_set(unsigned* flag, unsigned v) {
    __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the 36 lower 16 x 16 sub-blocks of a 128 x 128 tile, dealt to the two halves of a 512-thread workgroup ----------
__host__ __device__ constexpr int blk_bi(int idx) {
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= idx) ++bi;
    return bi;
}
__host__ __device__ constexpr int blk_bj(int idx) { return idx - blk_bi(idx) * (blk_bi(idx) + 1) / 2; }

// ------------------------------------------------------------------------------------------------------------------------
// Panel follower.  Thread t: row rr = t & 127 of tile (i, k), column group q = t >> 7 (32 columns, wave-uniform).
//   a[32]   this thread's piece of the tile row, in registers for the whole block
//   D1 (owners only): tile (i, i) lower, 18 sub-block elements per thread (half H = t >> 8 takes sub-blocks 2s + H)
// Per panel p: wait flag -> stage L_kk[:, 16p..16p+15] and 1/diag into LDS -> the threads that hold columns 16p..16p+15
// solve  x L16' = a  by substitution (x = 16 new entries of L(i, k)), publish x in LDS and to S -> everybody applies
// a[c'] -= x . L_kk[c'][16p..] to the columns still to come -> (owners) D1 -= x_i . x_j.
// ------------------------------------------------------------------------------------------------------------------------
// LDS of a follower: the published panel TRANSPOSED, LPt[m][i] = L_kk[i][16p + m], the panel rows XB[row][m] (first the
// not-yet-solved tile entries, then the solution x), and 1 / diag of the pivot block.
// Why this layout: an LDS read returns 64 lanes x its width whether or not the lanes share an address, so a thread that owns a
// whole row piece and reads every L_kk entry as a broadcast (the first version) moved 256 KB per wave and panel and the update
// took 10 us.  With an 8-row x 4-column register block per lane, a step of the update needs 8 + 4 operands for 32 FMAs.
constexpr int WK_LS = TILE + 2;   // row stride of LPt (even: 16-B aligned rows; +2 spreads the staging writes over the banks)

// x L16' = r  (L16 = pivot block), right-looking: once x[c] is known every later entry is updated at once, so the dependent
// chain is 16 x (mul, fma) instead of 120 fused multiply-adds in a row.
__device__ __forceinline__ void solve16(double (&r)[16], const double* LPt_p, const double* idl_s) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        r[c] *= idl_s[c];
        const double* col = LPt_p + c * WK_LS;   // L16[c'][c] for c' = 0..15 at col[c']
#pragma unroll
        for (int k = c + 1; k < 16; ++k) r[k] -= r[c] * col[k];
    }
}

// Thread t of a follower: wave w = t >> 6 owns columns 16w..16w+15 of tile (i, k); lane (r16 = lane & 15, cg = lane >> 4)
// holds rows r16 + 16 i (i < 8) x columns 16w + 4cg + e (e < 4):  a[4 i + e].
template <bool WITH_D1, int H>
__device__ __forceinline__ void follow_block(const double* __restrict__ Lmat, int64_t ld, double* __restrict__ S,
                                             int i_tile, int k_blk, const CholFlags& fl, double* wk, double (&a)[32],
                                             double (&d)[18]) {
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), r16 = lane & 15, cg = lane >> 4;
    double* LPt = wk;                        // [16][WK_LS]
    double* XB = wk + 16 * WK_LS;            // [128][17]
    double* idl_s = XB + TILE * WK_XS;       // [16]
    const double* Lkk = Lmat + (int64_t)k_blk * TILE * (ld + 1);
    const int ty = (t & 255) >> 4, tx = t & 15;
    for (int p = 0; p < CH_PANELS; ++p) {
        if (t == 0) { flag_wait_ge(fl.panel + k_blk * CH_PANELS + p, fl.panel_want, fl.abort); if (WITH_D1) CH_MARK(1024 + k_blk * CH_PANELS + p); }
        __syncthreads();
        {   // stage the published panel: rows 16p..127 of L_kk, columns 16p..16p+15 (4 threads per 128-B row segment)
            const int i = t >> 2, mq = t & 3;
            if (i >= 16 * p) {
                const double* src = Lkk + (int64_t)i * ld + 16 * p + 4 * mq;
                const double v0 = ld_agent(src), v1 = ld_agent(src + 1), v2 = ld_agent(src + 2), v3 = ld_agent(src + 3);
                double* dst = LPt + (4 * mq) * WK_LS + i;
                dst[0] = v0; dst[WK_LS] = v1; dst[2 * WK_LS] = v2; dst[3 * WK_LS] = v3;
            }
            if (t < 16) idl_s[t] = ld_agent(fl.idl_g + k_blk * TILE + 16 * p + t);
        }
        if (w == p) {   // the wave that holds the panel's columns hands them to the row solvers
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) XB[(r16 + 16 * i) * WK_XS + 4 * cg + e] = a[4 * i + e];
        }
        __syncthreads();
        if (WITH_D1 && t == 0 && k_blk == 1) CH_MARK(3584 + 8 * p + 0);
        if (t < TILE) {   // one thread per row: 16 new entries of L(i, k)
            double r[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) r[c] = XB[t * WK_XS + c];
            solve16(r, LPt + 16 * p, idl_s);
#pragma unroll
            for (int c = 0; c < 16; ++c) XB[t * WK_XS + c] = r[c];
            double* Srow = S + ((int64_t)i_tile * TILE + t) * ld + (int64_t)k_blk * TILE + 16 * p;
#pragma unroll
            for (int c = 0; c < 16; ++c) st_agent(Srow + c, r[c]);
        }
        __syncthreads();
        if (WITH_D1 && t == 0 && k_blk == 1) CH_MARK(3584 + 8 * p + 1);
        if (w > p) {   // wave-uniform: this wave's columns lie beyond the panel
            const double* lp = LPt + 16 * w + 4 * cg;
#pragma unroll 4
            for (int m = 0; m < 16; ++m) {
                const d2 l01 = *reinterpret_cast<const d2*>(lp + m * WK_LS), l23 = *reinterpret_cast<const d2*>(lp + m * WK_LS + 2);
                double xv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) xv[i] = XB[(r16 + 16 * i) * WK_XS + m];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    a[4 * i + 0] -= xv[i] * l01.x;
                    a[4 * i + 1] -= xv[i] * l01.y;
                    a[4 * i + 2] -= xv[i] * l23.x;
                    a[4 * i + 3] -= xv[i] * l23.y;
                }
            }
        }
        if (WITH_D1 && t == 511 && k_blk == 1) CH_MARK(3584 + 8 * p + 2);
        if constexpr (WITH_D1) {
            double acc[18];
#pragma unroll
            for (int s = 0; s < 18; ++s) acc[s] = 0.0;
#pragma unroll 4
            for (int m = 0; m < 16; ++m) {
                double li[8], lk[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    li[u] = XB[(16 * u + ty) * WK_XS + m];
                    lk[u] = XB[(16 * u + tx) * WK_XS + m];
                }
#pragma unroll
                for (int s = 0; s < 18; ++s) acc[s] += li[blk_bi(2 * s + H)] * lk[blk_bj(2 * s + H)];
            }
#pragma unroll
            for (int s = 0; s < 18; ++s) d[s] -= acc[s];
        }
        if (WITH_D1 && t == 511 && k_blk == 1) CH_MARK(3584 + 8 * p + 3);
        __syncthreads();   // LPt / XB are rewritten by the next panel
        if (WITH_D1 && t == 0) CH_MARK(2048 + k_blk * CH_PANELS + p);
    }
}

__device__ __forceinline__ void load_row_piece(const double* __restrict__ Lmat, int64_t ld, int i_tile, int k_blk,
                                               double (&a)[32]) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, r16 = lane & 15, cg = lane >> 4;
    const double* base = Lmat + ((int64_t)i_tile * TILE + r16) * ld + (int64_t)k_blk * TILE + 16 * w + 4 * cg;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[4 * i + e] = ld_agent(base + (int64_t)16 * i * ld + e);
}

// ------------------------------------------------------------------------------------------------------------------------
// Pivot role: the Cholesky part of k_potf2_inv (same panels, same look-ahead of the next 16 x 16 pivot block beside the
// trailing update) on the image `a` already in LDS, run by threads 0..255 of the 512-thread owner (the others only keep
// the barriers company).  Instead of one "L out" phase at the end, every finished 16-column panel goes to HBM/L2 at once
// and its flag is raised: that is what the followers are waiting for.
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void publish_panel(double* __restrict__ Lblk, int64_t ld, const double* a, const double* dl,
                                              const double* idl, double* idl_out, int P, int tid) {
    // columns P..P+15 of L_kk, rows P..127: thread (i = tid & 127, h = tid >> 7) writes 8 columns = 64 B
    const int i = tid & 127, h = tid >> 7;
    if (i >= P) {
        double* dst = Lblk + (int64_t)i * ld + P + 8 * h;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int c0 = P + 8 * h + c;
            st_agent(dst + c, (i > c0) ? a[c0 * PF_LD + i] : (i == c0 ? dl[i] : 0.0));
        }
    }
    if (tid < 16) st_agent(idl_out + P + tid, idl[P + tid]);
}

__device__ __forceinline__ void pivot_block(double* a, double* dl, double* idl, double* __restrict__ Lblk, int64_t ld,
                                            int k_blk, const CholFlags& fl, int* info, int row0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool act = tid < PF_THREADS;
    unsigned* pflag = fl.panel + k_blk * CH_PANELS;
    double* idl_out = fl.idl_g + k_blk * TILE;
    if (wave == 0) factor16(a, dl, idl, 0, lane, info, row0);
    __syncthreads();
    for (int jb = 0; jb < CH_PANELS; ++jb) {
        const int P = 16 * jb;
        const int base = P + 16, m = TILE - base;
        if (act && tid < m) {  // row solves below the pivot block
            const int i = base + tid;
            double x[16], rw[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) rw[c] = a[i * PF_LD + P + c];
#pragma unroll
            for (int c = 0; c < 16; ++c) {   // right-looking: 16 x (mul, fma) on the dependent chain instead of 120 fmas
                x[c] = rw[c] * idl[P + c];
#pragma unroll
                for (int k = c + 1; k < 16; ++k) rw[k] -= x[c] * a[(P + c) * PF_LD + P + k];
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) a[(P + c) * PF_LD + i] = x[c];
        }
        __syncthreads();
        // Panel jb is final.  The four waves that take no part in the factorisation publish it -- their stores and the wait for
        // them to land stay off the pivot chain: they are issued here, awaited behind the next barrier, and each wave then
        // adds 1 to the panel's flag (followers wait for 4).
        if (!act) publish_panel(Lblk, ld, a, dl, idl, idl_out, P, tid - PF_THREADS);
        if (m > 0 && act) {   // rank-16 update of the next pivot block
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
        if (!act) {
            release_wg();   // s_waitcnt vmcnt(0): this wave's part of the panel has left the CU
            if (lane == 0) { atomicAdd(pflag + jb, 1u); if (wave == 4) CH_MARK(k_blk * CH_PANELS + jb); }
        }
        if (m == 0) break;
        if (wave == 0) {
            factor16(a, dl, idl, base, lane, info, row0);
        } else if (act) {
            const int u = tid - 64;
            trailing_dispatch<true, -1>(a, P, m >> 4, 4 + (u >> 4), u & 15);
            const int ty0 = (u >> 4) & 3, tx0 = u & 15;
            if (wave == 1) trailing_dispatch<true, 0>(a, P, m >> 4, ty0, tx0);
            else if (wave == 2) trailing_dispatch<true, 1>(a, P, m >> 4, ty0, tx0);
            else trailing_dispatch<true, 2>(a, P, m >> 4, ty0, tx0);
        }
        __syncthreads();
    }
}

// diagonal tile (r, r): HBM -> registers (owner, follower role) / registers -> LDS image (owner, pivot role)
template <int H>
__device__ __forceinline__ void d1_load(const double* __restrict__ Lmat, int64_t ld, int r, double (&d)[18]) {
    const int t = threadIdx.x, ty = (t & 255) >> 4, tx = t & 15;
    const double* blk = Lmat + (int64_t)r * TILE * (ld + 1);
#pragma unroll
    for (int s = 0; s < 18; ++s) {
        const int bi = blk_bi(2 * s + H), bj = blk_bj(2 * s + H);
        d[s] = ld_agent(blk + (int64_t)(16 * bi + ty) * ld + 16 * bj + tx);
    }
}
template <int H>
__device__ __forceinline__ void d1_to_image(double* a, const double (&d)[18]) {
    const int t = threadIdx.x, ty = (t & 255) >> 4, tx = t & 15;
#pragma unroll
    for (int s = 0; s < 18; ++s) {
        const int bi = blk_bi(2 * s + H), bj = blk_bj(2 * s + H);
        const int i = 16 * bi + ty, j = 16 * bj + tx;
        if (bi != bj) {
            a[i * PF_LD + j] = d[s];
            a[j * PF_LD + i] = 0.0;       // the strict upper triangle is workspace and starts at zero
        } else {
            a[i * PF_LD + j] = (j <= i) ? d[s] : 0.0;
        }
    }
}

template <int H>
__device__ __forceinline__ void chain_owner(double* __restrict__ Lmat, int64_t ld, double* __restrict__ S, int T,
                                            const CholFlags& fl, int* info, double* sm, int w) {
    double* a = sm;
    double* dl = sm + TILE * PF_LD;
    double* idl = dl + TILE;
    const int tid = threadIdx.x;
    double ar[32], d[18];
    for (int r = w; r < T; r += 2) {
        if (r == 0) {   // tile (0, 0) straight from HBM into the image
            const double* Lblk = Lmat;
            for (int e = tid; e < TILE * TILE; e += CH_THREADS) {
                const int i = e >> 7, j = e & 127;
                a[i * PF_LD + j] = (j <= i) ? Lblk[(int64_t)i * ld + j] : 0.0;
            }
            __syncthreads();
        } else {
            // follower role during block r-1: tiles (r, r-1) and (r, r) must carry every update from blocks <= r-2
            if (r >= 2) {
                if (tid == 0) flag_wait_ge(fl.crit + (r - 2), fl.crit_want, fl.abort);
                __syncthreads();
            }
            load_row_piece(Lmat, ld, r, r - 1, ar);
            d1_load<H>(Lmat, ld, r, d);
            follow_block<true, H>(Lmat, ld, S, r, r - 1, fl, sm, ar, d);
            release_wg();
            __syncthreads();
            if (tid == 0) flag_set(fl.solved + (r - 1), 1u);   // L(r, r-1) is complete in S
            d1_to_image<H>(a, d);
            __syncthreads();
        }
        if (tid == 0) CH_MARK(3072 + 2 * r);
        pivot_block(a, dl, idl, Lmat + (int64_t)r * TILE * (ld + 1), ld, r, fl, info, r * TILE);
        if (tid == 0) CH_MARK(3072 + 2 * r + 1);
        __syncthreads();
    }
}

__global__ __launch_bounds__(CH_THREADS, 1) void k_chol_chain(double* __restrict__ Lmat, int64_t ld, double* __restrict__ S,
                                                           int T, CholFlags fl, int* __restrict__ info) {
    extern __shared__ double sm[];
    const int w = blockIdx.x;   // 0 / 1: owns rows w, w + 2, ...
    if ((threadIdx.x >> 8) == 0) chain_owner<0>(Lmat, ld, S, T, fl, info, sm, w);
    else chain_owner<1>(Lmat, ld, S, T, fl, info, sm, w);
}

// one workgroup per tile (i, k), i = k + 2 + blockIdx.x: solves L(i, k) panel by panel as block k's pivot chain runs
__global__ __launch_bounds__(CH_THREADS, 1) void k_chol_follow(const double* __restrict__ Lmat, int64_t ld,
                                                            double* __restrict__ S, int k_blk, CholFlags fl) {
    extern __shared__ double sm[];
    const int i_tile = k_blk + 2 + blockIdx.x;
    double ar[32], d[18];
    load_row_piece(Lmat, ld, i_tile, k_blk, ar);
    follow_block<false, 0>(Lmat, ld, S, i_tile, k_blk, fl, sm, ar, d);
}

// ------------------------------------------------------------------------------------------------------------------------
// W_kk = L_kk^-1 for every diagonal block at once (the seeds of the recursive triangular inverse): the inverse half of
// k_potf2_inv on an image rebuilt from the finished factor.  One workgroup per block.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PF_THREADS) void k_inv128(const double* __restrict__ Lmat, int64_t ld, double* __restrict__ W,
                                                    double* __restrict__ WT, int64_t ldw) {
    extern __shared__ double sm[];
    double* a = sm;
    double* idl = sm + TILE * PF_LD + TILE;
    const int tid = threadIdx.x;
    const int64_t off = (int64_t)blockIdx.x * TILE;
    const double* Lblk = Lmat + off * (ld + 1);
    {   // mirror image: a[c][r] = L[r][c] for c < r, the lower triangle is workspace for W
        const int j = tid & 127, ih = tid >> 7;
#pragma unroll 1
        for (int i0 = 0; i0 < TILE; i0 += 32) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = Lblk[(int64_t)(i0 + 2 * u + ih) * ld + j];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = i0 + 2 * u + ih;
                if (j < i) a[j * PF_LD + i] = v[u];
                else if (j == i) idl[i] = 1.0 / v[u];
            }
        }
    }
    __syncthreads();
    inverse_phase(a, idl, tid, W + off * (ldw + 1), WT + off * (ldw + 1), ldw);
}

}  // namespace bohip
