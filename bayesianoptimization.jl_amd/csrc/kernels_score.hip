// kernels_score.hip -- the acquisition hot path (rows A4-A7, A9 of SURVEY.md section 8):
//   k_kstar        cross-covariance K*' chunk [Rc][Npad] (candidate-major, contraction index contiguous)
//   k_trigemm_sq   V = W K* on FP64 MFMA with the sum-of-squares epilogue fused: V is never stored.
//                  Row N of W carries alpha', so the same contraction also yields mu - beta.
//   k_score        sigma^2 = max(s_f^2 - sum v^2, 0), the reference's acquisition formulas verbatim,
//                  per-block arg-max;  k_argmax_final reduces to ONE 16-byte record.
//   k_thompson     S x R independent posterior draws with a counter-based normal generator.
// Reference call sites replaced: mean_var / predict_f (src/models/gp.jl:2-8), the functors of
// src/acquisitionfunctions.jl:24-27,47-50,96,111,141 with src/utils.jl:48-49, and the arg-max of
// acquire_max (src/acquisition.jl:54-68).
#include "gemm_core.h"

namespace bohip {

__device__ __forceinline__ double cov_from_r_fast(int kern, double sigma2, double r) {
    if (kern == KERN_MAT52ARD) {
        const double R = sqrt(r), s = sqrt(5.0) * R;
        return sigma2 * (1.0 + s + 5.0 / 3.0 * r) * exp(-s);
    }
    return sigma2 * exp(-0.5 * r);
}

// ------------------------------------------------------------------------------------------------
// K*': thread = observation j (its coordinates live in registers), loop over the block's
// candidates whose coordinates are wave-uniform (scalar loads).  Stores are coalesced along j.
// KsT[r][j] = k(x_j, x*_r) for j < N, 0 for N <= j < Npad.  Xs is [R][d] (d contiguous).
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void k_kstar(const double* __restrict__ X, int64_t N, int64_t Npad,
                                               const double* __restrict__ Xs, int64_t r_begin, int64_t r_end,
                                               KernelHyper hp, double* __restrict__ KsT, int64_t ldk, int rb) {
    // the block's rb (<= 16) candidates are staged in LDS by one coalesced load (see k_build_cov: per-dimension scalar
    // loads behind `if (k < d)` made this kernel latency-bound); dimensions d <= k < DT carry zero weight
    __shared__ double xs_l[16 * DT];
    const int d = hp.d;
    const int64_t j = blockIdx.x * 256 + threadIdx.x;
    const int64_t r0 = r_begin + (int64_t)blockIdx.y * rb;
    const int64_t r1 = min(r0 + rb, r_end);
    for (int t = threadIdx.x; t < rb * DT; t += 256) {
        const int64_t r = r0 + t / DT;
        const int k = t % DT;
        xs_l[t] = (r < r1 && k < d) ? Xs[r * d + k] : 0.0;
    }
    double xj[DT], w[DT];
#pragma unroll
    for (int k = 0; k < DT; ++k) {
        xj[k] = (k < d && j < N) ? X[j * d + k] : 0.0;
        w[k] = k < d ? hp.il2[k] : 0.0;
    }
    __syncthreads();
    if (j >= Npad) return;
    const int nc = (int)(r1 - r0);
    for (int c = 0; c < nc; ++c) {
        double rr = 0.0;
#pragma unroll
        for (int k = 0; k < DT; ++k) {
            const double t = xj[k] - xs_l[c * DT + k];
            rr += w[k] * (t * t);
        }
        const double v = (j < N) ? cov_from_r_fast(hp.kern, hp.sigma2, rr) : 0.0;
        KsT[(r0 + c - r_begin) * ldk + j] = v;   // plain stores: non-temporal ones evict K*' from L2/MALL and k_trigemm_sq
                                                   // then runs at 0.70 instead of 0.635 ms (measured)
    }
}

// ------------------------------------------------------------------------------------------------
// Full posterior covariance of R candidates (predict_f(gp, X; full_cov = true), the input of the reference's joint
// draw myrand(model, X::Matrix), src/models/gp.jl:7):  cov[r][s] = k(x*_r, x*_s) - (V'V)[r][s].
// VV holds the LOWER triangle of V'V (k_gemm_nt on the stored V'); both halves of cov are written from it so the
// result is exactly symmetric.  Thread = column s, block walks 16 rows.
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void k_post_cov(const double* __restrict__ Xs, int64_t R, KernelHyper hp,
                                                  const double* __restrict__ VV, int64_t ldv,
                                                  double* __restrict__ cov, int64_t ldc) {
    const int d = hp.d;
    const int64_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= R) return;
    double xs[DT];
#pragma unroll
    for (int k = 0; k < DT; ++k) xs[k] = (k < d) ? Xs[s * d + k] : 0.0;
    const int64_t r0 = (int64_t)blockIdx.y * 16, r1 = min(R, r0 + 16);
    for (int64_t r = r0; r < r1; ++r) {
        double rr = 0.0;
#pragma unroll
        for (int k = 0; k < DT; ++k)
            if (k < d) {
                const double t = Xs[r * d + k] - xs[k];
                rr += hp.il2[k] * (t * t);
            }
        const double vv = (r >= s) ? VV[r * ldv + s] : VV[s * ldv + r];
        cov[r * ldc + s] = cov_from_r_fast(hp.kern, hp.sigma2, rr) - vv;
    }
}

// ------------------------------------------------------------------------------------------------
// V = W K*, fused epilogue.  Tile (rt, ct): rows [128 rt, 128 rt + 128) of W against candidates
// [64 ct, 64 ct + 64) of the chunk; W is lower-triangular so the contraction stops at k = 128 (rt + 1).
// Job length is set by that K extent, so the candidate tile is only 64 wide: with 128-wide tiles the longest
// job (24 units at N=3000) exceeds the average load per workgroup slot (18.75) and list scheduling cannot
// balance.  Work per tile grows with rt, so tiles are issued heaviest-first; blocks are dealt to XCDs
// (block b runs on XCD b % 8) so that each XCD owns a fixed subset of candidate tiles and walks the
// row tiles together: the W row-tile stream is then shared through that XCD's L2.
// Output: q_part[rt][r] = sum over the tile's rows of v^2 (fixed summation order -> deterministic),
//         mu_raw[r] = alpha' k*_r taken from the row of W that stores alpha (alpha_row).
// ------------------------------------------------------------------------------------------------
struct AcqParams {
    int acq;
    double p0, p1;
};
// Fused finish of k_trigemm_sq (large batches, value-only scoring): the workgroup that delivers the LAST row tile of a
// candidate tile turns the T partial sums into sigma^2, the acquisition value and the tile's arg-max; the one that
// completes the last candidate tile reduces the per-tile records to the 16-byte result.  Replaces k_score +
// k_argmax_final (two launches, ~16 us of a 0.69 ms step).  Same summation order as k_score -> bit-identical scores.
// Cross-workgroup data (q_part, mu_raw, tile records) moves with agent-scope accesses; the counters are left at zero.
struct FuseParams {
    unsigned* tile_cnt;      // [candidate tiles of the whole batch]  arrivals per tile (nullptr: no fused finish)
    unsigned* total_cnt;     // [1] finished candidate tiles
    Best* tile_best;         // [candidate tiles]
    int tiles_total;         // candidate tiles of the whole batch (all chunks)
    int T;                   // row tiles (the finish adds two partial sums per tile)
    int64_t R_total;         // candidates of the whole batch
    double sigma2, beta;
    AcqParams ap;
    double *mu_out, *var_out, *score_out;   // nullable, indexed by the global candidate number
    Best* best_out;          // nullable: the batch's arg-max record
    long long best_off;      // added to the winner's index (sharded scoring)
    unsigned long long* clk; // nullable: [2] core-clock and 100 MHz wall-clock ticks summed over a sample of workgroups (timing runs)
};
__device__ void trigemm_fused_finish(const FuseParams& fz, int tile_g, int T, const double* q_part, int64_t ldq,
                                     const double* mu_raw);

// BOHIP_TRACE (tools only, default 0 in gemm_core.h): per-workgroup start/end clocks of k_trigemm_sq (tools/trace_trigemm.py)
#if BOHIP_TRACE
__device__ unsigned long long g_trace[6 * 8192];   // per workgroup: wall start, wall end, HW_ID, XCC_ID, core-clock start, core-clock end
#endif
// ---- row pieces ------------------------------------------------------------------------------------------------------
// A job is one ROW PIECE of W against one 64-wide candidate tile.  A piece is a whole 128-row tile (job length = its K extent:
// rt + 1 units) or one 64-row HALF of a tile, run in the loop's half mode (both wave rows work on the 64 rows, even / odd
// members of every contraction-index pair, partial sums added at the end): half the rows, so about half the time.
// Why halves: with whole tiles only the launch ends raggedly -- the jobs in flight when the queue runs dry are 1 to 8 units
// long (two co-resident workgroups: up to 160 us), CUs finish over an 80 us window and idle 5 % of the launch on average
// (tools/trace_trigemm.py).  With the shortest third of the row tiles issued as halves the last jobs are a few units at most
// (tools/sim_trigemm_tail.py: idle tail 3.2 % -> 0.4 % in the model).  The HOST lists the pieces heaviest-first
// (trigemm_pieces in bohip.hip; the list depends on the number of row tiles and the position of the alpha row ONLY, never on
// the batch: scores stay batch- and shard-independent) and the kernel reads its piece from that table.
// Encoding: piece = rt | mode << 16;  mode 0 whole tile, 1 rows 0..63, 2 rows 64..127, 3 rows 0..63 of a tile whose rows 64..127
// are all padding (the last tile when the alpha row sits in its upper half): no sibling piece exists.
// q_part holds TWO partial sums per row tile (rows 0..63 and 64..127, [2 rt + h][r]); the finish adds (q[2t] + q[2t+1]) tile
// by tile -- for a whole-tile job exactly the sum it used to store, so a list without halves reproduces the old bits.
constexpr int PIECE_WHOLE = 0, PIECE_UPPER = 1, PIECE_LOWER = 2, PIECE_UPPER_SOLO = 3;

// one job: row piece (rt, mode) of W against candidate tile ct of the chunk (all arguments wave-uniform)
template <int KS>
__device__ __forceinline__ void trigemm_job(int rt, int mode, int ct, const double* __restrict__ W, int64_t ldw,
                                            const double* __restrict__ KsT, int64_t ldk, int NP, int64_t alpha_row,
                                            double* __restrict__ q_part, int64_t ldq, double* __restrict__ mu_raw, int64_t r_off,
                                            double* __restrict__ VT, int64_t ldv, const FuseParams& fz, double* smem, int tid) {
    constexpr int NJ = 4, CW = CTILE;
    double acc[8][NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = 0.0;
    const int h = mode == PIECE_LOWER ? 1 : 0;                 // which half the piece starts at
    const int64_t row_base = (int64_t)rt * TILE + 64 * h;
    // rows past alpha' are padding; a half piece has at most 64 live rows (-> the loop's half mode)
    const int active_rows = (int)min((int64_t)(mode == PIECE_WHOLE ? TILE : TILE / 2), alpha_row + 1 - row_base);
    const int tri_kc = rt * (TILE / KC) + 4 * h;              // first chunk of the piece's triangular block
    const int kc_end = mode == PIECE_WHOLE ? (rt + 1) * (TILE / KC) : tri_kc + 4;
    if constexpr (KS == 2)
        gemm_tile_loop_glds3_ks<NJ, BOHIP_ABL, true>(W + row_base * ldw, ldw, KsT + (int64_t)ct * CW * ldk, ldk, 0,
                                    kc_end, smem, acc, active_rows, tri_kc, tid);
    else
        gemm_tile_loop_glds3<NJ>(W + row_base * ldw, ldw, KsT + (int64_t)ct * CW * ldk, ldk, 0, kc_end, smem, acc, active_rows);
    const int lane = tid & 63, wave = tid >> 6, wr = (wave & 3) >> 1, wc = wave & 1;
    __syncthreads();     // (the raw-barrier loops end on s_barrier; make the reuse of smem below explicit)
    double* red = smem;  // [2][CW]
    // half mode leaves the piece's 64 rows in wave row 0; wave row 1 holds zeros that belong to NO row of this piece
    const bool rows_mine = mode == PIECE_WHOLE || wr == 0;
    if (wave < 4) {
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        double s = 0.0;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int64_t grow = row_base + acc_row(lane, wr, mi);
            const double v = acc[mi][nj];
            if (grow == alpha_row && rows_mine) {
                __hip_atomic_store(mu_raw + r_off + (int64_t)ct * CW + acc_col<NJ>(lane, wc, nj), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                s += v * v;
            }
            if (VT != nullptr && grow < alpha_row && rows_mine)
                VT[((int64_t)ct * CW + acc_col<NJ>(lane, wc, nj)) * ldv + grow] = v;
        }
        s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (lane < 8) red[wr * CW + wc * 8 * NJ + 8 * nj + lane] = s;
    }
    }
    __syncthreads();
    if (tid < 2 * CW) {   // thread t < 64: the sum over the piece's first 64 rows; t >= 64: over rows 64..127 of a whole tile
        const int hh = tid >> 6, c = tid & (CW - 1);
        double* dst = q_part + (int64_t)(2 * rt + (mode == PIECE_WHOLE ? hh : h)) * ldq + r_off + (int64_t)ct * CW + c;
        if (mode == PIECE_WHOLE || hh == 0) __hip_atomic_store(dst, red[hh * CW + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (mode == PIECE_UPPER_SOLO) __hip_atomic_store(dst + ldq, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // no sibling: its slot is zero
    }
    if (fz.tile_cnt != nullptr) {
        __shared__ int s_last;
        const int tile_g = (int)(r_off / CW) + ct;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the agent-scope stores above have landed (a workgroup-scope fence emits no such wait)
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(fz.tile_cnt + tile_g, 1u) == (unsigned)(NP - 1);
        __syncthreads();
        if (s_last && tid < 64) trigemm_fused_finish(fz, tile_g, fz.T, q_part, ldq, mu_raw);
    }
}

template <int KS>  // 1: 4 waves; 2: 8 waves, contraction index halved inside the workgroup (default)
__global__ __launch_bounds__(KS * GEMM_THREADS, 2 * KS) void k_trigemm_sq(const double* __restrict__ W, int64_t ldw,
                                                                const double* __restrict__ KsT, int64_t ldk,
                                                                const int* __restrict__ pieces, int NP, int CT, int64_t alpha_row,
                                                                double* __restrict__ q_part, int64_t ldq,
                                                                double* __restrict__ mu_raw, int64_t r_off,
                                                                double* __restrict__ VT, int64_t ldv, FuseParams fz) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
#if BOHIP_TRACE
    const unsigned long long t_start = wall_clock64(), c_start = clock64();
    struct TraceEnd {
        unsigned long long t0, c0;
        __device__ ~TraceEnd() {
            if (threadIdx.x == 0) {
                unsigned hw, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                unsigned long long* t = g_trace + 6 * (size_t)blockIdx.x;
                t[0] = t0; t[1] = wall_clock64(); t[2] = hw; t[3] = xcc; t[4] = c0; t[5] = clock64();
            }
        }
    } trace_end{t_start, c_start};
#endif
    // blocks are dealt to XCDs round-robin (block b runs on XCD b % 8): an XCD owns the candidate tiles ct = xcd (mod 8) and
    // walks the row pieces together, heaviest first
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int n_local = (CT + 7) >> 3;
    const int ct = xcd + 8 * (slot % n_local);
    if (ct >= CT) return;
    const int piece = __builtin_amdgcn_readfirstlane(pieces[slot / n_local]);   // (uniform: a scalar load)
    // timing runs (bohip_gp_enable_timing): every 33rd workgroup reports how many core-clock cycles and 100 MHz ticks it lived --
    // their ratio is the clock the chip sustained UNDER THIS KERNEL (MI355X clocks to its power budget: ~2.0 of 2.4 GHz here)
    __shared__ unsigned long long s_clk[2];
    if (fz.clk != nullptr && blockIdx.x % 33 == 0 && threadIdx.x == 0) { s_clk[0] = clock64(); s_clk[1] = wall_clock64(); }
    trigemm_job<KS>(piece & 0xffff, piece >> 16, ct, W, ldw, KsT, ldk, NP, alpha_row, q_part, ldq, mu_raw, r_off, VT, ldv, fz, smem,
                    (int)threadIdx.x);
    if (fz.clk != nullptr && blockIdx.x % 33 == 0 && threadIdx.x == 0) {
        atomicAdd(fz.clk, (unsigned long long)clock64() - s_clk[0]);
        atomicAdd(fz.clk + 1, (unsigned long long)wall_clock64() - s_clk[1]);
    }
}

// ---- split-K path for batches with too few candidate tiles to fill the chip (a few hundred candidates) -------------
// With one 64-wide candidate tile column per 64 candidates the longest job (the last row tile, K = N) runs alone on one
// CU: ~9 us per 128 of K, 215 us at N = 3000 whatever R.  Here the contraction index is cut into slices of kz chunks,
// k_gemm_nt computes the partial V' = K*' W' tiles of every slice in one batched launch (planes PT[s][r][i]), and this
// kernel adds the slices of a candidate in slice order, squares, and reduces: q[r], mu_raw[r], optionally V' itself.
// One workgroup per candidate; fixed summation order -> independent of the batch.
__global__ __launch_bounds__(256) void k_split_combine_v(const double* __restrict__ PT, int64_t plane, int64_t ldp, int kz,
                                                         int64_t N, double* __restrict__ q, double* __restrict__ mu_raw,
                                                         double* __restrict__ VT, int64_t ldv) {
    __shared__ double red[256];
    const int r = blockIdx.x;
    const double* base = PT + (int64_t)r * ldp;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i <= N; i += 256) {
        const int nsl = (int)(((i >> 7) + 1) * (TILE / KC) + kz - 1) / kz;   // slices that reach row tile i / 128
        double v = 0.0;
        for (int sl = 0; sl < nsl; ++sl) v += base[(int64_t)sl * plane + i];
        if (i < N) {
            s += v * v;
            if (VT) VT[(int64_t)r * ldv + i] = v;
        } else {
            mu_raw[r] = v;   // the alpha row
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) q[r] = red[0];
}
// U'[r][j] = sum over the slices that reach column tile j / 128 (i >= j) of the partial planes
__global__ __launch_bounds__(256) void k_split_combine_u(const double* __restrict__ PU, int64_t plane, int64_t ldp, int kz,
                                                         int nslices, int64_t N, double* __restrict__ UT, int64_t ldu) {
    const int r = blockIdx.x;
    const double* base = PU + (int64_t)r * ldp;
    for (int64_t j = threadIdx.x; j < N; j += 256) {
        const int s0 = (int)((j >> 7) * (TILE / KC)) / kz;   // first slice whose range ends beyond the start of the column tile
        double v = 0.0;
        for (int sl = s0; sl < nslices; ++sl) v += base[(int64_t)sl * plane + j];
        UT[(int64_t)r * ldu + j] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Acquisition functors -- verbatim operation order of the reference, contraction OFF (Julia never
// fuses a*b+c).  normal_pdf / normal_cdf: src/utils.jl:48-49.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double acq_eval(const AcqParams& a, double mu, double s2) {
#pragma clang fp contract(off)
    switch (a.acq) {
        case ACQ_EI: {  // src/acquisitionfunctions.jl:47-50  (D*Phi + sqrt(s2)*pdf  ==  D*Phi(z) + phi(z))
            const double tau = a.p0;
            if (s2 == 0.0) return mu > tau ? mu - tau : 0.0;
            const double D = mu - tau;
            const double cdf = 1.0 / 2.0 * (1.0 + erf(D / sqrt(2.0 * s2)));
            const double pdf = 1.0 / sqrt(2.0 * M_PI * s2) * exp(-(D * D) / (2.0 * s2));
            return D * cdf + sqrt(s2) * pdf;
        }
        case ACQ_PI: {  // :24-27
            const double tau = a.p0;
            if (s2 == 0.0) return mu > tau ? 1.0 : 0.0;
            return 1.0 / 2.0 * (1.0 + erf((mu - tau) / sqrt(2.0 * s2)));
        }
        case ACQ_UCB:  // :96
            return mu + a.p0 * sqrt(s2);
        case ACQ_MI:  // :141
            return mu + a.p0 * (sqrt(s2 + a.p1) - sqrt(a.p1));
        default:  // MaxMean :111
            return mu;
    }
}

// (value desc, index asc); NaN never wins (reference: `f > maxf` is false for NaN).
__device__ __forceinline__ bool better(double v, long long i, double bv, long long bi) {
    if (i < 0) return false;
    if (bi < 0) return v > -INFINITY;
    return v > bv || (v == bv && i < bi);
}
__device__ __forceinline__ void block_argmax(double& v, long long& i, Best* sh) {
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o);
        const long long oi = __shfl_xor(i, o);
        if (better(ov, oi, v, i)) { v = ov; i = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[wave].val = v; sh[wave].idx = i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (better(sh[w].val, sh[w].idx, v, i)) { v = sh[w].val; i = sh[w].idx; }
    }
}

// one wave: the 64 candidates of tile tile_g (see FuseParams)
__device__ void trigemm_fused_finish(const FuseParams& fz, int tile_g, int T, const double* q_part, int64_t ldq,
                                     const double* mu_raw) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)tile_g * CTILE + lane;
    double v = -INFINITY;
    long long idx = -1;
    if (r < fz.R_total) {
        double q = 0.0;
        for (int t = 0; t < T; ++t)   // (upper half + lower half) tile by tile: what a whole-tile job used to store as ONE number
            q += __hip_atomic_load(q_part + (int64_t)(2 * t) * ldq + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                 __hip_atomic_load(q_part + (int64_t)(2 * t + 1) * ldq + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double s2 = fz.sigma2 - q;
        if (s2 < 0.0) s2 = 0.0;  // predict_f: max(sigma2, 0)
        const double mu = fz.beta + __hip_atomic_load(mu_raw + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (fz.mu_out) fz.mu_out[r] = mu;
        if (fz.var_out) fz.var_out[r] = s2;
        const double f = acq_eval(fz.ap, mu, s2);
        if (fz.score_out) fz.score_out[r] = f;
        if (f > -INFINITY) { v = f; idx = r; }  // false for NaN and -Inf
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o);
        const long long oi = __shfl_xor(idx, o);
        if (better(ov, oi, v, idx)) { v = ov; idx = oi; }
    }
    int last = 0;
    if (lane == 0) {
        fz.tile_cnt[tile_g] = 0u;   // ready for the next call
        if (fz.best_out) {
            __hip_atomic_store(&fz.tile_best[tile_g].val, idx >= 0 ? v : -INFINITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&fz.tile_best[tile_g].idx, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile record stored before it is counted
            last = atomicAdd(fz.total_cnt, 1u) == (unsigned)(fz.tiles_total - 1);
        }
    }
    last = __shfl(last, 0);
    if (!last) return;
    v = -INFINITY;
    idx = -1;
    for (int i = lane; i < fz.tiles_total; i += 64) {
        const double bv = __hip_atomic_load(&fz.tile_best[i].val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long bi = __hip_atomic_load(&fz.tile_best[i].idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (better(bv, bi, v, idx)) { v = bv; idx = bi; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o);
        const long long oi = __shfl_xor(idx, o);
        if (better(ov, oi, v, idx)) { v = ov; idx = oi; }
    }
    if (lane == 0) {
        *fz.total_cnt = 0u;
        fz.best_out->val = idx >= 0 ? v : -INFINITY;
        fz.best_out->idx = idx >= 0 ? idx + fz.best_off : -1;
    }
}

// T > 0: T row tiles with TWO partial sums each ([2 t], [2 t + 1]: k_trigemm_sq's layout); T = 0: q_part[r] is the finished sum
// (row-wise and split-K paths)
__global__ __launch_bounds__(256) void k_score(const double* __restrict__ q_part, int64_t ldq, int T,
                                               const double* __restrict__ mu_raw, int64_t R, double sigma2,
                                               double beta, AcqParams ap, double* __restrict__ mu_out,
                                               double* __restrict__ var_out, double* __restrict__ score_out,
                                               Best* __restrict__ block_best) {
#pragma clang fp contract(off)
    __shared__ Best sh[4];
    const int64_t r = blockIdx.x * 256 + threadIdx.x;
    double v = -INFINITY;
    long long idx = -1;
    if (r < R) {
        double q = 0.0;
        for (int t = 0; t < T; ++t) q += q_part[(int64_t)(2 * t) * ldq + r] + q_part[(int64_t)(2 * t + 1) * ldq + r];
        if (T == 0) q = q_part[r];
        double s2 = sigma2 - q;
        if (s2 < 0.0) s2 = 0.0;  // predict_f: max(sigma2, 0)
        const double mu = beta + mu_raw[r];
        if (mu_out) mu_out[r] = mu;
        if (var_out) var_out[r] = s2;
        const double f = acq_eval(ap, mu, s2);
        if (score_out) score_out[r] = f;
        if (f > -INFINITY) { v = f; idx = r; }  // false for NaN and -Inf
    }
    if (block_best) {
        block_argmax(v, idx, sh);
        if (threadIdx.x == 0) { block_best[blockIdx.x].val = idx >= 0 ? v : -INFINITY; block_best[blockIdx.x].idx = idx; }
    }
}

// Small batches (R <= 256, row-wise posterior): one workgroup per candidate r adds q[r] = sum_j V'[r][j]^2 in a fixed
// order and takes mu_raw[r] from the alpha row; the LAST workgroup to finish then does what k_score + k_argmax_final do
// for a large batch (same formulas, same operation order) for all R candidates -- two launches less per call, which
// at this size are a fifth of its latency.  The arrival counter is left at zero.
__global__ __launch_bounds__(256) void k_small_finish(const double* __restrict__ VT, int64_t ldv, int64_t N, int R,
                                                      double* __restrict__ q, double* __restrict__ mu_raw,
                                                      unsigned* __restrict__ counter, double sigma2, double beta,
                                                      AcqParams ap, double* __restrict__ mu_out,
                                                      double* __restrict__ var_out, double* __restrict__ score_out,
                                                      Best* __restrict__ best_out, long long idx_off) {
#pragma clang fp contract(off)
    __shared__ double red[256];
    __shared__ Best sh[4];
    __shared__ int is_last;
    const int r = blockIdx.x;
    const double* v = VT + (int64_t)r * ldv;
    double s = 0.0;
    for (int64_t j = threadIdx.x; j < N; j += 256) s += v[j] * v[j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // agent-scope stores + an explicit wait instead of __threadfence(): the fence is an L2 write-back on this chip (7-60 us for
        // a device-wide hand-over, tools/ubench_gridbar.hip), the write-through stores cost nothing beside it
        __hip_atomic_store(q + r, red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mu_raw + r, v[N], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        is_last = atomicAdd(counter, 1u) == (unsigned)(R - 1);
    }
    __syncthreads();
    if (!is_last) return;
    if (threadIdx.x == 0) *counter = 0u;
    double f_best = -INFINITY;
    long long idx = -1;
    if ((int)threadIdx.x < R) {
        const int c = threadIdx.x;
        double s2 = sigma2 - __hip_atomic_load(q + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s2 < 0.0) s2 = 0.0;  // predict_f: max(sigma2, 0)
        const double mu = beta + __hip_atomic_load(mu_raw + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mu_out) mu_out[c] = mu;
        if (var_out) var_out[c] = s2;
        const double f = acq_eval(ap, mu, s2);
        if (score_out) score_out[c] = f;
        if (f > -INFINITY) { f_best = f; idx = c; }  // false for NaN and -Inf
    }
    if (best_out) {
        block_argmax(f_best, idx, sh);
        if (threadIdx.x == 0) { best_out->val = idx >= 0 ? f_best : -INFINITY; best_out->idx = idx >= 0 ? idx + idx_off : -1; }
    }
}

// idx_off: global column of this shard's first candidate (sharded scoring: the record leaves the kernel ready for the exchange)
__global__ __launch_bounds__(256) void k_argmax_final(const Best* __restrict__ in, int n, Best* __restrict__ out, long long idx_off) {
    __shared__ Best sh[4];
    double v = -INFINITY;
    long long idx = -1;
    for (int i = threadIdx.x; i < n; i += 256)
        if (better(in[i].val, in[i].idx, v, idx)) { v = in[i].val; idx = in[i].idx; }
    block_argmax(v, idx, sh);
    if (threadIdx.x == 0) { out->val = idx >= 0 ? v : -INFINITY; out->idx = idx >= 0 ? idx + idx_off : -1; }
}

// The exchange step of sharded scoring: `all` holds nrec records per draw-slot layout [rec][S] gathered
// from every shard (global indices); thread s reduces slot s over the records in (value desc, index asc) order -- the
// same rule on every rank, so every rank holds the same winner, and it equals the unsharded arg-max (first maximum wins,
// reference src/acquisition.jl:62).
__global__ __launch_bounds__(256) void k_reduce_records(const Best* __restrict__ all, int nrec, int S, Best* __restrict__ out) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    double v = -INFINITY;
    long long idx = -1;
    for (int r = 0; r < nrec; ++r) {
        const Best b = all[(int64_t)r * S + s];
        if (better(b.val, b.idx, v, idx)) { v = b.val; idx = b.idx; }
    }
    out[s].val = idx >= 0 ? v : -INFINITY;
    out[s].idx = idx;
}

// ------------------------------------------------------------------------------------------------
// A8: gradient of the acquisition w.r.t. the candidate (role of ForwardDiff in wrap_gradient,
// reference src/acquisition.jl:11-17), analytic:
//     d k*_j / d x_k = fac(r_j) il2_k (x_k - X_jk)      fac = -k (SE), -(5/3) s2 (1+s) e^-s (Mat52)
//     grad mu  = sum_j alpha_j dk*_j          grad s2 = -2 sum_j u_j dk*_j,   u = K^-1 k* = W'(W k*)
// U' = V' W is a second triangular contraction on the MFMA engine (k_gemm, B = W in N-major form);
// this kernel finishes: one wave per candidate, lanes stride the observations, 2d butterfly sums,
// then the chain rule through the REFERENCE's acquisition formulas (not the textbook EI).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void acq_partials(const AcqParams& a, double mu, double s2, double& dmu, double& ds2) {
    const double inv_sqrt_2pi = 0.3989422804014327;
    switch (a.acq) {
        case ACQ_EI: {
            if (s2 == 0.0) { dmu = mu > a.p0 ? 1.0 : 0.0; ds2 = 0.0; return; }
            const double D = mu - a.p0, s = sqrt(s2), z = D / s;
            const double Phi = 0.5 * (1.0 + erf(z / sqrt(2.0))), phi = inv_sqrt_2pi * exp(-0.5 * z * z);
            dmu = Phi + D * phi / s - z * phi / s;
            ds2 = (D * phi - z * phi) * (-z / (2.0 * s2));
            return;
        }
        case ACQ_PI: {
            if (s2 == 0.0) { dmu = 0.0; ds2 = 0.0; return; }
            const double D = mu - a.p0, s = sqrt(s2), z = D / s, phi = inv_sqrt_2pi * exp(-0.5 * z * z);
            dmu = phi / s;
            ds2 = phi * (-z / (2.0 * s2));
            return;
        }
        case ACQ_UCB: dmu = 1.0; ds2 = s2 > 0.0 ? a.p0 / (2.0 * sqrt(s2)) : 0.0; return;
        case ACQ_MI: dmu = 1.0; ds2 = a.p0 / (2.0 * sqrt(s2 + a.p1)); return;
        default: dmu = 1.0; ds2 = 0.0; return;
    }
}

// Small batches: the posterior finish (q = sum V^2, mu, sigma^2, acquisition value: what k_small_finish does) rides on this kernel --
// one launch (~7 us, whatever it does) less per value+gradient pass.  VT == nullptr: off (mu / var come from the caller).
// q = sum_j V'[r][j]^2, sigma^2 and mu in EXACTLY the operations and order of k_small_finish (256 lane-strided sums, halving tree,
// no contraction): the value path and the gradient path agree on the scores bit for bit (tests/test_parity_gpu.py test_score_grad_vs_oracle)
__device__ __forceinline__ double sumsq_like_small_finish(const double* __restrict__ v, int64_t N, double* red) {
#pragma clang fp contract(off)
    double s = 0.0;
    for (int64_t j = threadIdx.x; j < N; j += 256) s += v[j] * v[j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    return red[0];
}
__device__ __forceinline__ void posterior_like_small_finish(double sigma2, double beta, double q, double mu_raw, double& mu, double& s2) {
#pragma clang fp contract(off)
    s2 = sigma2 - q;
    if (s2 < 0.0) s2 = 0.0;  // predict_f: max(sigma2, 0)
    mu = beta + mu_raw;
}
struct GradQ {
    const double* VT;   // [R][ldv]  V' rows; entry N of a row is mu - beta
    int64_t ldv;
    double sigma2, beta;
    double *mu_out, *var_out, *score_out;
    const unsigned* go;   // not null: return at once when the word is 0 (free-running ascent, see AscentState::ticket)
};
template <int DT>
__global__ __launch_bounds__(256) void k_grad_finish(const double* __restrict__ X, int64_t N,
                                                     const double* __restrict__ Xs, int64_t r_begin, int64_t r_end,
                                                     KernelHyper hp, const double* __restrict__ alpha,
                                                     const double* __restrict__ UT, int64_t ldu,
                                                     const double* __restrict__ mu, const double* __restrict__ var,
                                                     AcqParams ap, double* __restrict__ grad,
                                                     double* __restrict__ parts, unsigned* __restrict__ counters, GradQ gq) {
    // workgroup (r, sp): candidate r, observations [sp len, (sp + 1) len): 256 threads stride them, 2d sums reduced in
    // a fixed order.  gridDim.y = 1 for large batches (one workgroup per candidate is plenty); for a handful of
    // candidates the observations are split over gridDim.y workgroups (a single workgroup walking N = 10^4
    // observations is latency-bound: 140 us), the LAST one to finish adds the partial sums in split order -- the
    // result depends on (N, gridDim.y) only, never on the batch -- and leaves the counter at zero for the next call.
    __shared__ double red[4][2 * DT + 2];
    __shared__ int is_last;
    if (gq.go && *gq.go == 0u) return;
    constexpr int PS = 2 * DT + 2;   // partial record of one split: 2 DT gradient sums + the split's part of q
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = r_begin + blockIdx.x;
    if (r >= r_end) return;
    const int d = hp.d, S = gridDim.y, sp = blockIdx.y;
    const int64_t len = ((N + S - 1) / S + 255) / 256 * 256;
    const int64_t j_lo = sp * len, j_hi = min(N, j_lo + len);
    const double* xs = Xs + r * d;
    double gm[DT], gv[DT];
#pragma unroll
    for (int k = 0; k < DT; ++k) { gm[k] = 0.0; gv[k] = 0.0; }
    const double* u = UT + (r - r_begin) * ldu;
    const double* vq = gq.VT ? gq.VT + (r - r_begin) * gq.ldv : nullptr;
    for (int64_t j = j_lo + threadIdx.x; j < j_hi; j += 256) {
        double t[DT], rr = 0.0;
#pragma unroll
        for (int k = 0; k < DT; ++k)
            if (k < d) {
                t[k] = xs[k] - X[j * d + k];
                rr += hp.il2[k] * (t[k] * t[k]);
            }
        double fac;
        if (hp.kern == KERN_MAT52ARD) {
            const double s = sqrt(5.0) * sqrt(rr);
            fac = -(5.0 / 3.0) * hp.sigma2 * (1.0 + s) * exp(-s);
        } else {
            fac = -(hp.sigma2 * exp(-0.5 * rr));
        }
        const double a = alpha[j], uj = u[j];
#pragma unroll
        for (int k = 0; k < DT; ++k)
            if (k < d) {
                const double dk = fac * t[k] * hp.il2[k];
                gm[k] += dk * a;
                gv[k] += dk * uj;
            }
    }
#pragma unroll
    for (int k = 0; k < DT; ++k)
        if (k < d) {
            double a = gm[k], b = gv[k];
            for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
            if (lane == 0) { red[wave][2 * k] = a; red[wave][2 * k + 1] = b; }
        }
    __syncthreads();
    // q = sum_j V'[r][j]^2 in k_small_finish's exact order: ONE workgroup adds it -- the only one, or split 0, which does so BEFORE it
    // counts itself in (round 3 left it to the last arriver: a pass over the row + a tree BEHIND the counter, ~4 us of the critical path)
    double q_all = 0.0;
    __shared__ double redq[256];
    if (vq && (S == 1 || sp == 0)) q_all = sumsq_like_small_finish(vq, N, redq);
    if (S > 1) {
        double* mine = parts + ((int64_t)blockIdx.x * S + sp) * PS;
        if (threadIdx.x < 2 * d)
            __hip_atomic_store(mine + threadIdx.x, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (vq && sp == 0 && threadIdx.x == 0) __hip_atomic_store(mine + 2 * DT, q_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (agent-scope stores + this wait, not __threadfence(): see k_small_finish)
        __syncthreads();
        if (threadIdx.x == 0) is_last = atomicAdd(&counters[blockIdx.x], 1u) == (unsigned)(S - 1);
        __syncthreads();
        if (!is_last) return;
        if (threadIdx.x == 0) counters[blockIdx.x] = 0u;
        if (vq) q_all = __hip_atomic_load(parts + (int64_t)blockIdx.x * S * PS + 2 * DT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x < d) {
        const int k = threadIdx.x;
        double a, b;
        if (S > 1) {
            a = 0.0; b = 0.0;
            const double* all = parts + (int64_t)blockIdx.x * S * PS;
            for (int q = 0; q < S; ++q) {
                a += __hip_atomic_load(all + q * PS + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b += __hip_atomic_load(all + q * PS + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            a = (red[0][2 * k] + red[1][2 * k]) + (red[2][2 * k] + red[3][2 * k]);
            b = (red[0][2 * k + 1] + red[1][2 * k + 1]) + (red[2][2 * k + 1] + red[3][2 * k + 1]);
        }
        double dmu, ds2;
        double m, v;
        if (vq) {   // the posterior finish of k_small_finish, per candidate: the reference's clamp and formulas
            posterior_like_small_finish(gq.sigma2, gq.beta, q_all, vq[N], m, v);
            if (k == 0) {
                if (gq.mu_out) gq.mu_out[r] = m;
                if (gq.var_out) gq.var_out[r] = v;
                if (gq.score_out) gq.score_out[r] = acq_eval(ap, m, v);
            }
        } else {
            m = mu[r];
            v = var[r];
        }
        acq_partials(ap, m, v, dmu, ds2);
        // a clamped variance (sigma^2 == 0 exactly) has zero gradient, like max(., 0) under ForwardDiff
        grad[r * d + k] = dmu * a + (v > 0.0 ? ds2 * (-2.0 * b) : 0.0);
    }
}

// Large batches: one workgroup per GC = 2 candidates.  With one candidate per workgroup every workgroup streams the
// whole observation block X and alpha through L2 (R x 216 KB = 0.9 GB at R = 4096, N = 3000: 180 us); two candidates
// share each X row and alpha element in registers (0.11 ms).  Same per-candidate summation order as k_grad_finish with one split.
constexpr int GC = 2;   // 4 needs 259 VGPRs (320 with the coordinates in LDS) and is slower: 0.16 vs 0.11 ms at R = 4096
template <int DT>
__global__ __launch_bounds__(256) void k_grad_finish_tiled(const double* __restrict__ X, int64_t N,
                                                           const double* __restrict__ Xs, int64_t r_begin, int64_t r_end,
                                                           KernelHyper hp, const double* __restrict__ alpha,
                                                           const double* __restrict__ UT, int64_t ldu,
                                                           const double* __restrict__ mu, const double* __restrict__ var,
                                                           AcqParams ap, double* __restrict__ grad) {
    __shared__ double red[4][GC][2 * DT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t rb = r_begin + (int64_t)blockIdx.x * GC;
    if (rb >= r_end) return;
    const int d = hp.d, nc = (int)min((int64_t)GC, r_end - rb);
    double xs[GC][DT], gm[GC][DT], gv[GC][DT], w[DT];
#pragma unroll
    for (int k = 0; k < DT; ++k) w[k] = k < d ? hp.il2[k] : 0.0;
#pragma unroll
    for (int c = 0; c < GC; ++c)
#pragma unroll
        for (int k = 0; k < DT; ++k) {
            xs[c][k] = (c < nc && k < d) ? Xs[(rb + c) * d + k] : 0.0;
            gm[c][k] = 0.0;
            gv[c][k] = 0.0;
        }
    for (int64_t j = threadIdx.x; j < N; j += 256) {
        double xj[DT];
#pragma unroll
        for (int k = 0; k < DT; ++k) xj[k] = k < d ? X[j * d + k] : 0.0;
        const double a = alpha[j];
#pragma unroll
        for (int c = 0; c < GC; ++c) {
            double t[DT], rr = 0.0;
#pragma unroll
            for (int k = 0; k < DT; ++k) {
                t[k] = xs[c][k] - xj[k];
                rr += w[k] * (t[k] * t[k]);
            }
            double fac;
            if (hp.kern == KERN_MAT52ARD) {
                const double s = sqrt(5.0) * sqrt(rr);
                fac = -(5.0 / 3.0) * hp.sigma2 * (1.0 + s) * exp(-s);
            } else {
                fac = -(hp.sigma2 * exp(-0.5 * rr));
            }
            const double uj = c < nc ? UT[(rb + c - r_begin) * ldu + j] : 0.0;
#pragma unroll
            for (int k = 0; k < DT; ++k) {
                const double dk = fac * t[k] * w[k];
                gm[c][k] += dk * a;
                gv[c][k] += dk * uj;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < GC; ++c)
#pragma unroll
        for (int k = 0; k < DT; ++k)
            if (k < d) {
                double a = gm[c][k], b = gv[c][k];
                for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
                if (lane == 0) { red[wave][c][2 * k] = a; red[wave][c][2 * k + 1] = b; }
            }
    __syncthreads();
    if ((int)threadIdx.x < GC * d) {
        const int c = threadIdx.x / d, k = threadIdx.x % d;
        if (c < nc) {
            const int64_t r = rb + c;
            const double a = (red[0][c][2 * k] + red[1][c][2 * k]) + (red[2][c][2 * k] + red[3][c][2 * k]);
            const double b = (red[0][c][2 * k + 1] + red[1][c][2 * k + 1]) + (red[2][c][2 * k + 1] + red[3][c][2 * k + 1]);
            double dmu, ds2;
            const double m = mu[r], v = var[r];
            acq_partials(ap, m, v, dmu, ds2);
            grad[r * d + k] = dmu * a + (v > 0.0 ? ds2 * (-2.0 * b) : 0.0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// A9: counter-based standard normals.  z(seed, s, j) = Box-Muller of two splitmix64-derived
// uniforms keyed on (seed, s, j); identical on host and device, independent of sharding.
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline double thompson_normal(uint64_t seed, int64_t s, int64_t j) {
    const uint64_t h = splitmix64(seed ^ splitmix64((uint64_t)s * 0xD1B54A32D192ED03ull + (uint64_t)j));
    const uint64_t h2 = splitmix64(h);
    const double u1 = ((double)(h >> 11) + 1.0) * (1.0 / 9007199254740993.0);  // (0,1)
    const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);         // [0,1)
    return sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
}

// one workgroup per draw s; threads stride the candidates; never materialises S x R.
__global__ __launch_bounds__(256) void k_thompson(const double* __restrict__ mu, const double* __restrict__ var,
                                                  int64_t R, uint64_t seed, int64_t j0, Best* __restrict__ out,
                                                  long long idx_off) {
#pragma clang fp contract(off)
    __shared__ Best sh[4];
    const int64_t s = blockIdx.x;
    double v = -INFINITY;
    long long idx = -1;
    for (int64_t r = threadIdx.x; r < R; r += 256) {
        const double f = mu[r] + sqrt(var[r]) * thompson_normal(seed, s, j0 + r);
        if (better(f, r, v, idx)) { v = f; idx = r; }
    }
    block_argmax(v, idx, sh);
    if (threadIdx.x == 0) { out[s].val = idx >= 0 ? v : -INFINITY; out[s].idx = idx >= 0 ? idx + idx_off : -1; }
}

}  // namespace bohip
