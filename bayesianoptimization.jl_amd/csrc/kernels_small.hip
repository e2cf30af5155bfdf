// kernels_small.hip -- round 5: the small-batch posterior pass (what the reference's DEFAULT usage runs: 10 L-BFGS restarts per
// acquire_max, reference src/acquisition.jl:4-6,54-68, each evaluation a value + gradient of <= 16 candidates) as TWO kernels:
//
//   k_small_v   K*' assembly (in the prologue of the tiles that need it)  ->  V' = K*' W'  ->  q = sum v^2, mu, sigma^2, the
//               acquisition value and the arg-max of the batch                               (replaces k_kstar + k_trimv_stream + k_small_finish)
//   k_small_u   U' = V' W  ->  the analytic gradient of the acquisition                     (replaces k_trimv_stream + k_grad_finish)
//
// Round 4's pass was five kernels whose dependent latency chains abutted (kstar 5 us, V' 15.4, U' 17.2, gradient 15.6 at N = 3000): each
// triangular product was a barrier-stepped LDS-DMA ring at 0.35 of the plain read rate of W, and the gradient kernel a 15 us chain for
// 0.4 MB.  Here a triangular product with <= 16 right-hand sides is a dense contraction against K-MAJOR tiles of the resident matrix:
//   out[c][r] = sum_k A[k][c] rhs[k][r],   A = W' (k <= c) for V',  A = W (k >= c) for U'
// (row k of A contiguous in c: a lane loads 16 bytes of ONE row, a wave four rows x 256 bytes, every byte of the triangle exactly once, no
// LDS staging of the matrix, no barrier inside the stream) on v_mfma_f64_4x4x4_4b: the four blocks of one instruction are four groups
// of four COLUMNS c against the same 4 x 4 block of right-hand sides (the A operand is read from LDS with an address that ignores the
// block index, which replicates it for free), i.e. 16 columns x 4 right-hand sides x 4 contraction indices in 16 cycles, ceil(P / 4)
// instructions per k-step -- the 16x16x4 form of the same product runs at 100-140 cycles per instruction on this chip (gemm_core.h).
// Work decomposition: a workgroup (8 waves = 2 contraction halves x 4 column strips of 32) owns a TILE = one 128-column block x a
// segment of m 128-chunks of the contraction index; the tiles of a column block leave their partial sums in a scratch plane (16-byte
// write-through stores), count themselves in, and the LAST ARRIVER adds them in segment order -- the result depends on (N, m) only,
// never on the batch, the candidate's position in it or which workgroup came last (reference property test/acquisitionfunctions.jl:
// 8-11: batch == single, bit for bit).  The column block's finisher goes straight on: sum v^2 over its 128 columns (V pass) or the
// gradient sums over its 128 observations (U pass) into a per-block record; the finisher of the LAST column block adds the records in
// block order and writes mu, sigma^2, value, arg-max / the gradient.  Counters are left at zero.
// Right-hand-side SLOTS: candidate r of a pass of 16 lives in slot 4 (r % 4) + r / 4 of every [.][16] array (the lane that holds the
// results of MFMA g for right-hand side 4 g + q then owns four consecutive slots 4 q ... 4 q + 3).
#include "gemm_core.h"

namespace bohip {

constexpr int SP_THREADS = 512;

struct SmallCommon {
    const double* A;        // W' (V pass) or W (U pass), K-major, leading dimension ld
    int64_t ld;
    int N, T, m, P;         // observations, 128-blocks, chunks per tile segment, candidates of the call
    double* part;           // [pass][ntiles][128][16] partial tiles
    int ntiles;
    unsigned* cnt;          // [pass][T] tile arrivals per column block, then [pass] finished column blocks, then [1] finished passes
    const double* X;        // [N][d]
    const double* Xs;       // [P][d] candidates
    const double* alpha;
    const unsigned* go;     // not null: return at once when the word is 0 (free-running ascent)
#ifdef BOHIP_SMALL_TRACE
    unsigned long long* trace;   // [2][workgroup][16] wall-clock marks (tools/small_pass_trace.py; measurement build only)
#endif
};
#ifdef BOHIP_SMALL_TRACE
#define SM_MARK(sc, kern, i) do { if (threadIdx.x == 0) (sc).trace[(((kern) * 4096 + blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define SM_MARK(sc, kern, i) do { } while (0)
#endif
struct SmallV {
    double* v16;            // [pass][T * 128][16]   V' in slot layout (the U pass's right-hand sides)
    double* qpart;          // [pass][T][16]
    double* mupart;         // [pass][T][16]
    double* fstash;         // [SMALL_MAX] scores of all passes (arg-max over several passes)
    double* ks16;           // [pass][T * 128][16]   K*' in slot layout: written by the tile that holds a column block's diagonal chunk, read by the U
                            // pass's finisher of that block (SE kernels: d k*_j / d x = -k*_j (x - X_j) / l^2, no second exponential)
    double sigma2, beta;
    AcqParams ap;
    double *mu_out, *var_out, *score_out;
    Best* best_out;
    long long idx_off;
    int finish;             // 1: this kernel also finishes the posterior (value-only call); 0: k_small_u's last workgroup does (one counter level and
                            // one serial tail less in front of the U pass)
};
// Round 6: the free-running ascent's step (kernels_ascent.hip asc_step_one) runs in k_small_u's LAST workgroup, right behind the gradient --
// its own launch was 4.6 us of launch + one round of loads before the first flop, the third kernel of every pass.  One pass of <= 16
// candidates and d <= 16 only (four start points per wave); everything else keeps k_asc_step.
struct SmallFold {
    int on;                 // 0: no step here; 1: asc_step_compute (a pass of the search); 2: asc_first_rows (the start points' own evaluation)
    int R, ring_slot;
    AscentState st;
    const double *lb, *ub;
    double ftol_rel, xtol_abs, first_step_scale;
};
struct SmallU {
    const double* v16;
    SmallV sv;              // the posterior finish rides in this kernel's last workgroup (same device function as the value-only call: same bits)
    double* gpart;          // [pass][ntiles][16][DT]  the tiles' u-weighted gradient sums
    double* gmpart;         // [pass][T][16][DT]       the column blocks' alpha-weighted gradient sums
    double* grad;           // [P][d]
    SmallFold fold;
};

__device__ __forceinline__ int small_slot_to_r(int slot) { return 4 * (slot & 3) + (slot >> 2); }
// sum over the 16 lanes of a DPP row by four rotate-and-add levels (row_ror 8, 4, 2, 1: VALU moves, no trip through the LDS crossbar);
// every lane ends with the sum of its row in its own association -- callers use ONE fixed lane's
template <int N>
__device__ __forceinline__ double dpp_row_ror(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x120 + N, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x120 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_row_ror<8>(v);
    v += dpp_row_ror<4>(v);
    v += dpp_row_ror<2>(v);
    v += dpp_row_ror<1>(v);
    return v;
}

// which tile is workgroup b: column blocks heaviest first (lower: T-1 ... 0, upper: 0 ... T-1), a block's segments in contraction order
template <int UPPER>
__device__ __forceinline__ void small_tile(int b, int T, int m, int& cb, int& kc0, int& kc1, int& t0, int& nseg) {
    // the extents run T, T - 1, ..., 1 in both forms; nseg = ceil(ext / m) drops by one every m steps (no integer division on the scalar unit)
    t0 = 0;
    int ext = T, rem = T % m;                     // rem = ext mod m
    nseg = T / m + (rem ? 1 : 0);
    for (int o = 0; o < T; ++o) {
        if (b < nseg) break;
        b -= nseg;
        t0 += nseg;
        --ext;
        if (rem == 1) --nseg;                     // ext went from q m + 1 to q m
        rem = rem == 0 ? m - 1 : rem - 1;
        if (m == 1) nseg = ext;
    }
    cb = UPPER ? T - ext : ext - 1;
    const int klo = UPPER ? cb : 0;
    kc0 = klo + b * m;
    kc1 = min(klo + ext, kc0 + m);
}
static int small_ntiles(int T, int m) {
    int n = 0;
    for (int e = 1; e <= T; ++e) n += (e + m - 1) / m;
    return n;
}


// The contraction of one tile: chunks [kc0, kc1) of the contraction index against the workgroup's 128 columns; every wave leaves its
// partial sums in acc (lane (q, p), wave strip wc: column cb * 128 + 32 wc + 2 p + e, right-hand sides 4 g + q).
// Order of issue = order of need: the first right-hand-side tile's inputs (pre), the candidates / observations for LDS (setup_load;
// setup_store writes them and ends with a barrier), then the 16 loads of the matrix; from there on the next chunk's inputs are requested BEFORE the current
// chunk's products so that they land behind them:  pre(kc) -> registers,  fill(kc, registers, tile) -> LDS.  Two LDS tiles: the fill of
// chunk kc + 1 writes the tile that chunk kc - 1 used, and every wave has left chunk kc - 1 when it passes chunk kc's barrier.
// On return rt2 + ((kc1 - 1 - kc0) & 1) * 2048 holds the tile of chunk kc1 - 1.
template <int UPPER, int G, bool PRE_FIRST, class SetupL, class SetupS, class Pre, class Fill>
__device__ __forceinline__ bool small_contract(const SmallCommon& sc, unsigned gow, double* rt2, int cb, int kc0, int kc1, SetupL&& setup_load, SetupS&& setup_store, Pre&& pre,
                                               Fill&& fill, double (&acc)[2][4]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, wc = wave & 3, q = lane >> 4, p = lane & 15;
    const int N = sc.N;
    const int c0 = cb * 128 + wc * 32;
    const int64_t ld = sc.ld;
    const double* ap = sc.A + (int64_t)(kc0 * 128 + 64 * kh + q) * ld + c0 + 2 * p;
    // (loads return in order: what is needed first is requested first -- the first right-hand-side tile's inputs, the candidates, then the
    // sixteen loads of the matrix; the tile is then built while the matrix is on its way)
    // (PRE_FIRST = false -- the U pass: its right-hand sides may have to be WAITED for in the fused form, the matrix must be on its way by then)
    decltype(pre(kc0)) regs;
    if (PRE_FIRST) regs = pre(kc0);
    auto sregs = setup_load();
    // DEPTH loads of 16 bytes per lane in flight (8 = half a chunk ahead, 64 KB per CU).  A whole chunk ahead (16: 128 KB per CU, 32 MB on the
    // chip) is past what the memory side takes at full rate -- tools/ubench_readbw shows the same for a bare read of 400 MB (32 KB per CU in
    // flight: 6.4 TB/s, 128 KB: 4.7) -- and costs 32 registers: value + gradient of ten candidates, 16 / 12 / 8 / 6 / 4 in flight:
    // N = 4500 76.6 / 75.9 / 75.3 / 75.5 / 74.4 us, N = 6000 99.6 / 98.5 / 97.0 / 102.5 / 95.8, N = 10^4 226.5 / 222.9 / 221.9 / 237 / 229,
    // N = 14000 397 / 390 / 386 / 426 / 408 (profiles/r05_small_pass_prefetch_depth.txt); N = 3000 (W in the Infinity Cache): within 1 %.
#ifndef SMALL_DEPTH
#define SMALL_DEPTH 8
#endif
    constexpr int DEPTH = SMALL_DEPTH;
    d2 w[DEPTH];
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) w[s] = *(const d2*)(ap + (int64_t)(4 * s) * ld);
    SM_MARK(sc, UPPER, 1);
    setup_store(sregs);
    // gow: the free-running ascent's "anybody still active?" word, requested as the kernel's FIRST load and looked at only here, with every
    // input of the first chunk on its way (as the kernel's first statement the word's round trip stood in front of all of them; nothing up to
    // here has a side effect outside the CU)
    if (gow == 0u) return false;
    if (!PRE_FIRST) regs = pre(kc0);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[e][g] = 0.0;
    for (int kc = kc0; kc < kc1; ++kc) {
        double* rt = rt2 + ((kc - kc0) & 1) * 2048;
        fill(kc, regs, rt);
        __syncthreads();
        if (kc == kc0) SM_MARK(sc, UPPER, 2);
        const bool edge = kc == cb || kc * 128 + 128 > N, more = kc + 1 < kc1;
        if (more) regs = pre(kc + 1);
        const double* apn = ap + (int64_t)(kc + 1 - kc0) * 128 * ld;
        const double* ra = rt + (64 * kh + q) * 16 + 4 * (lane & 3);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const d2 a01 = *(const d2*)(ra + s * 64);
            d2 a23 = {0.0, 0.0};
            if (G > 2) a23 = *(const d2*)(ra + s * 64 + 2);
            d2 wv = w[s % DEPTH];
            if (s + DEPTH < 16) w[s % DEPTH] = *(const d2*)(apn - (int64_t)128 * ld + (int64_t)(4 * (s + DEPTH)) * ld);
            else if (more) w[s % DEPTH] = *(const d2*)(apn + (int64_t)(4 * (s + DEPTH - 16)) * ld);
            if (edge) {
                const int k = kc * 128 + 64 * kh + 4 * s + q, c = c0 + 2 * p;
                const bool okx = k < N && (UPPER ? k >= c : k <= c), oky = k < N && (UPPER ? k >= c + 1 : k <= c + 1);
                if (!okx) wv.x = 0.0;
                if (!oky) wv.y = 0.0;
            }
            acc[0][0] = mfma444(a01.x, wv.x, acc[0][0]);
            acc[1][0] = mfma444(a01.x, wv.y, acc[1][0]);
            if (G > 1) { acc[0][1] = mfma444(a01.y, wv.x, acc[0][1]); acc[1][1] = mfma444(a01.y, wv.y, acc[1][1]); }
            if (G > 2) { acc[0][2] = mfma444(a23.x, wv.x, acc[0][2]); acc[1][2] = mfma444(a23.x, wv.y, acc[1][2]); }
            if (G > 3) { acc[0][3] = mfma444(a23.y, wv.x, acc[0][3]); acc[1][3] = mfma444(a23.y, wv.y, acc[1][3]); }
        }
    }
    return true;
}
// Publication of the tile + the column block's combine.  Returns true in the LAST ARRIVER of the column block, with the combined sums of
// pieces (tid, tid + 512) in sum[0..1]: piece e = c_local * 8 + sp  <->  out[cb * 128 + c_local][slots 2 sp, 2 sp + 1].
// meanwhile(): a hook behind the first round of loads (unused since the U pass stopped publishing partial tiles; the V pass passes nothing).
// The partial tiles come as 16-byte agent-scope loads issued and waited for in ONE asm sequence (8-byte atomic loads read handed-off data at
// 0.54-0.70 of the 16-byte rate, MI355X_MICROARCH.md: 196 KB per finisher at N = 3000).
template <int UPPER, class Meanwhile>
__device__ __forceinline__ bool small_publish_combine(const SmallCommon& sc, int pass, int tile, double* red, int* flag, int cb, int t0, int nseg,
                                                      const double (&acc)[2][4], d2 (&sum)[2], Meanwhile&& meanwhile) {
    SM_MARK(sc, UPPER, 3);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, wc = wave & 3, q = lane >> 4, p = lane & 15;
    // contraction half 1 hands its sums to half 0 (fixed order: half 0 + half 1), which publishes the tile
    if (kh == 1) {
        double* dst = red + (wc * 64 + lane) * 8;
#pragma unroll
        for (int e = 0; e < 2; ++e) { *(d2*)(dst + 4 * e) = d2{acc[e][0], acc[e][1]}; *(d2*)(dst + 4 * e + 2) = d2{acc[e][2], acc[e][3]}; }
    }
    __syncthreads();
    if (kh == 0) {
        const double* src = red + (wc * 64 + lane) * 8;
        double* pt = sc.part + ((int64_t)pass * sc.ntiles + tile) * 2048;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const d2 r01 = *(const d2*)(src + 4 * e), r23 = *(const d2*)(src + 4 * e + 2);
            double* dst = pt + (wc * 32 + 2 * p + e) * 16 + 4 * q;      // lane (q, p) holds column c0 + 2 p + e, right-hand sides 4 g + q = slots 4 q + g
            st_agent2(dst, acc[e][0] + r01.x, acc[e][1] + r01.y);
            st_agent2(dst + 2, acc[e][2] + r23.x, acc[e][3] + r23.y);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tile (and whatever else this workgroup published) has left the CU before it is counted in
    __syncthreads();
    SM_MARK(sc, UPPER, 4);
    if (tid == 0) *flag = atomicAdd(&sc.cnt[pass * sc.T + cb], 1u) == (unsigned)(nseg - 1);
    __syncthreads();
    SM_MARK(sc, UPPER, 5);
    if (!*flag) return false;
    if (tid == 0) sc.cnt[pass * sc.T + cb] = 0u;
    const double* p0 = sc.part + ((int64_t)pass * sc.ntiles + t0) * 2048 + 2 * tid;
    sum[0] = d2{0.0, 0.0};
    sum[1] = d2{0.0, 0.0};
    for (int s0 = 0; s0 < nseg; s0 += 12) {         // twelve segments in flight per piece as 16-byte loads (one round trip to the memory side), added in segment order
        d2 v[2][12];
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int si = min(s0 + j, nseg - 1);
            ld_agent_x2_issue(p0 + (int64_t)si * 2048, v[0][j]);
            ld_agent_x2_issue(p0 + (int64_t)si * 2048 + 1024, v[1][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[0][3]), "+v"(v[0][4]), "+v"(v[0][5]), "+v"(v[0][6]), "+v"(v[0][7]),
                       "+v"(v[0][8]), "+v"(v[0][9]), "+v"(v[0][10]), "+v"(v[0][11]), "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[1][2]), "+v"(v[1][3]),
                       "+v"(v[1][4]), "+v"(v[1][5]), "+v"(v[1][6]), "+v"(v[1][7]), "+v"(v[1][8]), "+v"(v[1][9]), "+v"(v[1][10]), "+v"(v[1][11])
                     :
                     : "memory");
        if (s0 == 0) meanwhile();
#pragma unroll
        for (int j = 0; j < 12; ++j)
            if (s0 + j < nseg) { sum[0] += v[0][j]; sum[1] += v[1][j]; }
    }
    SM_MARK(sc, UPPER, 6);
    return true;
}

// The posterior finish of one pass of 16 candidates (k_small_finish's formulas): q = sum of the column blocks' records in block order,
// mu - beta likewise, sigma^2 = max(s_f^2 - q, 0), the acquisition value.  The records are fetched by all threads at once (one round
// trip), then added by one thread per slot from LDS.  buf: >= 4096 doubles of LDS.  Thread `slot` (< 16) returns (f, candidate index) and
// (mu, sigma^2); threads 64 + slot return (mu, sigma^2) too.
template <class Meanwhile>
__device__ __forceinline__ void small_posterior_final(const SmallV& sv, int pass, int T, int P, double* buf, double& f_out, long long& idx_out,
                                                      double& mu_o, double& s2_o, Meanwhile&& meanwhile) {
    const int tid = threadIdx.x;
    double qs = 0.0, mr = 0.0;
    for (int b0 = 0; b0 < T; b0 += 128) {
        const int nb = min(128, T - b0);
        __syncthreads();
        double qv[4], mv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * SP_THREADS;
            qv[i] = e < nb * 16 ? ld_agent(sv.qpart + ((int64_t)pass * T + b0) * 16 + e) : 0.0;
            mv[i] = e < nb * 16 ? ld_agent(sv.mupart + ((int64_t)pass * T + b0) * 16 + e) : 0.0;
        }
        if (b0 == 0) meanwhile();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * SP_THREADS;
            if (e < nb * 16) { buf[e] = qv[i]; buf[2048 + e] = mv[i]; }
        }
        __syncthreads();
        if (tid < 128 && (tid & 63) < 16) {      // slot = lane, in waves 0 and 1 (the second copy: see mu_o / s2_o below); eight LDS reads in flight per step
            const int sl_ = tid & 63;
            for (int b = 0; b < nb; b += 8) {
                double tq[8], tm[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { tq[j] = buf[min(b + j, nb - 1) * 16 + sl_]; tm[j] = buf[2048 + min(b + j, nb - 1) * 16 + sl_]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) if (b + j < nb) qs += tq[j];
#pragma unroll
                for (int j = 0; j < 8; ++j) if (b + j < nb) mr += tm[j];
            }
        }
    }
    f_out = -INFINITY;
    idx_out = -1;
    mu_o = 0.0; s2_o = 0.0;
    if (tid >= 64 && tid < 80) {     // wave 1's copy of (mu, sigma^2) of slot tid - 64: the caller may run the chain rule's partials there, beside wave 0's acquisition value
        if (pass * 16 + small_slot_to_r(tid - 64) < P) posterior_like_small_finish(sv.sigma2, sv.beta, qs, mr, mu_o, s2_o);
    }
    if (tid < 16) {
        const int rr = pass * 16 + small_slot_to_r(tid);
        if (rr < P) {
            double mu, s2;
            posterior_like_small_finish(sv.sigma2, sv.beta, qs, mr, mu, s2);
            mu_o = mu; s2_o = s2;
            if (sv.mu_out) sv.mu_out[rr] = mu;
            if (sv.var_out) sv.var_out[rr] = s2;
            const double f = acq_eval(sv.ap, mu, s2);
            if (sv.score_out) sv.score_out[rr] = f;
            f_out = f;
            idx_out = rr;
        }
    }
}

// ---- V pass ---------------------------------------------------------------------------------------------------------------------------
template <int DT>
struct SmallXRow { double x[DT]; };
// LDS of one workgroup of these kernels (the fused kernel runs either body in the same block)
template <int DT>
struct SmallLds {
    double* lbuf;      // [3][2048]: two right-hand-side tiles + the hand-over of the contraction halves; the finishers' scratch afterwards
    double* xs_l;      // [16][DT] candidates of the pass
    double* xc_l;      // [128][DT] the column block's observations (U pass)
    double* al_l;      // [128] their alpha (U pass)
    double* small;     // [256] mu partial sums (V pass) / [32] posterior of the pass (U pass)
    Best* shb;         // [SP_THREADS / 64]
    int* flag;
};
#define SMALL_LDS_DECL(DT_, L_)                                                          \
    __shared__ __attribute__((aligned(16))) double sl_lbuf_[3 * 2048];                    \
    __shared__ double sl_xs_[16 * DT_];                                                   \
    __shared__ double sl_xc_[128 * DT_];                                                  \
    __shared__ double sl_al_[128];                                                        \
    __shared__ double sl_small_[256];                                                     \
    __shared__ Best sl_shb_[SP_THREADS / 64];                                             \
    __shared__ int sl_flag_;                                                              \
    SmallLds<DT_> L_{sl_lbuf_, sl_xs_, sl_xc_, sl_al_, sl_small_, sl_shb_, &sl_flag_}

template <int DT, int G>
__device__ __forceinline__ void small_v_body(const SmallCommon& sc, unsigned gow, const SmallV& sv, const KernelHyper& hp, int tile, const SmallLds<DT>& L) {
    double* const lbuf = L.lbuf;
    double* const rt2 = lbuf;             // 2 x [128][16] right-hand-side tiles
    double* const red = lbuf + 4096;      // [4][64][8] hand-over of the contraction halves, later the finishers' scratch
    double* const xs_l = L.xs_l;
    double* const mured = L.small;
    Best* const shb = L.shb;
    int& flag = *L.flag;
    const int tid = threadIdx.x, pass = blockIdx.y, d = hp.d, N = sc.N, T = sc.T;
    SM_MARK(sc, 0, 0);
    int cb, kc0, kc1, t0, nseg;
    small_tile<0>(tile, T, sc.m, cb, kc0, kc1, t0, nseg);
    // the pass's candidates (row = r): 16 DT <= 1024 values, two per thread
    auto setup_load = [&]() {
        d2 v = {0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = tid + u * SP_THREADS, r = t / DT, k = t % DT, rr = pass * 16 + r;
            const double x = (t < 16 * DT && rr < sc.P && k < d) ? sc.Xs[(int64_t)rr * d + k] : 0.0;
            if (u) v.y = x; else v.x = x;
        }
        return v;
    };
    auto setup_store = [&](const d2& v) {
        if (tid < 16 * DT) xs_l[tid] = v.x;
        if (tid + SP_THREADS < 16 * DT) xs_l[tid + SP_THREADS] = v.y;
        __syncthreads();
    };
    // K*' of a chunk: thread = (observation k, right-hand sides 4 g + i for g < G); k_kstar's expression, operation for operation
    auto pre = [&](int kc) {
        SmallXRow<DT> xr;
        const int k = kc * 128 + (tid >> 2);
#pragma unroll
        for (int kk = 0; kk < DT; ++kk) xr.x[kk] = (kk < d && k < N) ? sc.X[(int64_t)k * d + kk] : 0.0;
        return xr;
    };
    auto fill = [&](int kc, const SmallXRow<DT>& xr, double* rt_) {
        const int kl = tid >> 2, i = tid & 3, k = kc * 128 + kl;
        double v[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int r = 4 * g + i;
            double rr = 0.0;
#pragma unroll
            for (int kk = 0; kk < DT; ++kk) {
                const double t = xr.x[kk] - xs_l[r * DT + kk];
                rr += (kk < d ? hp.il2[kk] : 0.0) * (t * t);
            }
            v[g] = (k < N && pass * 16 + r < sc.P) ? cov_from_r_fast(hp.kern, hp.sigma2, rr) : 0.0;
        }
        *(d2*)(rt_ + kl * 16 + 4 * i) = d2{v[0], v[1]};
        *(d2*)(rt_ + kl * 16 + 4 * i + 2) = d2{v[2], v[3]};
    };
    d2 sum[2];
    double acc[2][4];
    if (!small_contract<0, G, true>(sc, gow, rt2, cb, kc0, kc1, setup_load, setup_store, pre, fill, acc)) return;
    // mu - beta = alpha' k*: the tile that holds the DIAGONAL chunk of its column block (the last chunk of the block's last segment) adds that
    // chunk's share in a fixed order from the K*' tile still in LDS, and publishes the record BEFORE it counts itself in
    if (kc1 - 1 == cb) {
        const double* rt = rt2 + ((kc1 - 1 - kc0) & 1) * 2048;
        if (!sv.finish) {     // (gradient call) K*' of the chunk for the U pass's finisher
            double* kd = sv.ks16 + ((int64_t)pass * T * 128 + cb * 128) * 16 + 4 * tid;
            *(d2*)kd = *(const d2*)(rt + 4 * tid);
            *(d2*)(kd + 2) = *(const d2*)(rt + 4 * tid + 2);
        }
        if (tid < 256) {
            const int slot = tid & 15, pt = tid >> 4;
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = cb * 128 + pt * 8 + j;
                s += (k < N ? sc.alpha[k] : 0.0) * rt[(pt * 8 + j) * 16 + slot];
            }
            mured[pt * 16 + slot] = s;
        }
        __syncthreads();
        if (tid < 16) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) s += mured[j * 16 + tid];
            st_agent(sv.mupart + ((int64_t)pass * T + cb) * 16 + tid, s);
        }
    }
    if (!small_publish_combine<0>(sc, pass, tile, red, &flag, cb, t0, nseg, acc, sum, [] {})) return;
    // ---- the column block's finisher: V' of its 128 columns, q record
    const int sp = tid & 7, cl0 = tid >> 3;                 // pieces (cl0, sp) and (cl0 + 64, sp)
    double* vrow = sv.v16 + ((int64_t)pass * T * 128 + cb * 128) * 16;
    *(d2*)(vrow + (cl0) * 16 + 2 * sp) = sum[0];          // (plain stores: the next KERNEL reads them)
    *(d2*)(vrow + (cl0 + 64) * 16 + 2 * sp) = sum[1];
    d2 qq = {0.0, 0.0};
    if (cb * 128 + cl0 < N) { qq.x += sum[0].x * sum[0].x; qq.y += sum[0].y * sum[0].y; }
    if (cb * 128 + cl0 + 64 < N) { qq.x += sum[1].x * sum[1].x; qq.y += sum[1].y * sum[1].y; }
    *(d2*)(lbuf + cl0 * 16 + 2 * sp) = qq;                  // (the right-hand-side tiles are done with: every wave is past the contraction)
    __syncthreads();
    if (tid < 16) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 64; ++j) s += lbuf[j * 16 + tid];
        st_agent(sv.qpart + ((int64_t)pass * T + cb) * 16 + tid, s);
    }
    SM_MARK(sc, 0, 7);
    if (!sv.finish) return;                 // (gradient call: the U pass's last workgroup finishes the posterior)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) flag = atomicAdd(&sc.cnt[gridDim.y * T + pass], 1u) == (unsigned)(T - 1);
    __syncthreads();
    if (!flag) return;
    SM_MARK(sc, 0, 8);
    if (tid == 0) sc.cnt[gridDim.y * T + pass] = 0u;
    // ---- the pass's finisher: q, mu, sigma^2, value per candidate, arg-max of the batch
    double f_best, mu_, s2_;
    long long idx;
    const int npass = gridDim.y;
    small_posterior_final(sv, pass, T, sc.P, lbuf, f_best, idx, mu_, s2_, [] {});
    SM_MARK(sc, 0, 9);
    if (!sv.best_out) return;
    if (npass > 1) {
        if (idx >= 0) st_agent(sv.fstash + idx, f_best);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) flag = atomicAdd(&sc.cnt[npass * T + npass], 1u) == (unsigned)(npass - 1);
        __syncthreads();
        if (!flag) return;
        if (tid == 0) sc.cnt[npass * T + npass] = 0u;
        f_best = -INFINITY; idx = -1;
        if (tid < sc.P) { f_best = ld_agent(sv.fstash + tid); idx = tid; }
    }
    if (!(f_best > -INFINITY)) idx = -1;   // NaN and -Inf never win (reference: `f > maxf` is false for NaN)
    block_argmax(f_best, idx, shb);
    SM_MARK(sc, 0, 10);
    if (tid == 0) { sv.best_out->val = idx >= 0 ? f_best : -INFINITY; sv.best_out->idx = idx >= 0 ? idx + sv.idx_off : -1; }
}
template <int DT, int G>
__global__ __launch_bounds__(SP_THREADS) void k_small_v(SmallCommon sc, SmallV sv, KernelHyper hp) {
    const unsigned gow = sc.go ? *sc.go : 1u;
    SMALL_LDS_DECL(DT, L);
    small_v_body<DT, G>(sc, gow, sv, hp, blockIdx.x, L);
}

// ---- U pass ---------------------------------------------------------------------------------------------------------------------------
// The gradient (k_grad_finish's formulas: reference src/acquisition.jl:11-17 wrap_gradient's role, analytic) needs u = W'(W k*) only inside
//   gv[k] = sum_j (d k*_j / d x_k) u_j ,        u_j = sum_{i >= j} W[i][j] v_i
// which is LINEAR in u: a tile's share of u (its 128 columns j, its segment of rows i) is contracted with d k*_j / d x_k inside the tile,
// and what leaves the tile is one record of 16 slots x 2 d sums (gv, and -- from the tile of the block's first segment -- the alpha-weighted
// gm[k] = sum_j (d k*_j / d x_k) alpha_j of those 128 observations).  No partial tiles of u, no column-block combine, ONE counter per pass:
// the LAST tile adds the records in tile order and applies the chain rule.  (Until this form the U pass published 16-KiB partial tiles like
// the V pass, a block's last arriver combined them and added the gradient sums of its 128 observations alone: 22 us of serial tail behind a
// 10 us contraction, profiles/r05_small_pass_trace_*.txt.)  For the SE kernels d k*_j / d x_k = -k*_j (x_k - X_jk) / l_k^2 with k*_j taken
// from the V pass's own K*' record (no second exponential).
struct SmallVPair { d2 a, b; };
template <int DT, int G>
__device__ __forceinline__ void small_u_body(const SmallCommon& sc, unsigned gow, const SmallU& su, const KernelHyper& hp, int tile, const SmallLds<DT>& L) {
    double* const lbuf = L.lbuf;
    double* const rt2 = lbuf;
    double* const xs_l = L.xs_l;
    double* const xc_l = L.xc_l;          // the column block's 128 observations and their alpha (fetched behind the matrix loads)
    double* const al_l = L.al_l;
    double* const post_l = L.small;
    int& flag = *L.flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), pass = blockIdx.y, d = hp.d, N = sc.N, T = sc.T;
    const int kh = wave >> 2, wc = wave & 3, q = lane >> 4, p = lane & 15;
    SM_MARK(sc, 1, 0);
    int cb, kc0, kc1, t0, nseg;
    small_tile<1>(tile, T, sc.m, cb, kc0, kc1, t0, nseg);
    const double* vsrc = su.v16 + (int64_t)pass * T * 128 * 16;
    d2 kv[2][2];                                 // K*' of this lane's two columns, slots 4 q ... 4 q + 3 (from the V pass)
    auto setup_load = [&]() { return 0; };
    auto setup_store = [&](int) {
        // candidates, the block's observations, alpha and K*': requested BEHIND the matrix loads (nothing needs them before the contraction is
        // done); no barrier here (read behind the contraction's barriers)
        for (int t = tid; t < 16 * DT; t += SP_THREADS) {
            const int r = t / DT, k = t % DT, rr = pass * 16 + r;
            xs_l[t] = (rr < sc.P && k < d) ? sc.Xs[(int64_t)rr * d + k] : 0.0;
        }
        const int64_t e0 = (int64_t)cb * 128 * d, e1 = min((int64_t)N * d, e0 + 128 * d);
        for (int t = tid; t < 128 * d; t += SP_THREADS) xc_l[t] = e0 + t < e1 ? sc.X[e0 + t] : 0.0;
        if (tid < 128) al_l[tid] = cb * 128 + tid < N ? sc.alpha[cb * 128 + tid] : 0.0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const double* ks = su.sv.ks16 + ((int64_t)pass * T * 128 + cb * 128 + wc * 32 + 2 * p + e) * 16 + 4 * q;
            kv[e][0] = *(const d2*)ks;
            kv[e][1] = *(const d2*)(ks + 2);
        }
    };
    auto pre = [&](int kc) {                    // V' of the chunk (slot layout, written by the V pass), rows beyond N as zeros
        SmallVPair v{{0.0, 0.0}, {0.0, 0.0}};
        if (kc * 128 + (tid >> 2) < N) {
            const double* src = vsrc + (int64_t)kc * 2048 + 4 * tid;
            v.a = *(const d2*)src;
            v.b = *(const d2*)(src + 2);
        }
        return v;
    };
    auto fill = [&](int, const SmallVPair& v, double* rt_) {
        *(d2*)(rt_ + 4 * tid) = v.a;
        *(d2*)(rt_ + 4 * tid + 2) = v.b;
    };
    double acc[2][4];
    if (!small_contract<1, G, true>(sc, gow, rt2, cb, kc0, kc1, setup_load, setup_store, pre, fill, acc)) return;
    SM_MARK(sc, 1, 3);
    // ---- the tile's share of the gradient sums: lane (q, p) of strip wc holds u's share for columns j_e = 128 cb + 32 wc + 2 p + e and the
    // candidates r_g = 4 g + q
    double fac[2][4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int cl = wc * 32 + 2 * p + e, j = cb * 128 + cl;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            fac[e][g] = 0.0;
            if (g < G && j < N && pass * 16 + 4 * g + q < sc.P) {
                if (hp.kern == KERN_MAT52ARD) {
                    double rr = 0.0;
#pragma unroll
                    for (int k = 0; k < DT; ++k)
                        if (k < d) { const double t = xs_l[(4 * g + q) * DT + k] - xc_l[cl * d + k]; rr += hp.il2[k] * (t * t); }
                    const double s5 = sqrt(5.0) * sqrt(rr);
                    fac[e][g] = -(5.0 / 3.0) * hp.sigma2 * (1.0 + s5) * exp(-s5);
                } else {
                    fac[e][g] = -(g < 2 ? (g == 0 ? kv[e][0].x : kv[e][0].y) : (g == 2 ? kv[e][1].x : kv[e][1].y));   // -(sigma2 exp(-rr / 2)): the V pass's own value
                }
            }
        }
    }
    // z[c][slot] = (u's share, both contraction halves) x (kernel factor) for the tile's 128 columns -> LDS (slot layout); then
    //   gv[slot][k] = sum_c z[c][slot] (x*_k - X_ck) / l_k^2
    // by threads (slot, k, group of 32 columns) walking their columns, four groups added in group order -- 64 LDS reads and 32 FMAs per thread and
    // round of 8 dimensions.  (Until this form every lane multiplied its own 2 x G values by d k*/dx and 24 sixteen-lane DPP sums per lane followed:
    // 5 us per tile; lane sums are what a latency-bound tail should not be made of.)  The alpha-weighted sums of the block's first tile the same
    // way from the factors themselves.
    double* const zt = lbuf;                     // [128][16]  (the right-hand-side tiles are done with)
    double* const fat = lbuf + 2048;             // [128][16]  factors x alpha, first tile of the block only
    double* const redh = lbuf + 4096;            // [4][64][8] hand-over of the contraction halves, then [4 groups][16 slots][8][2] partial sums
    if (kh == 1) {
        double* dst = redh + (wc * 64 + lane) * 8;
#pragma unroll
        for (int e = 0; e < 2; ++e) { *(d2*)(dst + 4 * e) = d2{acc[e][0], acc[e][1]}; *(d2*)(dst + 4 * e + 2) = d2{acc[e][2], acc[e][3]}; }
    }
    __syncthreads();
    if (kh == 0) {
        const double* src = redh + (wc * 64 + lane) * 8;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int cl = wc * 32 + 2 * p + e;
            const d2 r01 = *(const d2*)(src + 4 * e), r23 = *(const d2*)(src + 4 * e + 2);
            *(d2*)(zt + cl * 16 + 4 * q) = d2{(acc[e][0] + r01.x) * fac[e][0], (acc[e][1] + r01.y) * fac[e][1]};
            *(d2*)(zt + cl * 16 + 4 * q + 2) = d2{(acc[e][2] + r23.x) * fac[e][2], (acc[e][3] + r23.y) * fac[e][3]};
            if (kc0 == cb) {
                const double a_ = al_l[cl];
                *(d2*)(fat + cl * 16 + 4 * q) = d2{fac[e][0] * a_, fac[e][1] * a_};
                *(d2*)(fat + cl * 16 + 4 * q + 2) = d2{fac[e][2] * a_, fac[e][3] * a_};
            }
        }
    }
    __syncthreads();
    double* gvrec = su.gpart + (((int64_t)pass * sc.ntiles + tile) * 16) * DT;            // this tile's u-weighted sums [slot][DT]
    double* gmrec = su.gmpart + (((int64_t)pass * T + cb) * 16) * DT;                   // the block's alpha-weighted sums (first segment's tile)
    {
        const int slot = tid & 15, k8 = (tid >> 4) & 7, cg = tid >> 7, rs = small_slot_to_r(slot);
#pragma unroll
        for (int kb = 0; kb < DT; kb += 8) {
            if (kb >= d) break;
            const int k = kb + k8;
            double sv_ = 0.0, sm = 0.0;
            if (k < DT && k < d) {
                const double xk = xs_l[rs * DT + k], w = hp.il2[k];
                const double* zc = zt + (32 * cg) * 16 + slot;
                const double* fc = fat + (32 * cg) * 16 + slot;
                const double* xc = xc_l + (32 * cg) * d + k;
#pragma unroll 8
                for (int c = 0; c < 32; ++c) {
                    const double t = (xk - xc[c * d]) * w;
                    sv_ += zc[c * 16] * t;
                    if (kc0 == cb) sm += fc[c * 16] * t;
                }
            }
            if (kb > 0) __syncthreads();         // (the previous round's readers are done; round 0: redh was last read before the barrier above)
            *(d2*)(redh + ((cg * 16 + slot) * 8 + k8) * 2) = d2{sm, sv_};
            __syncthreads();
            if (tid < 64) {                       // (slot, dimension pair): the four column groups in group order
                const int sl = tid >> 2, kk = 2 * (tid & 3);
                d2 s0 = {0.0, 0.0}, s1 = {0.0, 0.0};       // (gm, gv) of dimensions kb + kk, kb + kk + 1
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    s0 += *(const d2*)(redh + ((g4 * 16 + sl) * 8 + kk) * 2);
                    s1 += *(const d2*)(redh + ((g4 * 16 + sl) * 8 + kk + 1) * 2);
                }
                if (kb + kk < DT) {
                    st_agent2(gvrec + sl * DT + kb + kk, s0.y, s1.y);
                    if (kc0 == cb) st_agent2(gmrec + sl * DT + kb + kk, s0.x, s1.x);
                }
            }
        }
    }
    SM_MARK(sc, 1, 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) flag = atomicAdd(&sc.cnt[pass], 1u) == (unsigned)(sc.ntiles - 1);
    __syncthreads();
    SM_MARK(sc, 1, 5);
    if (!flag) return;
    if (tid == 0) sc.cnt[pass] = 0u;
    SM_MARK(sc, 1, 9);
    // ---- the pass's finisher: the posterior of the pass's candidates (what k_small_v's finisher does in a value-only call: same function,
    // same bits) and the tiles' records in tile order; then the chain rule through the reference's acquisition formulas.  (slot, dimension)
    // pairs; the records of a pair are fetched by `nparts` threads, each a contiguous range of tiles, twenty 16-byte loads in flight, and the
    // parts added in part order: the order depends on (T, m, d) only
    // Two things run side by side in this workgroup.  WAVES 0 AND 1, lanes 0..15 (slot = lane): the posterior of the pass's candidates -- each
    // lane fetches its slot's T q records and T mu records itself (all in flight at once) and adds them in block order, exactly the sums of
    // small_posterior_final (the value-only call's finisher): same numbers, same order, same bits; wave 0 then evaluates the acquisition,
    // wave 1 the chain rule's partials (erf / exp in both).  WAVES 2..7: pairs = (candidate, two dimensions); a pair's tile records are
    // fetched by `nparts` threads, each a contiguous range of tiles (and of column blocks for the alpha-weighted sums), all of them in
    // flight at once; the parts are added in part order: the order depends on (T, m, d) only.
    const double* gp = su.gpart + ((int64_t)pass * sc.ntiles * 16) * DT;
    const double* gmp = su.gmpart + ((int64_t)pass * T * 16) * DT;
    const int pcand = min(16, sc.P - pass * 16), dh = (d + 1) / 2;      // candidates of this pass, dimension pairs
    const int npairs = pcand * dh;                                       // <= 16 * 32
    constexpr int FT = SP_THREADS - 128;                                 // fetching threads
    // (the step's state: requested now, used behind the gradient -- its round trip to memory rides under the posterior and the records' fetch)
    AscStepRegs fold_regs;
    if (su.fold.on == 1 && wave < ((su.fold.R + 3) >> 2))
        asc_step_load<true>(su.fold.st, 4 * wave + (lane >> 4), lane & 15, d, su.fold.R, su.fold.lb, su.fold.ub, fold_regs);
    double* gfin = lbuf;                                                 // [nparts][npairs][4]  (<= 384 x 4)
    double* const fold_g = lbuf + 4096;                                  // [16][16] the pass's gradients, [16] its values: the ascent's step below
    double* const fold_f = lbuf + 4096 + 256;
    unsigned* const fold_cnt = reinterpret_cast<unsigned*>(lbuf + 4096 + 256 + 16);
    if (tid < 128) {
        const int slot_ = tid & 63;
        if (slot_ < 16 && pass * 16 + small_slot_to_r(slot_) < sc.P) {
            const double* qsrc = su.sv.qpart + (int64_t)pass * T * 16 + slot_;
            const double* msrc = su.sv.mupart + (int64_t)pass * T * 16 + slot_;
            double qs = 0.0, mr = 0.0;
            for (int b0 = 0; b0 < T; b0 += 16) {
                double tq[16], tm[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) { tq[j] = ld_agent(qsrc + (int64_t)min(b0 + j, T - 1) * 16); tm[j] = ld_agent(msrc + (int64_t)min(b0 + j, T - 1) * 16); }
#pragma unroll
                for (int j = 0; j < 16; ++j) if (b0 + j < T) qs += tq[j];
#pragma unroll
                for (int j = 0; j < 16; ++j) if (b0 + j < T) mr += tm[j];
            }
            double mu, s2;
            posterior_like_small_finish(su.sv.sigma2, su.sv.beta, qs, mr, mu, s2);
            if (tid < 64) {
                const int rr = pass * 16 + small_slot_to_r(slot_);
                if (su.sv.mu_out) su.sv.mu_out[rr] = mu;
                if (su.sv.var_out) su.sv.var_out[rr] = s2;
                const double fv = acq_eval(su.sv.ap, mu, s2);
                if (su.sv.score_out) su.sv.score_out[rr] = fv;
                fold_f[rr & 15] = fv;
            } else {
                double dmu, ds2;
                acq_partials(su.sv.ap, mu, s2, dmu, ds2);
                post_l[2 * slot_] = dmu;
                post_l[2 * slot_ + 1] = ds2;
                post_l[32 + slot_] = s2;
            }
        }
        SM_MARK(sc, 1, 7);
    }
    const int ft = tid - 128;
    if (npairs <= FT) {
        int nparts = max(1, min(16, FT / npairs));
        const int bpp = (sc.ntiles + nparts - 1) / nparts, bpm = (T + nparts - 1) / nparts;
        const int pr = ft >= 0 ? ft % npairs : 0, pt = ft >= 0 ? ft / npairs : 0;
        const bool mine = ft >= 0 && ft < npairs * nparts;
        const int ri = pr / dh, k2 = pr % dh, sl = 4 * (ri & 3) + (ri >> 2);       // candidate ri of the pass lives in slot 4 (ri % 4) + ri / 4
        if (mine) {
            d2 ga = {0.0, 0.0}, gb = {0.0, 0.0};                 // (gm, gv) of dimensions 2 k2, 2 k2 + 1
            const double* src = gp + (int64_t)sl * DT + 2 * k2;
            const double* srm = gmp + (int64_t)sl * DT + 2 * k2;
            for (int b0 = 0; b0 < bpp; b0 += 24) {
                const int bl0 = pt * bpp + b0, bl1 = min(sc.ntiles, pt * bpp + min(bpp, b0 + 24));
                const int ml0 = pt * bpm, ml1 = b0 == 0 ? min(T, ml0 + min(bpm, 4)) : ml0;
                d2 v[24], vm[4];
#pragma unroll
                for (int j = 0; j < 24; ++j) ld_agent_x2_issue(src + (int64_t)max(0, min(bl0 + j, bl1 - 1)) * 16 * DT, v[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) ld_agent_x2_issue(srm + (int64_t)max(0, min(ml0 + j, ml1 - 1)) * 16 * DT, vm[j]);
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
                               "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(v[16]), "+v"(v[17]), "+v"(v[18]), "+v"(v[19]),
                               "+v"(v[20]), "+v"(v[21]), "+v"(v[22]), "+v"(v[23]), "+v"(vm[0]), "+v"(vm[1]), "+v"(vm[2]), "+v"(vm[3])
                             :
                             : "memory");
#pragma unroll
                for (int j = 0; j < 24; ++j)
                    if (bl0 + j < bl1) gb += v[j];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (ml0 + j < ml1) ga += vm[j];
            }
            for (int b = pt * bpm + 4; b < min(T, (pt + 1) * bpm); ++b) {        // (more than four column blocks per part: T > 64)
                d2 vmx;
                ld_agent_x2_issue(srm + (int64_t)b * 16 * DT, vmx);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(vmx) : : "memory");
                ga += vmx;
            }
            *(d2*)(gfin + (pt * npairs + pr) * 4) = ga;
            *(d2*)(gfin + (pt * npairs + pr) * 4 + 2) = gb;
            if (ft == 0) SM_MARK(sc, 1, 6);
        }
        __syncthreads();
        SM_MARK(sc, 1, 10);
        if (tid < npairs) {
            const int ri2 = tid / dh, k22 = tid % dh, sl2 = 4 * (ri2 & 3) + (ri2 >> 2);
            d2 a = {0.0, 0.0}, b = {0.0, 0.0};
#pragma unroll
            for (int pp = 0; pp < 16; ++pp)
                if (pp < nparts) { a += *(const d2*)(gfin + (pp * npairs + tid) * 4); b += *(const d2*)(gfin + (pp * npairs + tid) * 4 + 2); }
            const int rr = pass * 16 + ri2;
            const double dmu = post_l[2 * sl2], ds2 = post_l[2 * sl2 + 1], v = post_l[32 + sl2];
            // a clamped variance (sigma^2 == 0 exactly) has zero gradient, like max(., 0) under ForwardDiff
            const double g0 = dmu * a.x + (v > 0.0 ? ds2 * (-2.0 * b.x) : 0.0), g1 = dmu * a.y + (v > 0.0 ? ds2 * (-2.0 * b.y) : 0.0);
            su.grad[(int64_t)rr * d + 2 * k22] = g0;
            if (2 * k22 + 1 < d) su.grad[(int64_t)rr * d + 2 * k22 + 1] = g1;
            fold_g[ri2 * 16 + ((2 * k22) & 15)] = g0;
            fold_g[ri2 * 16 + ((2 * k22 + 1) & 15)] = g1;
        }
        if (su.fold.on) {
            // ---- the ascent's step of every start point of the pass (free-running form: what k_asc_step did in a launch of its own)
            if (tid == 0) *fold_cnt = 0u;
            __syncthreads();
            const int nw = (su.fold.R + 3) >> 2;
            if (wave < nw) {
                const int row = lane >> 4, k = lane & 15, r = 4 * wave + row;
                const double ft_ = fold_f[r & 15], gt_ = k < d ? fold_g[(r & 15) * 16 + k] : 0.0;
                const int act = su.fold.on == 1 ? asc_step_compute<true>(su.fold.st, fold_regs, r, k, d, su.fold.R, su.fold.ftol_rel, su.fold.xtol_abs,
                                                                         su.fold.ring_slot, ft_, gt_)
                                                : asc_first_rows(su.fold.st, r, k, d, su.fold.R, su.fold.lb, su.fold.ub, su.fold.first_step_scale, ft_, gt_);
                const unsigned long long bal = __ballot(act != 0 && k == 0 && r < su.fold.R);
                if (lane == 0) {
                    const unsigned mine = (unsigned)__popcll(bal);
                    const unsigned before = atomicAdd(fold_cnt, (mine << 8) | 1u);
                    if ((before & 0xffu) == (unsigned)nw - 1u) {      // the last of the step's waves publishes the pass's count (asc_step_one's tail)
                        const unsigned n = (before >> 8) + mine;
                        __hip_atomic_store(su.fold.st.ticket, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(su.fold.st.h_cnt + su.fold.ring_slot, (int)n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
        }
    } else {
        // (more pairs than fetching threads: d > 48 with a full pass -- every thread of the workgroup takes pairs in turn, one record at a time)
        __syncthreads();
        for (int e = tid; e < npairs; e += SP_THREADS) {
            const int ri2 = e / dh, k22 = e % dh, sl2 = 4 * (ri2 & 3) + (ri2 >> 2);
            d2 a = {0.0, 0.0}, b = {0.0, 0.0};
            for (int t0 = 0; t0 < sc.ntiles; t0 += 8) {          // eight records in flight, added in tile order
                d2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) ld_agent_x2_issue(gp + ((int64_t)min(t0 + j, sc.ntiles - 1) * 16 + sl2) * DT + 2 * k22, v[j]);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) if (t0 + j < sc.ntiles) b += v[j];
            }
            for (int t0 = 0; t0 < T; t0 += 8) {
                d2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) ld_agent_x2_issue(gmp + ((int64_t)min(t0 + j, T - 1) * 16 + sl2) * DT + 2 * k22, v[j]);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) if (t0 + j < T) a += v[j];
            }
            const int rr = pass * 16 + ri2;
            const double dmu = post_l[2 * sl2], ds2 = post_l[2 * sl2 + 1], v = post_l[32 + sl2];
            su.grad[(int64_t)rr * d + 2 * k22] = dmu * a.x + (v > 0.0 ? ds2 * (-2.0 * b.x) : 0.0);
            if (2 * k22 + 1 < d) su.grad[(int64_t)rr * d + 2 * k22 + 1] = dmu * a.y + (v > 0.0 ? ds2 * (-2.0 * b.y) : 0.0);
        }
    }
    SM_MARK(sc, 1, 11);
}
template <int DT, int G>
__global__ __launch_bounds__(SP_THREADS) void k_small_u(SmallCommon sc, SmallU su, KernelHyper hp) {
    const unsigned gow = sc.go ? *sc.go : 1u;
    SMALL_LDS_DECL(DT, L);
    small_u_body<DT, G>(sc, gow, su, hp, blockIdx.x, L);
}

}  // namespace bohip
