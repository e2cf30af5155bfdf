// kernels_chol.hip -- the dataflow form of the blocked Cholesky factorisation (row A2 of SURVEY.md section 8; reference
// call sites: update!/append!/fit! at src/models/gp.jl:11-18, i.e. the ElasticPDMats / LAPACK potrf behind them).
//
// Why.  The launch-chained form (k_potf2_inv -> panel solve -> in-block update, one round per 128-block) is bound by its
// serial chain: ~66 us of single-workgroup diagonal-block work + two dependent launches (~2 x 20 us) per block, 24 blocks
// at N = 3000 = 2.75 ms for 9 GF.  Only three things are intrinsically serial: the pivot chain inside a diagonal block,
// the solve of the tile just below it, and the update of the NEXT diagonal block.  Here they are pipelined at 16-column
// granularity and never leave the chip's registers / LDS:
//
//   k_chol_chain   8 persistent workgroups, one per CU:
//                  2 row owners (rows alternate).  The owner of row r
//                    during block r-1  holds the diagonal tile (r, r) [lower, registers] and gives it one rank-16 update per
//                                      16-column panel of L(r, r-1) that row r's critical follower delivers through S;
//                    during block r    moves (r, r) -- fully updated -- into LDS and runs the pivot chain itself
//                                      (the body of k_potf2_inv without the inverse), publishing each finished panel of
//                                      L_rr and the inverse of its 16 x 16 pivot block with a release flag.
//                  2 critical followers: rows k+1 and k+2 of the current block k, one tile (r, k) each in registers:
//                    per published panel the row solve of their 128 rows (x = r W16'), the right-looking update of their
//                    remaining columns, their 16 new columns of L to S with a per-panel flag.
//                  4 gated-update workgroups: the two tiles (k+2, k+1), (k+2, k+2) the chain needs next, as an MFMA tile
//                    product whose contraction index arrives 16 columns at a time (gated_tile).
//                  The chain per block is therefore the pivot time + one flag hand-over, not pivot + 2 launches + 2 GEMMs.
//   k_chol_rows    one persistent workgroup per row i >= 3: the same panel follower for tile (i, k), k = 0 .. i-3.
//   k_chol_cols    two persistent workgroups per row i >= 3: gated_tile on (i, k+1), the tile the row follows next.
//   k_gemm_nt      (kernels_linalg.hip) the rest of block k's update -- rows >= k+3, columns >= k+2 -- on the MFMA engine,
//                  one launch per block on its own stream, gated by flags instead of host events and counting into the
//                  counters the persistent workgroups wait on.
//   k_inv128       afterwards: the inverses W_kk of all diagonal blocks at once (no inverse is needed during the factorisation
//                  any more), seeds of the triangular inverse W = L^-1.
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
#include <utility>

namespace bohip {

#ifndef BOHIP_PIVOT_W16GROW
#define BOHIP_PIVOT_W16GROW 1   // 1 (round 6): W16 grown inside the pivot block's factorisation, the rows below solved as a PRODUCT with it on the matrix pipe; 0: round 5's substitutions
#endif
constexpr int CH_THREADS = 512;
// Row stride of the owner's LDS image of the diagonal tile.  k_potf2_inv's 129 makes a thread-per-row walk conflict-free; the chain's pivot
// block no longer has one: its operand fetches are MFMA fragments (8 rows x 4 consecutive entries per 32 lanes), which want
// stride = 4 (mod 32): 132.  (With 129 the row-solve phase took 1.2 us of LDS conflicts per panel instead of 0.3.)
constexpr int CH_LD = BOHIP_PIVOT_W16GROW ? TILE + 4 : PF_LD;
constexpr int CH_WLS = BOHIP_PIVOT_W16GROW ? 20 : 16;   // row stride of the pivot block's inverse in LDS (20: see WK_S)
constexpr int CH_PANELS = TILE / 16;          // 8 panels of 16 columns per 128-block
constexpr int WK_LPS = 16;                    // worker LDS: published panel LP[128][16]
constexpr int WK_XS = 18;                     //             solved rows   XB[128][18]: rows 16-byte aligned (two entries per ds_read_b128), 16 consecutive rows on 16 different bank quads
#ifndef BOHIP_FOLLOW_MFMA
#define BOHIP_FOLLOW_MFMA 1   // 1 (round 6): the panel followers solve and update on the matrix pipe (follow_block below); 0: round 5's register-tiled VALU form
#endif
constexpr int WK_S = 20;                      // round 6, row stride of the follower's LDS arrays: 20 r + k (8-byte units) hits 32 different bank pairs for r < 8, k < 4 -- the
                                              // footprint of one MFMA operand fetch -- and keeps 16-byte alignment
#if BOHIP_FOLLOW_MFMA
constexpr int WK_LDS_DOUBLES = 2 * TILE * WK_S + 16 * WK_S + 2;   // Lp[128][20] | X[128][20] | W16[16][20] | next-panel-ready word: 43.5 KB (round 5: 55.5 -- a CU must still hold a column updater and a flagged launch's workgroup beside a follower)
#else
constexpr int WK_LDS_DOUBLES = 16 * (TILE + 2) + 2 * TILE * WK_XS + 256 + 2;   // LPt[16][130] | XB[128][18] | XS[128][18] | W16[16][16] | next-panel-ready word
#endif
constexpr int INV_LS = TILE - 16 + 4;           // row stride of the inverter's rows of L (116 = 20 mod 32: see WK_S)
constexpr int INV_LDS_DOUBLES = (TILE - 16) * TILE + 16 * TILE + 16 * INV_LS + 16 * WK_S;   // the inverter workgroup's image (inverter_role): Wimg | Tl | Lrow | W16
constexpr int CH_LDS_BYTES = (INV_LDS_DOUBLES > TILE * CH_LD + 2 * TILE + 2 * 16 * CH_WLS ? INV_LDS_DOUBLES : TILE * CH_LD + 2 * TILE + 2 * 16 * CH_WLS) * 8;   // the potf2 image + dl + idl + W16 scratch (the worker arrays alias its start), or the inverter's
static_assert(WK_LDS_DOUBLES * 8 <= CH_LDS_BYTES, "worker arrays must fit inside the diagonal-block image");

struct CholFlags {
    unsigned* panel;      // [T * 8]   panel p of block k is published (columns of L_kk in L, 1/diag in idl_g)
    unsigned* solved;     // [T]       mode2: W_kk = L_kk^-1 is in W / W' (k_chol_inverter)
    unsigned* crit;       // [T]       counter: waves of the update launch for row k+2 of block k that have finished
    unsigned* abort;      // [1]       set on a spin time-out: every wait returns at once
    double* w16_g;        // [T * 8][16][16] inverse of each 16 x 16 pivot block (lower, zeros above), published with its panel
    unsigned* xp;         // [T * T * 8] panel p of L(i, k) is complete in S: xp[(k T + i) 8 + p], every row i > k
    unsigned* colr;       // [T * T]     counter: waves that have stored their part of tile (i, k+1) updated with block k's panel: colr[k T + i]
    int T;
    unsigned* rest;       // [T]       counter: storing waves of row k+3 (the first row tile) of block k's FAR update (columns >= k+2)
    unsigned* col;        // [T]       the same for block k's update of column k+1 (tile (k+3, k+1): what row k+3's follower needs next)
    unsigned* farall;     // [T]       storing waves of the first 128 columns (column k+2) of block k's far update
    unsigned* fol;        // [T]       bulk followers of block k that have finished
    unsigned* colall;     // [T]       (every storing wave of block k's column update; not waited on)
    int mode2;            // > 0: large-T form, the value is the window `win` (6): block k's flagged update covers columns k+1 .. 4 (k / 4) + win -- (cholesky_dataflow2): no persistent followers beyond the chain -- rows >= k+3 are solved by
                          // a launch against the inverse W_kk (k_chol_inverter raises solved[k]), block k's flagged update covers the
                          // columns k+1 .. 4 (k / 4) + 6, the columns beyond get four blocks at a time from plain K = 512 launches
    unsigned crit_want;   // value of crit[k] when the whole row-(k+2) update launch of block k is in memory
    unsigned panel_want;  // value of panel[..] when every publishing wave has seen its stores land
    unsigned long long spin_ticks;   // bound of every in-kernel wait (wall_clock64 ticks), see flag_wait_ge
    int nsf;              // executor form: solve-follower workgroups of the chain kernel (rows k+3 .. k+2+nsf of block k)
    unsigned* xp3;        // [T * 8]   mode2 >= 200: panel p of S(k+3, k) is complete (solve follower 0): what the gated updates of row k+3 consume
    unsigned* pre3;       // [T * 4]   mode2 >= 200: tile (k+3, k+1+j) carries every block before k (executor: 16 per tile): pre3[4 k + j]
    unsigned* resident;   // [1]       persistent workgroups that have started (form 1: what k_chol_gate holds the flagged launches back for)
};

// Bound of every in-kernel wait, in ticks of wall_clock64() (100 MHz constant clock): 200 ms by default.  The longest legitimate
// wait of a whole factorisation is a few milliseconds, so a wait this long means a producer is not on the chip (another
// process holds the CUs, two streams share a hardware queue, ...): the waiter sets the abort word, every other wait returns
// at once, and the host falls back to the launch-chained form.  BOHIP_CHOL_SPIN_US overrides (tests force the time-out path).
__device__ __forceinline__ void flag_wait_ge(const unsigned* flag, unsigned want, unsigned* abort, unsigned long long spin_ticks) {
    // one thread spins; callers put a barrier behind it
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return;
        if (__hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > spin_ticks) {   // something upstream never arrived
            // (the first one to give up leaves the flag's word address behind: the host names it in its message)
            atomicCAS(abort, 0u, (unsigned)(reinterpret_cast<uintptr_t>(flag) >> 2) | 0x80000000u);
            return;
        }
    }
}
#ifndef BOHIP_CHOL_TRACE
#define BOHIP_CHOL_TRACE 0
#endif
#if BOHIP_CHOL_TRACE
__device__ unsigned long long g_chol_trace[8 * 1024];   // [4096, 4352): inverse of block k published; [4352, 4608): row k+2 of block k saw rest[k-1]; [4608, 4864): S(k+3, k) complete;
// [4864, 5120): owner of row r starts waiting for crit[r-2]; [5120, 5376): sees it; [5376, 5632): owner saw the last panel of L(r, r-1); [5632, 5888): follower of row k+1 sees crit[k-1]   // [0,1024): panel published; [1024,2048): owner saw panel; [2048,3072): owner finished panel; [3072..): block marks
#define CH_MARK(slot) g_chol_trace[(slot) & 8191] = wall_clock64()
#else
#define CH_MARK(slot) do {} while (0)
#endif
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16-byte agent-scope (sc1, write-through) store: ONE fabric write where two 8-byte atomic stores are two at 2.7x the time per byte
// (MI355X_MICROARCH.md, "stores of each flavour"); a panel hand-over was 2048 of the small ones.  Inline asm (the atomic builtin stops
// at 8 bytes); the s_nop is the wait state a > 64-bit VMEM store needs before its data registers are rewritten.
// (`volatile` 16-byte accesses were tried first: the compiler emits them sc0 sc1 = system scope, and a pivot panel then took 9.5 us
// instead of 7.)
__device__ __forceinline__ void st_agent2(double* p, double x, double y) {
    d2 v; v.x = x; v.y = y;
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
// 16-byte agent-scope loads (the atomic builtin stops at 8 bytes): issue, then ONE wait that also ties the results in
__device__ __forceinline__ void ld_agent_x2_issue(const double* p, d2& out) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(out) : "v"(p) : "memory");
}
// "All my stores have left this CU and were acknowledged": an explicit s_waitcnt.  NOT __builtin_amdgcn_fence(release,
// "workgroup") -- in the default (non-tgsplit) execution mode the compiler emits no vmcnt wait for that scope (checked in
// the ISA), the flag then overtook the data and large factorisations came out wrong once in a few runs.
__device__ __forceinline__ void release_wg() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void flag_set(unsigned* flag, unsigned v) {
    __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned* xp_at(const CholFlags& fl, int k, int i) { return fl.xp + ((size_t)k * fl.T + i) * CH_PANELS; }

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

#if BOHIP_FOLLOW_MFMA
// ---- round 6: the panel follower on the matrix pipe ------------------------------------------------------------------------------------
// Round 5's follower (kept below, BOHIP_FOLLOW_MFMA=0) held 8 rows x 4 columns per lane and paid, per panel, 32 broadcast reads of 16 bytes
// for the row solve (0.9 us of LDS time: an LDS read returns 64 lanes x its width whether or not the lanes share an address) and 96 per
// updating wave for the rank-16 update (2.5 us with seven waves updating): 4.3-5 us per panel against a pivot that publishes one every
// 4.3 us since the pivot block's own phases moved to the matrix pipe.  Here both are v_mfma_f64_4x4x4 products (2 x 2 blocks = 8 x 8 x 4 per
// instruction, gemm_core.h): wave w owns columns 16w .. 16w+15 of tile (i, k) as 16 x 2 accumulator blocks (32 doubles per lane, as before),
//     solve    X = R W16'            wave w: rows 16w .. 16w+15, 16 instructions, 16 operand fetches
//     update   A -= X Lp'            wave w > p: 128 instructions, 64 + 8 operand fetches of 8 bytes (one lane = one element, no broadcasts)
// The rate of the update is the CU's FP64 rate either way (MI355X: matrix = vector); what changes is the LDS traffic under it (3 x less)
// and the solve.  LDS arrays, row stride WK_S = 20:  Lp[c][m] = L_kk[c][16p + m] (staged as it arrives: rows, not transposed), W16[c][m], and
// X[r][m] = the panel's 16 columns -- handed over by the wave that holds them while the staging loads are in flight (one phase and one
// barrier less per panel), then solved IN PLACE: every wave reads the 16 rows it solves and writes -x over them.
// workgroup barrier that orders LDS accesses only (s_waitcnt lgkmcnt(0) + s_barrier): global loads and stores stay in flight across it
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ double lane_swap1(double v) {   // the value of lane ^ 1 (DPP quad_perm [1, 0, 3, 2])
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int H>
__device__ __forceinline__ void d1_rank16(const double* XB, double (&d)[18]);
template <bool WITH_D1, int H>
__device__ __forceinline__ void follow_block(const double* __restrict__ Lmat, int64_t ld, double* __restrict__ S,
                                             int i_tile, int k_blk, const CholFlags& fl, double* wk, double (&a)[32],
                                             double (&d)[18], unsigned* xf) {
    static_assert(!WITH_D1, "the owner keeps its own loop (chain_owner)");
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kq = lane >> 4, bb = (lane >> 2) & 3, t4 = lane & 3;
    const int ar = 4 * (bb >> 1) + t4, bc = 4 * (bb & 1) + t4, dr = 4 * (bb >> 1) + kq, dc = bc;
    double* Lp = wk;                          // [128][WK_S]
    double* XB = Lp + TILE * WK_S;            // [128][WK_S]  before the solve: the panel's columns;  after it: -x
    double* XS = XB;
    double* W16s = XB + TILE * WK_S;          // [16][WK_S]
    int* nready = reinterpret_cast<int*>(W16s + 16 * WK_S);
    const double* Lkk = Lmat + (int64_t)k_blk * TILE * (ld + 1);
    double pv0 = 0.0, pv1 = 0.0, pv2 = 0.0, pv3 = 0.0, pw = 0.0;
    bool have = false;
    auto hand_over = [&]() {   // this wave's 16 columns, final for the panel that comes next, to XB
#pragma unroll
        for (int rb = 0; rb < 16; ++rb) {
            XB[(8 * rb + dr) * WK_S + dc] = a[2 * rb];
            XB[(8 * rb + dr) * WK_S + 8 + dc] = a[2 * rb + 1];
        }
    };
    // Per panel, a follower that is BEHIND the pivot (its tile arrived late; every solve follower and far row is, at the start of a block)
    // must not pay a memory round trip it does not need: the look at the NEXT panel's flag is issued behind the first barrier and read
    // behind the solve (it used to sit in front of the barrier: ~1 us per panel), the prefetch of that panel rides under the update, and
    // the flag of panel p's columns of S is raised one panel later (p <= 5), when their stores have long landed, instead of behind a drain.
    // The barriers are LDS barriers (barrier_lds): __syncthreads() also waits for every outstanding GLOBAL access -- this panel's stores to S,
    // the prefetch of the next one -- which is exactly the latency the schedule keeps off the panel.
    const bool trc = BOHIP_CHOL_TRACE && t == 0 && k_blk == 5 && i_tile - k_blk == 4;   // (trace build: solve follower 1 of block 5, [5920 + 5 p + phase])
    unsigned fnext = 0u;   // thread 0: the flag of panel p + 1 as seen during the update of panel p - 1 (looked at again behind the solve if it was not up then)
    for (int p = 0; p < CH_PANELS; ++p) {
        if (!have) {   // (uniform: read from LDS by everybody)
            if (t == 0) flag_wait_ge(fl.panel + k_blk * CH_PANELS + p, fl.panel_want, fl.abort, fl.spin_ticks);
            barrier_lds();
        }
        if (trc) CH_MARK(5920 + 5 * p);
        {   // stage the published panel: rows 16p..127 of L_kk, columns 16p..16p+15 (4 threads per 128-B row segment), and W16;
            // the wave that holds the panel's columns hands them over under the loads
            const int i = t >> 2, mq = t & 3;
            const bool ldp = !have && i >= 16 * p, ldw = !have && t < 128;
            d2 u0, u1, uw;
            u0.x = pv0; u0.y = pv1; u1.x = pv2; u1.y = pv3; uw.x = uw.y = 0.0;
            if (ldp) {
                const double* src = Lkk + (int64_t)i * ld + 16 * p + 4 * mq;
                ld_agent_x2_issue(src, u0);
                ld_agent_x2_issue(src + 2, u1);
            }
            if (ldw) ld_agent_x2_issue(fl.w16_g + ((size_t)k_blk * CH_PANELS + p) * 256 + 2 * t, uw);
            if (w == p) hand_over();
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(u0), "+v"(u1), "+v"(uw) : : "memory");
            if (i >= 16 * p) {
                *reinterpret_cast<d2*>(Lp + i * WK_S + 4 * mq) = u0;
                *reinterpret_cast<d2*>(Lp + i * WK_S + 4 * mq + 2) = u1;
            }
            if (have) {
                if (t < 256) W16s[(t >> 4) * WK_S + (t & 15)] = pw;
            } else if (t < 128) {
                *reinterpret_cast<d2*>(W16s + (t >> 3) * WK_S + 2 * (t & 7)) = uw;
            }
        }
        if (trc) CH_MARK(5920 + 5 * p + 1);
        barrier_lds();
        if (trc) CH_MARK(5920 + 5 * p + 2);
        {   // solve: rows 16w .. 16w+15, X = R W16'
            double x[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double b0 = W16s[bc * WK_S + 4 * ks + kq], b1 = W16s[(8 + bc) * WK_S + 4 * ks + kq];
                const double a0 = XB[(16 * w + ar) * WK_S + 4 * ks + kq], a1 = XB[(16 * w + 8 + ar) * WK_S + 4 * ks + kq];
                x[0][0] = mfma444(a0, b0, x[0][0]);
                x[0][1] = mfma444(a0, b1, x[0][1]);
                x[1][0] = mfma444(a1, b0, x[1][0]);
                x[1][1] = mfma444(a1, b1, x[1][1]);
            }
            const bool odd = (lane & 1) != 0;
            if (trc) { asm volatile("" :: "v"(x[0][0]), "v"(x[1][1])); CH_MARK(5960 + 3 * p); }
            if (xf) release_wg();   // the stores of the panel BEFORE (issued a whole panel ago) have landed: its flag goes up behind the barrier
            // (BEFORE this panel's stores are issued: they are inline asm the compiler does not count, so its wait for the flag word below
            // would be a wait for them too -- 0.8 us per panel in the trace)
            if (t == 0) {
                if (p + 1 < CH_PANELS && fnext < fl.panel_want) fnext = __hip_atomic_load(fl.panel + k_blk * CH_PANELS + p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *nready = (p + 1 < CH_PANELS && fnext >= fl.panel_want) ? 1 : 0;
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int row = 16 * w + 8 * rb + dr;
                XS[row * WK_S + dc] = -x[rb][0];
                XS[row * WK_S + 8 + dc] = -x[rb][1];
                // neighbouring lanes hold neighbouring columns: the even lane stores both lanes' entries of the left column half, the odd
                // lane both of the right one -- 16-byte agent-scope pieces
                const double got = lane_swap1(odd ? x[rb][0] : x[rb][1]);
                double* Srow = S + ((int64_t)i_tile * TILE + row) * ld + (int64_t)k_blk * TILE + 16 * p + (odd ? 8 + dc - 1 : dc);
                st_agent2(Srow, odd ? got : x[rb][0], odd ? x[rb][1] : got);
            }
            if (trc) CH_MARK(5960 + 3 * p + 1);
        }
        barrier_lds();
        if (trc) CH_MARK(5920 + 5 * p + 3);
        const bool nxt = *nready != 0;
        if (t == 0) fnext = p + 2 < CH_PANELS ? __hip_atomic_load(fl.panel + k_blk * CH_PANELS + p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;   // (lands under the update)
        if (xf && t == 0 && p >= 1 && p <= CH_PANELS - 2) flag_set(xf + p - 1, 1u);
        if (nxt) {   // the next panel's staging loads: in flight during this panel's update
            const int i = t >> 2, mq = t & 3;
            if (i >= 16 * (p + 1)) {
                const double* src = Lkk + (int64_t)i * ld + 16 * (p + 1) + 4 * mq;
                pv0 = ld_agent(src); pv1 = ld_agent(src + 1); pv2 = ld_agent(src + 2); pv3 = ld_agent(src + 3);
            }
            if (t < 256) pw = ld_agent(fl.w16_g + ((size_t)k_blk * CH_PANELS + p + 1) * 256 + t);
        }
        have = nxt;
        if (w > p) {   // wave-uniform: this wave's columns lie beyond the panel
            // eight chunks (contraction step ks, row half): the operands of chunk c + 1 are fetched while the 16 MFMAs of chunk c run
            // (fetched per contraction step in front of its 32 MFMAs, every step waited for its own LDS round trip: 1.3 us for a wave
            // that has its SIMD to itself, against 0.93 us of matrix-pipe time)
            double axA[8], axB[8], bA0, bA1, bB0, bB1;
            const double* lp0 = Lp + (16 * w + bc) * WK_S + kq;
            const double* xs0 = XS + ar * WK_S + kq;
            bA0 = lp0[0]; bA1 = lp0[8 * WK_S];
#pragma unroll
            for (int i = 0; i < 8; ++i) axA[i] = xs0[8 * i * WK_S];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ks = c >> 1, hf = c & 1;
                if (c + 1 < 8) {
                    const int ks1 = (c + 1) >> 1, hf1 = (c + 1) & 1;
                    if (hf == 0) {
                        bB0 = bA0; bB1 = bA1;
#pragma unroll
                        for (int i = 0; i < 8; ++i) axB[i] = xs0[(8 * (8 * hf1 + i)) * WK_S + 4 * ks1];
                    } else {
                        bA0 = lp0[4 * ks1]; bA1 = lp0[8 * WK_S + 4 * ks1];
#pragma unroll
                        for (int i = 0; i < 8; ++i) axA[i] = xs0[(8 * (8 * hf1 + i)) * WK_S + 4 * ks1];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (hf == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        a[2 * i] = mfma444(axA[i], bA0, a[2 * i]);
                        a[2 * i + 1] = mfma444(axA[i], bA1, a[2 * i + 1]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        a[2 * (8 + i)] = mfma444(axB[i], bB0, a[2 * (8 + i)]);
                        a[2 * (8 + i) + 1] = mfma444(axB[i], bB1, a[2 * (8 + i) + 1]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                (void)ks;
            }
        }
        if (xf && p >= CH_PANELS - 2) release_wg();   // the last two panels: what the next pivot block waits for -- drained and flagged at once
        if (trc) CH_MARK(5920 + 5 * p + 4);
        barrier_lds();   // Lp / XS are rewritten by the next panel
        if (xf && t == 0) {
            if (p >= CH_PANELS - 2) flag_set(xf + p, 1u);
            if (k_blk < 24 && i_tile - k_blk <= 2) CH_MARK(3648 + (k_blk * 2 + (i_tile - k_blk - 1)) * 9 + p);
        }
    }
}

// tile (i, k) into the accumulator layout of follow_block: lane pairs fetch 16-byte pieces (even lane: its row's two entries of the left
// column half, odd lane: of the right one) and swap one entry each
__device__ __forceinline__ void load_row_piece(const double* __restrict__ Lmat, int64_t ld, int i_tile, int k_blk,
                                               double (&a)[32]) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int kq = lane >> 4, bb = (lane >> 2) & 3, t4 = lane & 3, dr = 4 * (bb >> 1) + kq, dc = 4 * (bb & 1) + t4;
    const bool odd = (lane & 1) != 0;
    const double* base = Lmat + ((int64_t)i_tile * TILE + dr) * ld + (int64_t)k_blk * TILE + 16 * w + (odd ? 8 + dc - 1 : dc);
    d2 v[16];
#pragma unroll
    for (int rb = 0; rb < 16; ++rb) ld_agent_x2_issue(base + (int64_t)8 * rb * ld, v[rb]);
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                   "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                 :
                 : "memory");
#pragma unroll
    for (int rb = 0; rb < 16; ++rb) {
        const double got = lane_swap1(odd ? v[rb].x : v[rb].y);
        a[2 * rb] = odd ? got : v[rb].x;
        a[2 * rb + 1] = odd ? v[rb].y : got;
    }
}
#else
// x L16' = r  <=>  x = r W16',  W16 = L16^-1 published by the pivot workgroup: 4 of the 16 entries per thread, all 512
// threads, no dependent chain (the substitution by one thread per row -- 16 dependent steps behind two barriers -- took 4 us
// of a 10 us panel).
// Thread t of a follower: wave w = t >> 6 owns columns 16w..16w+15 of tile (i, k); lane (r16 = lane & 15, cg = lane >> 4)
// holds rows r16 + 16 i (i < 8) x columns 16w + 4cg + e (e < 4):  a[4 i + e].
template <int H>
__device__ __forceinline__ void d1_rank16(const double* XB, double (&d)[18]);
template <bool WITH_D1, int H>
__device__ __forceinline__ void follow_block(const double* __restrict__ Lmat, int64_t ld, double* __restrict__ S,
                                             int i_tile, int k_blk, const CholFlags& fl, double* wk, double (&a)[32],
                                             double (&d)[18], unsigned* xf) {
    // xf (rows k+1, k+2 only): raised per panel once its 16 columns of L(i, k) are in S -- the gated update of row k+2 consumes
    // them chunk by chunk while the block is still being factored
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), r16 = lane & 15, cg = lane >> 4;
    double* LPt = wk;                        // [16][WK_LS]
    double* XB = wk + 16 * WK_LS;            // [128][18]  panel entries before the solve
    double* XS = XB + TILE * WK_XS;          // [128][18]  x = the 16 new columns of L(i, k)
    double* W16s = XS + TILE * WK_XS;        // [16][16]
    const double* Lkk = Lmat + (int64_t)k_blk * TILE * (ld + 1);
    const int ty = (t & 255) >> 4, tx = t & 15;
    // A follower that is BEHIND the pivot (its tile arrived late) finds the next panel already published: its staging loads are then
    // issued before this panel's arithmetic and land beside it -- one memory round trip per panel less while catching up (the
    // pivot publishes a panel every ~7 us, a follower needed ~6 us per panel, so a late start was never made up)
    int* nready = reinterpret_cast<int*>(W16s + 256);
    double pv0 = 0.0, pv1 = 0.0, pv2 = 0.0, pv3 = 0.0, pw = 0.0;
    bool have = false;
    for (int p = 0; p < CH_PANELS; ++p) {
        if (t == 0) {
            if (WITH_D1 && k_blk == 1) CH_MARK(5912 + p);   // starts waiting (or finds the panel staged ahead)
            if (!have) flag_wait_ge(fl.panel + k_blk * CH_PANELS + p, fl.panel_want, fl.abort, fl.spin_ticks);
            if (WITH_D1) CH_MARK(1024 + k_blk * CH_PANELS + p);
            *nready = (p + 1 < CH_PANELS && __hip_atomic_load(fl.panel + k_blk * CH_PANELS + p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= fl.panel_want) ? 1 : 0;
        }
        __syncthreads();
        const bool nxt = *nready != 0;
        {   // stage the published panel: rows 16p..127 of L_kk, columns 16p..16p+15 (4 threads per 128-B row segment)
            const int i = t >> 2, mq = t & 3;
            if (i >= 16 * p) {
                double v0 = pv0, v1 = pv1, v2 = pv2, v3 = pv3;
                if (!have) {
                    const double* src = Lkk + (int64_t)i * ld + 16 * p + 4 * mq;
                    d2 u0, u1;
                    ld_agent_x2_issue(src, u0);
                    ld_agent_x2_issue(src + 2, u1);
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(u0), "+v"(u1) : : "memory");
                    v0 = u0.x; v1 = u0.y; v2 = u1.x; v3 = u1.y;
                }
                double* dst = LPt + (4 * mq) * WK_LS + i;
                dst[0] = v0; dst[WK_LS] = v1; dst[2 * WK_LS] = v2; dst[3 * WK_LS] = v3;
            }
            if (have) {
                if (t < 256) W16s[t] = pw;
            } else if (t < 128) {
                d2 u;
                ld_agent_x2_issue(fl.w16_g + ((size_t)k_blk * CH_PANELS + p) * 256 + 2 * t, u);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(u) : : "memory");
                *reinterpret_cast<d2*>(W16s + 2 * t) = u;
            }
        }
        if (nxt) {   // the next panel's staging loads: in flight during this panel's solve and update
            const int i = t >> 2, mq = t & 3;
            if (i >= 16 * (p + 1)) {
                const double* src = Lkk + (int64_t)i * ld + 16 * (p + 1) + 4 * mq;
                pv0 = ld_agent(src); pv1 = ld_agent(src + 1); pv2 = ld_agent(src + 2); pv3 = ld_agent(src + 3);   // (in flight across the panel's phases: loads the compiler tracks)
            }
            if (t < 256) pw = ld_agent(fl.w16_g + ((size_t)k_blk * CH_PANELS + p + 1) * 256 + t);
        }
        have = nxt;
        if (w == p) {   // the wave that holds the panel's columns hands them to the row solvers
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) XB[(r16 + 16 * i) * WK_XS + 4 * cg + e] = a[4 * i + e];
        }
        __syncthreads();
        {   // thread (row = t & 127, part = t >> 7): x[4 part .. 4 part + 3] = sum_k r[k] W16[c][k]
            const int row = t & 127, part = t >> 7;
            double r[16], x4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int c = 0; c < 16; c += 2) {
                const d2 rv = *reinterpret_cast<const d2*>(XB + row * WK_XS + c);
                r[c] = rv.x; r[c + 1] = rv.y;
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const d2* wrow = reinterpret_cast<const d2*>(W16s + (4 * part + cc) * 16);
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    const d2 wv = wrow[k2];
                    x4[cc] += r[2 * k2] * wv.x;
                    x4[cc] += r[2 * k2 + 1] * wv.y;
                }
            }
            double* Srow = S + ((int64_t)i_tile * TILE + row) * ld + (int64_t)k_blk * TILE + 16 * p + 4 * part;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) XS[row * WK_XS + 4 * part + cc] = x4[cc];
            st_agent2(Srow, x4[0], x4[1]);
            st_agent2(Srow + 2, x4[2], x4[3]);
        }
        __syncthreads();
        if (w > p) {   // wave-uniform: this wave's columns lie beyond the panel
            const double* lp = LPt + 16 * w + 4 * cg;
#pragma unroll 2
            for (int m = 0; m < 16; m += 2) {   // two contraction indices per round of LDS reads (x as 16-byte pieces); per entry still m = 0, 1, 2, ...
                d2 xv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) xv[i] = *reinterpret_cast<const d2*>(XS + (r16 + 16 * i) * WK_XS + m);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const d2 l01 = *reinterpret_cast<const d2*>(lp + (m + h) * WK_LS), l23 = *reinterpret_cast<const d2*>(lp + (m + h) * WK_LS + 2);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const double x = h == 0 ? xv[i].x : xv[i].y;
                        a[4 * i + 0] -= x * l01.x;
                        a[4 * i + 1] -= x * l01.y;
                        a[4 * i + 2] -= x * l23.x;
                        a[4 * i + 3] -= x * l23.y;
                    }
                }
            }
        }
        if constexpr (WITH_D1) d1_rank16<H>(XS, d);
        if (xf) release_wg();   // the S stores of this panel were issued two phases ago: they have landed by now
        __syncthreads();   // LPt / XB are rewritten by the next panel
        if (xf && t == 0) { flag_set(xf + p, 1u); if (!WITH_D1 && k_blk < 24 && i_tile - k_blk <= 2) CH_MARK(3648 + (k_blk * 2 + (i_tile - k_blk - 1)) * 9 + p); }
        if (WITH_D1 && t == 0) CH_MARK(2048 + k_blk * CH_PANELS + p);
    }
}

__device__ __forceinline__ void load_row_piece(const double* __restrict__ Lmat, int64_t ld, int i_tile, int k_blk,
                                               double (&a)[32]) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, r16 = lane & 15, cg = lane >> 4;
    const double* base = Lmat + ((int64_t)i_tile * TILE + r16) * ld + (int64_t)k_blk * TILE + 16 * w + 4 * cg;
    d2 v[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ld_agent_x2_issue(base + (int64_t)16 * i * ld, v[2 * i]);
        ld_agent_x2_issue(base + (int64_t)16 * i * ld + 2, v[2 * i + 1]);
    }
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                   "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                 :
                 : "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[2 * i] = v[i].x;
        a[2 * i + 1] = v[i].y;
    }
}

#endif

// ------------------------------------------------------------------------------------------------------------------------
// Pivot role: the Cholesky part of k_potf2_inv (same panels, same look-ahead of the next 16 x 16 pivot block beside the
// trailing update) on the image `a` already in LDS, run by threads 0..255 of the 512-thread owner (the others only keep
// the barriers company).  Instead of one "L out" phase at the end, every finished 16-column panel goes to HBM/L2 at once
// and its flag is raised: that is what the followers are waiting for.
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void publish_panel(double* __restrict__ Lblk, int64_t ld, const double* a, const double* dl,
                                              const double* w16s, double* w16_out, int P, int tid) {
    // columns P..P+15 of L_kk, rows P..127: thread (i = tid & 127, h = tid >> 7) writes 8 columns = 64 B
    const int i = tid & 127, h = tid >> 7;
    if (i >= P) {
        double* dst = Lblk + (int64_t)i * ld + P + 8 * h;
        double v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int c0 = P + 8 * h + c;
            v[c] = (i > c0) ? a[c0 * CH_LD + i] : (i == c0 ? dl[i] : 0.0);
        }
#pragma unroll
        for (int c = 0; c < 8; c += 2) st_agent2(dst + c, v[c], v[c + 1]);
    }
    if (tid < 128) st_agent2(w16_out + 2 * tid, w16s[(tid >> 3) * CH_WLS + 2 * (tid & 7)], w16s[(tid >> 3) * CH_WLS + 2 * (tid & 7) + 1]);   // 256 entries, 16 bytes per store
}

// ---- forward substitution against a 16 x 16 pivot block on LPR lanes per row (used by pivot_block) ----------------------------------
// Lane p of a group of LPR (= 2 or 4) neighbouring lanes holds the entries k = p + LPR j of its row.  The owner of entry C finishes it, a DPP
// quad_perm move hands it to the group, everybody updates the entries it holds.
#ifndef BOHIP_FSUB_LPR
#define BOHIP_FSUB_LPR 4   // measured: 2 lanes per row (four waves, 150 registers of pivot-block entries per lane) 1.36 ms at N = 3000, 4 lanes (seven waves) 1.33-1.35;
                          // since the owner keeps x in its slot the 2-lane form no longer fits 256 registers (it spills, and the chain kernel then times out)
#endif
constexpr int FSUB_LPR = BOHIP_FSUB_LPR, FSUB_NJ = 16 / FSUB_LPR;
static_assert(FSUB_LPR == 4, "the two-lane form needs more than 256 registers (see above)");
template <int S>
__device__ __forceinline__ double group_bcast(double v) {   // the value lane S of this lane's group holds (DPP quad_perm: a VALU move, no LDS)
    constexpr int ctrl = FSUB_LPR == 4 ? (S | (S << 2) | (S << 4) | (S << 6)) : (S | (S << 2) | ((2 + S) << 4) | ((2 + S) << 6));
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, ctrl, 0xf, 0xf, true);   // (old = the source itself: no zero to materialise; every lane has a source lane)
    hi = __builtin_amdgcn_update_dpp(hi, hi, ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// The pivot block's entries a lane needs -- L16[p + LPR j][C] for the steps C that still reach slot j, and 1 / diag -- are ALL fetched before
// the first step (one round of LDS reads): inside the steps nothing but the product, the group broadcast and the updates is left on the
// dependent chain (fetched where they were used, every step waited for its own LDS round trip).
struct GroupL16 {
    double l[16][FSUB_NJ], id[16];
};
__device__ __forceinline__ void group_l16_load(const double* a, const double* idl, int P, int p, GroupL16& L) {
#pragma unroll
    for (int C = 0; C < 16; ++C) {
        L.id[C] = idl[P + C];
#pragma unroll
        for (int j = 0; j < FSUB_NJ; ++j) L.l[C][j] = (FSUB_LPR * j + FSUB_LPR - 1 <= C) ? 0.0 : a[(P + C) * PF_LD + P + p + FSUB_LPR * j];
    }
}
// one row below the pivot block:  x L16' = a[i][P .. P+15].
// Step C: x_C = rw_C / L_CC (owner: lane C % LPR), written to the image as column P + C; then rw_k -= x_C L16[k][C] for every k > C.
template <int C>
__device__ __forceinline__ void group_fsub_row_step(double* a, int P, int i, bool on, int p, double (&rw)[FSUB_NJ], const GroupL16& L) {
    const double x = group_bcast<(C % FSUB_LPR)>(rw[C / FSUB_LPR] * L.id[C]);
    rw[C / FSUB_LPR] = (p == (C % FSUB_LPR)) ? x : rw[C / FSUB_LPR];   // the owner keeps x_C in the slot it no longer needs: written out after the last step
#pragma unroll
    for (int j = 0; j < FSUB_NJ; ++j) {
        if (FSUB_LPR * j + FSUB_LPR - 1 <= C) continue;   // no lane holds an entry k > C in this slot
        if (FSUB_LPR * j > C || p + FSUB_LPR * j > C) rw[j] -= x * L.l[C][j];
    }
    // (keep the steps apart: left to itself the scheduler sinks every update next to its consumer to save registers, and entry C then waits
    // for a chain of C dependent FMAs -- the left-looking form's latency with the right-looking form's code)
    __builtin_amdgcn_sched_barrier(0);
}
template <int... Cs>
__device__ __forceinline__ void group_fsub_row_steps(double* a, int P, int i, bool on, int p, double (&rw)[FSUB_NJ], const GroupL16& L,
                                                     std::integer_sequence<int, Cs...>) {
    (group_fsub_row_step<Cs>(a, P, i, on, p, rw, L), ...);
}
__device__ __forceinline__ void group_fsub_rows(double* a, const double* idl, int P, int i, bool on, int p) {
    double rw[FSUB_NJ];
#pragma unroll
    for (int j = 0; j < FSUB_NJ; ++j) rw[j] = on ? a[i * PF_LD + P + p + FSUB_LPR * j] : 0.0;
    GroupL16 L;
    group_l16_load(a, idl, P, p, L);
    group_fsub_row_steps(a, P, i, on, p, rw, L, std::make_integer_sequence<int, 16>{});
    if (on) {   // x_k for k = p + LPR j: column P + k of the image, row i
#pragma unroll
        for (int j = 0; j < FSUB_NJ; ++j) a[(P + p + FSUB_LPR * j) * PF_LD + i] = rw[j];
    }
}
// column cc of W16 = L16^-1 (right-looking: w_k is final, every later partial sum takes its term at once).  Lane p holds the partial
// sums of the rows i = p + LPR j.
template <int K>
__device__ __forceinline__ void group_fsub_w16_step(double* w16s, int cc, int p, double (&sacc)[FSUB_NJ], const GroupL16& L) {
    const double w_own = (K < cc) ? 0.0 : (K == cc ? L.id[K] : -sacc[K / FSUB_LPR] * L.id[K]);   // (meaningful in the owner lane K % LPR)
    const double wk = group_bcast<(K % FSUB_LPR)>(w_own);
    if (p == (K % FSUB_LPR)) w16s[K * 16 + cc] = wk;
#pragma unroll
    for (int j = 0; j < FSUB_NJ; ++j) {
        if (FSUB_LPR * j + FSUB_LPR - 1 <= K) continue;
        if (FSUB_LPR * j > K || p + FSUB_LPR * j > K) sacc[j] += L.l[K][j] * wk;
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int... Ks>
__device__ __forceinline__ void group_fsub_w16_steps(double* w16s, int cc, int p, double (&sacc)[FSUB_NJ], const GroupL16& L, std::integer_sequence<int, Ks...>) {
    (group_fsub_w16_step<Ks>(w16s, cc, p, sacc, L), ...);
}
__device__ __forceinline__ void group_fsub_w16(const double* a, const double* idl, double* w16s, int P, int cc, int p) {
    double sacc[FSUB_NJ];
#pragma unroll
    for (int j = 0; j < FSUB_NJ; ++j) sacc[j] = 0.0;
    GroupL16 L;
    group_l16_load(a, idl, P, p, L);
    group_fsub_w16_steps(w16s, cc, p, sacc, L, std::make_integer_sequence<int, 16>{});
}

#if BOHIP_PIVOT_W16GROW
// Round 6.  Per panel the critical path was  factor16 (2.7 us) -> [16-step substitution of the rows below the pivot block (1.6 us) beside the
// 16-step inverse W16 on wave 4 (2.1 us)] -> update of the next pivot block (0.5 us) -> factor16 ...: 5.2 us, 41 us per 128-block.  W16 now
// comes out of factor16w itself (kernels_linalg.hip: the inverse's partial sums live in the triangle the Schur complement has left and take
// the same rank-1 update), so the rows below are solved the way the followers solve theirs -- x = r W16', 64 multiply-adds per thread on all
// eight waves, no dependent chain -- and the phase shrinks to ~0.4 us.  W16 is double-buffered in LDS: the publishing waves read panel
// jb's while wave 0 writes panel jb+1's.
__device__ __forceinline__ void pivot_block(double* a, double* dl, double* idl, double* __restrict__ Lblk, int64_t ld,
                                            int k_blk, const CholFlags& fl, int* info, int row0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool act = tid < PF_THREADS;
    unsigned* pflag = fl.panel + k_blk * CH_PANELS;
    double* w16b = idl + TILE;   // [2][16][CH_WLS] inverse of the pivot block: panel jb's in buffer jb & 1
    if (wave == 0) factor16w<CH_LD, CH_WLS>(a, dl, idl, w16b, 0, lane, info, row0);
    __syncthreads();
    for (int jb = 0; jb < CH_PANELS; ++jb) {
        const int P = 16 * jb;
        const int base = P + 16, m = TILE - base;
        const double* w16s = w16b + 16 * CH_WLS * (jb & 1);
        if (m > 0) {
            // Phase A: the rows below the pivot block, X = R W16' (m x 16 x 16), on the matrix pipe: 8-row blocks dealt to the eight waves,
            // one v_mfma_f64_4x4x4 (2 x 2 blocks = 8 x 8 x 4, gemm_core.h) per column half and contraction step.  (As 64 multiply-adds per
            // thread with W16 read as broadcasts it took 1.3 us: an LDS read returns 64 lanes x its width whether or not the lanes share an
            // address -- 32 such 16-byte reads per thread, 0.9 us of LDS time per panel.)  Reads the lower triangle of the image (row i,
            // columns P .. P+15), writes the mirror (column P + c of L, row i): disjoint.
            const int kq = lane >> 4, bb = (lane >> 2) & 3, t4 = lane & 3;
            const int ar = 4 * (bb >> 1) + t4, bc = 4 * (bb & 1) + t4, dr = 4 * (bb >> 1) + kq, dc = 4 * (bb & 1) + t4;
            if (8 * wave < m) {
                double bw[2][4];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) bw[cb][ks] = w16s[(8 * cb + bc) * CH_WLS + 4 * ks + kq];
                for (int rb = wave; 8 * rb < m; rb += 8) {
                    const double* rrow = a + (base + 8 * rb + ar) * CH_LD + P + kq;
                    double av[4], x0 = 0.0, x1 = 0.0;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) av[ks] = rrow[4 * ks];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        x0 = mfma444(av[ks], bw[0][ks], x0);
                        x1 = mfma444(av[ks], bw[1][ks], x1);
                    }
                    a[(P + dc) * CH_LD + base + 8 * rb + dr] = x0;
                    a[(P + 8 + dc) * CH_LD + base + 8 * rb + dr] = x1;
                }
            }
            if (tid == 0 && k_blk == 1) CH_MARK(3584 + 8 * jb + 6);
            __syncthreads();
            if (tid == 0 && k_blk == 1) CH_MARK(3584 + 8 * jb + 0);
        }
        // Phase C.  Who does what follows from where the waves sit: wave w runs on SIMD w % 4, so wave 4 shares its SIMD with wave 0.  The
        // pivot chain (wave 0) gets the lightest neighbour: waves 4, 6, 7 publish (a few LDS reads and stores each, the wait for them to
        // land, +1 each on the panel's flag: followers wait for 3), waves 1, 2, 3, 5 carry the trailing update.
        if (wave == 4 || wave >= 6) {   // 192 threads walk the 256 publishing slots
            const int pw = wave == 4 ? 0 : wave - 5;
            for (int s_ = 64 * pw + lane; s_ < PF_THREADS; s_ += 192)
                publish_panel(Lblk, ld, a, dl, w16s, fl.w16_g + ((size_t)k_blk * CH_PANELS + jb) * 256, P, s_);
            if (lane == 0 && wave == 4 && k_blk == 1) CH_MARK(5904 + jb);   // wave 4 starts waiting for its stores
            release_wg();   // s_waitcnt vmcnt(0): this wave's part of the panel has left the CU
            if (lane == 0) {
                atomicAdd(pflag + jb, 1u);
                if (wave == 4) CH_MARK(k_blk * CH_PANELS + jb);
                if (wave == 6 && k_blk == 1) CH_MARK(5888 + jb);
                if (wave == 7 && k_blk == 1) CH_MARK(5896 + jb);
            }
        }
        if (m == 0) break;
        if (wave == 0) {
            // the rank-16 update of the NEXT pivot block by this wave alone, D = Xn Xn' on the matrix pipe (its own phase on 256 threads and
            // a barrier until round 6: 0.6 us of every panel), then straight into its factorisation (same wave: LDS is in order)
            {
                const int kq = lane >> 4, bb = (lane >> 2) & 3, t4 = lane & 3;
                const int ar = 4 * (bb >> 1) + t4, bc = 4 * (bb & 1) + t4, dr = 4 * (bb >> 1) + kq, dc = 4 * (bb & 1) + t4;
                double d00 = 0.0, d10 = 0.0, d11 = 0.0;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const double* xc = a + (P + 4 * ks + kq) * CH_LD + base;   // column 4 ks + kq of X: the next pivot block's 16 rows
                    const double a0 = xc[ar], a1 = xc[8 + ar], b0 = xc[bc], b1 = xc[8 + bc];
                    d00 = mfma444(a0, b0, d00);
                    d10 = mfma444(a1, b0, d10);
                    d11 = mfma444(a1, b1, d11);
                }
                if (dc <= dr) {
                    a[(base + dr) * CH_LD + base + dc] -= d00;
                    a[(base + 8 + dr) * CH_LD + base + 8 + dc] -= d11;
                }
                a[(base + 8 + dr) * CH_LD + base + dc] -= d10;
                if (tid == 0 && k_blk == 1) CH_MARK(3584 + 8 * jb + 1);
            }
            factor16w<CH_LD, CH_WLS>(a, dl, idl, w16b + 16 * CH_WLS * ((jb + 1) & 1), base, lane, info, row0);
            if (tid == 0 && k_blk == 1) CH_MARK(3584 + 8 * jb + 3);
        } else if (wave <= 3 || wave == 5) {
            const int u = 64 * (wave == 5 ? 3 : wave - 1) + lane;
            trailing_dispatch<true, -1, CH_LD>(a, P, m >> 4, u >> 4, u & 15);
            if (tid == 64 && k_blk == 1) CH_MARK(3584 + 8 * jb + 4);
        }
        __syncthreads();
        if (tid == 0 && k_blk == 1) CH_MARK(3584 + 8 * jb + 2);
    }
}
#else
__device__ __forceinline__ void pivot_block(double* a, double* dl, double* idl, double* __restrict__ Lblk, int64_t ld,
                                            int k_blk, const CholFlags& fl, int* info, int row0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool act = tid < PF_THREADS;
    unsigned* pflag = fl.panel + k_blk * CH_PANELS;
    double* w16s = idl + TILE;   // [16][16] inverse of the current pivot block
    if (wave == 0) factor16(a, dl, idl, 0, lane, info, row0);
    __syncthreads();
    for (int jb = 0; jb < CH_PANELS; ++jb) {
        const int P = 16 * jb;
        const int base = P + 16, m = TILE - base;
        // Phase A: W16 = inverse of the pivot block (for the followers) on wave 4 beside the row solves below the pivot block on the other
        // waves -- LPR = 4 LANES per column of W16 / per row (group_fsub_*: lane p of a group holds the entries p, p + 4, ...; the
        // owner of entry c finishes it, a DPP quad_perm move hands it to the group, everybody updates the entries it holds; two lanes per row
        // were measured too: the same 1.9 us -- a step is ~22 wave instructions either way and issue-bound).  Until
        // round 4 one thread per row / column walked all 16 steps with 136 broadcast LDS reads in its dependent chain: 2.4 us of a 6.15 us
        // panel for a few hundred flops per thread.  Same operations in the same order per entry: the factor does not change by a bit.
        if (wave == 4) {
            if (lane < 16 * FSUB_LPR) group_fsub_w16(a, idl, w16s, P, lane / FSUB_LPR, lane % FSUB_LPR);
            if (tid == PF_THREADS && k_blk == 1) CH_MARK(3584 + 8 * jb + 5);
        } else {
            const int t4 = wave < 4 ? tid : tid - 64, ri = t4 / FSUB_LPR;
            if (t4 < FSUB_LPR * (TILE - 16)) group_fsub_rows(a, idl, P, base + ri, ri < m, t4 % FSUB_LPR);   // (whole waves in or out)
            if (tid == 0 && k_blk == 1) CH_MARK(3584 + 8 * jb + 6);
        }
        __syncthreads();
        if (tid == 0 && k_blk == 1) CH_MARK(3584 + 8 * jb + 0);
        // Panel jb is final.  Two phases follow.  First the rank-16 update of the NEXT pivot block alone, on 256 threads (16 products per
        // thread) -- until round 4 the publishing waves issued their stores in this phase too and everybody waited at its barrier for them:
        // 1.4 us of every 7 us panel.  (The update on wave 0 alone, four elements per lane and no barrier, was measured as well: 2.0 us.)
        // Then: waves 4, 6, 7 publish the panel (stores, the wait for them to land, +1 each on the panel's flag: followers wait for 3), wave 0
        // factors the next pivot block, waves 1, 2, 3, 5 update the rest of the trailing matrix.
        if (m > 0 && act) {
            const int ty = tid >> 4, tx = tid & 15;
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const double* col = a + (P + c) * PF_LD + base;
                acc += col[ty] * col[tx];
            }
            if (tx <= ty) a[(base + ty) * PF_LD + base + tx] -= acc;
        }
        if (m > 0) __syncthreads();
        if (tid == 0 && k_blk == 1) CH_MARK(3584 + 8 * jb + 1);
        // Who does what follows from where the waves sit: wave w runs on SIMD w % 4, so wave 4 shares its SIMD with wave 0.  The pivot
        // chain (wave 0) gets the lightest neighbour: waves 4, 6, 7 publish (a few LDS reads and stores each), waves 1, 2, 3, 5 carry the
        // trailing update.  (With the trailing update on waves 1-4, factor16 took 7.3 us instead of 2.9 beside the first, widest update
        // of a block -- the "slow first panel" of rounds 2-3.)
        if (wave == 4 || wave >= 6) {   // 192 threads walk the 256 publishing slots
            const int pw = wave == 4 ? 0 : wave - 5;
            for (int s_ = 64 * pw + lane; s_ < PF_THREADS; s_ += 192)
                publish_panel(Lblk, ld, a, dl, w16s, fl.w16_g + ((size_t)k_blk * CH_PANELS + jb) * 256, P, s_);
            if (lane == 0 && wave == 4 && k_blk == 1) CH_MARK(5904 + jb);   // wave 4 starts waiting for its stores
            release_wg();   // s_waitcnt vmcnt(0): this wave's part of the panel has left the CU
            if (lane == 0) {
                atomicAdd(pflag + jb, 1u);
                if (wave == 4) CH_MARK(k_blk * CH_PANELS + jb);
                if (wave == 6 && k_blk == 1) CH_MARK(5888 + jb);
                if (wave == 7 && k_blk == 1) CH_MARK(5896 + jb);
            }
        }
        if (m == 0) break;
        if (wave == 0) {
            factor16(a, dl, idl, base, lane, info, row0);
            if (tid == 0 && k_blk == 1) CH_MARK(3584 + 8 * jb + 3);
        } else if (wave <= 3 || wave == 5) {
            // waves 1, 2, 3, 5: one (ty, tx) position of every live 16 x 16 sub-block per thread.  (k_potf2_inv has three waves for this
            // and splits the fourth wave's rows three ways: four code variants per size, 28 in all, ~30 KB.)
            const int u = 64 * (wave == 5 ? 3 : wave - 1) + lane;
            trailing_dispatch<true, -1>(a, P, m >> 4, u >> 4, u & 15);
            if (tid == 64 && k_blk == 1) CH_MARK(3584 + 8 * jb + 4);
        }
        __syncthreads();
        if (tid == 0 && k_blk == 1) CH_MARK(3584 + 8 * jb + 2);
    }
}

#endif
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
            a[i * CH_LD + j] = d[s];
            a[j * CH_LD + i] = 0.0;       // the strict upper triangle is workspace and starts at zero
        } else {
            a[i * CH_LD + j] = (j <= i) ? d[s] : 0.0;
        }
    }
}

// rank-16 update of the diagonal tile held in registers (18 sub-block elements per thread) with the panel rows in XB
template <int H>
__device__ __forceinline__ void d1_rank16(const double* XB, double (&d)[18]) {
    const int t = threadIdx.x, ty = (t & 255) >> 4, tx = t & 15;
    double acc[18];
#pragma unroll
    for (int s = 0; s < 18; ++s) acc[s] = 0.0;
#pragma unroll 2
    for (int m = 0; m < 16; m += 2) {   // two contraction indices per round of LDS reads (16-byte pieces): 128 reads per thread instead of 256
        d2 li[8], lk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            li[u] = *reinterpret_cast<const d2*>(XB + (16 * u + ty) * WK_XS + m);
            lk[u] = *reinterpret_cast<const d2*>(XB + (16 * u + tx) * WK_XS + m);
        }
#pragma unroll
        for (int s = 0; s < 18; ++s) acc[s] += li[blk_bi(2 * s + H)].x * lk[blk_bj(2 * s + H)].x;
#pragma unroll
        for (int s = 0; s < 18; ++s) acc[s] += li[blk_bi(2 * s + H)].y * lk[blk_bj(2 * s + H)].y;
    }
#pragma unroll
    for (int s = 0; s < 18; ++s) d[s] -= acc[s];
}

// ---- role 1 (workgroups 0, 1): row owner.  During block r-1 it only keeps the diagonal tile (r, r) up to date -- in
// registers, one rank-16 update per panel of L(r, r-1), which the critical follower of row r delivers through S -- and
// during block r it runs the pivot chain on that tile.
template <int H>
__device__ __forceinline__ void chain_owner(double* __restrict__ Lmat, int64_t ld, double* __restrict__ S, int T,
                                            const CholFlags& fl, int* info, double* sm, int w) {
    double* a = sm;
    double* dl = sm + TILE * CH_LD;
    double* idl = dl + TILE;
    double* XB = sm;   // [128][18] while the image is not in use
    const int tid = threadIdx.x;
    double d[18];
    for (int r = w; r < T; r += 2) {
        if (r == 0) {   // tile (0, 0) straight from HBM into the image
            const double* Lblk = Lmat;
            for (int e = tid; e < TILE * TILE; e += CH_THREADS) {
                const int i = e >> 7, j = e & 127;
                a[i * CH_LD + j] = (j <= i) ? Lblk[(int64_t)i * ld + j] : 0.0;
            }
            __syncthreads();
        } else {
            if (r >= 2) {   // tile (r, r) carries every update from blocks <= r-2 once the gated update of block r-2 is in memory
                if (tid == 0) { CH_MARK(3300 + 4 * (r - 2) + 3); CH_MARK(4864 + r); flag_wait_ge(fl.crit + (r - 2), fl.crit_want, fl.abort, fl.spin_ticks); CH_MARK(3400 + r); CH_MARK(5120 + r); }
                __syncthreads();
            }
            d1_load<H>(Lmat, ld, r, d);
            if (tid == 0) CH_MARK(3500 + r);
            // (The owner following block r-1 ITSELF -- tile (r, r-1) in its registers beside the diagonal tile, follow_block<true, H>, no
            // hop through S and a second workgroup -- was built and measured in round 4: its panel then takes longer than the pivot's
            // and it falls behind, 1.85 against 1.57 ms at N = 3000.  The split stays: the follower solves, this workgroup only adds.)
            // (Round 4 also tried: the follower publishes the LAST panel's right-hand side one panel early and the owner solves that panel
            // itself from the pivot's flag -- bit-identical, no faster: the owner's own panel 6 ends only ~3 us before the follower's panel 7
            // would have arrived, and its panel 7 then costs 5.4 us instead of 4.  What bounds the hand-over is follower 4-5 us + owner 4 us
            // per panel in a pipeline that runs at the pivot's pace, not the last hop.)
            const unsigned* xf = xp_at(fl, r - 1, r);
            const double* Sx = S + ((int64_t)r * TILE + (tid >> 2)) * ld + (int64_t)(r - 1) * TILE + 4 * (tid & 3);
            for (int p = 0; p < CH_PANELS; ++p) {
                if (tid == 0) { flag_wait_ge(xf + p, 1u, fl.abort, fl.spin_ticks); CH_MARK(1024 + (r - 1) * CH_PANELS + p); if (p == CH_PANELS - 1) CH_MARK(5376 + r); }
                __syncthreads();
                {   // panel p of L(r, r-1): 128 rows x 16 columns, 32 B per thread
                    d2 v0, v1;
                    ld_agent_x2_issue(Sx + 16 * p, v0);
                    ld_agent_x2_issue(Sx + 16 * p + 2, v1);
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1) : : "memory");
                    double* dst = XB + (tid >> 2) * WK_XS + 4 * (tid & 3);
                    *reinterpret_cast<d2*>(dst) = v0;
                    *reinterpret_cast<d2*>(dst + 2) = v1;
                }
                __syncthreads();
                d1_rank16<H>(XB, d);
                __syncthreads();
                if (tid == 0) CH_MARK(2048 + (r - 1) * CH_PANELS + p);
            }
            d1_to_image<H>(a, d);
            __syncthreads();
        }
        if (tid == 0) CH_MARK(3072 + 2 * r);
        pivot_block(a, dl, idl, Lmat + (int64_t)r * TILE * (ld + 1), ld, r, fl, info, r * TILE);
        if (tid == 0) CH_MARK(3072 + 2 * r + 1);
        __syncthreads();
    }
}

// last column of block k's flagged update in mode2
__host__ __device__ __forceinline__ int near_last_col(int T, int k, int win) { return min(T - 1, 4 * (k / 4) + win); }
// waves of the first row tile (row k+3) of block k's flagged update: every workgroup of that row adds 8, with or without a tile
// (host: nt64 = 2 (T - k - 2) columns k+2 ..; mode2: columns k+1 .. near_last_col)
__device__ __forceinline__ unsigned rest_want(const CholFlags& fl, int k) {
    if (fl.mode2 >= 200) return 24u;   // executor form: the six gated-update workgroups of row k+3 inside the chain kernel, four storing waves each
    if (fl.mode2 >= 100) return 48u;   // left-looking form: three launches of two workgroups each hold row k+3's tiles
    return fl.mode2 ? 8u * 2u * (unsigned)(near_last_col(fl.T, k, fl.mode2) - k) : 8u * 2u * (unsigned)(fl.T - k - 2);
}

// ---- role 2 (workgroups 2, 3): critical followers.  Row r is followed twice -- as "row k+2" during block k = r-2 and as
// "row k+1" during block k = r-1 -- always by the workgroup of its parity, so at any block the two rows next to the pivot
// are each in one workgroup's registers and their panels of L reach S (and the flags xp) 16 columns at a time.
__device__ __forceinline__ void crit_follower(double* __restrict__ Lmat, int64_t ld, double* __restrict__ S, int T,
                                              const CholFlags& fl, double* sm, int pf) {
    const int tid = threadIdx.x;
    double ar[32], dd[18];
    for (int k = 0; k + 1 < T; ++k) {
        const int r = (((k + 1) & 1) == pf) ? k + 1 : k + 2;
        if (r >= T) continue;
        if (k >= 1) {   // tile (r, k) must carry block k-1's update: row k+1 gets it from the gated update, row k+2 from the column launch
            if (tid == 0) {
                if (r == k + 1) flag_wait_ge(fl.crit + (k - 1), fl.crit_want, fl.abort, fl.spin_ticks);
                else if (fl.mode2) flag_wait_ge(fl.rest + (k - 1), rest_want(fl, k - 1), fl.abort, fl.spin_ticks);   // tile (k+2, k): first row of block k-1's launch
                else flag_wait_ge(fl.colr + (size_t)(k - 1) * T + r, 8u, fl.abort, fl.spin_ticks);   // tile (k+2, k): its column updaters of block k-1
            }
            __syncthreads();
            if (tid == 0 && r == k + 2) CH_MARK(4352 + k);
            if (tid == 0 && r == k + 1) CH_MARK(5632 + k);
        }
        load_row_piece(Lmat, ld, r, k, ar);
        if (tid == 0 && k < 24) CH_MARK(3648 + (k * 2 + (r - k - 1)) * 9 + 8);
        follow_block<false, 0>(Lmat, ld, S, r, k, fl, sm, ar, dd, xp_at(fl, k, r));
    }
}

// ---- chunk-gated tile update: C (128 x 64) -= A (128 x 128) B' (64 x 128) where the contraction index ARRIVES 16 columns
// at a time: chunk c of A and B is consumed as soon as the followers that produce it raise its panel flags, so the tile is
// final a few microseconds after the block's last panel instead of one launch + one K = 128 GEMM later.  Waves 0-3 compute
// (the contraction engine's fragment layout, gemm_core.h, on an LDS image staged with agent-scope loads); any other
// waves of the workgroup only keep the barriers company.
__device__ __forceinline__ void gated_tile(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C,
                                           int64_t ld, int64_t row0, int64_t col0, const unsigned* fa, const unsigned* fb,
                                           const unsigned* pre, unsigned pre_want, unsigned* signal, const CholFlags& fl,
                                           double* sm) {
    double* As = sm;                       // [128][16]
    double* Bs = sm + TILE * GL_ROW;       // [64][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = (wave & 3) >> 1, wc = wave & 1;
    const bool act = tid < GEMM_THREADS;
    const int kq = lane >> 4, b = (lane >> 2) & 3, t = lane & 3;
    const int r7a = 4 * (b >> 1) + t, r7b = 4 * (b & 1) + t;
    int oa[2], ob[2];
#pragma unroll
    for (int S_ = 0; S_ < 2; ++S_) {
        oa[S_] = ((4 * S_ + kq) ^ r7a) << 1;
        ob[S_] = ((4 * S_ + kq) ^ r7b) << 1;
    }
    const int a_frag = (wr * 64 + r7a) * GL_ROW, b_frag = (wc * 32 + r7b) * GL_ROW;
    // The tile's current value (all updates from earlier blocks: *pre) is fetched BEFORE the contraction, into the accumulators:
    // the launch that writes it finishes early in the block, and the read then overlaps the wait for the panels (read in the
    // epilogue it added ~10 us of scattered agent-scope round trips to the hand-over).
    if (pre) {
        if (tid == 0) flag_wait_ge(pre, pre_want, fl.abort, fl.spin_ticks);
        __syncthreads();
    }
    double acc[8][4];   // holds  -C  so that the contraction adds A B' and the store writes  -(acc)
    if (act) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int r = acc_row(lane, wr, mi);
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
                const int cc = acc_col<4>(lane, wc, nj);
                acc[mi][nj] = (col0 + cc > row0 + r) ? 0.0 : -ld_agent(C + (int64_t)r * ld + cc);
            }
        }
    }
    for (int c = 0; c < TILE / KC; ++c) {
        if (tid == 0) { flag_wait_ge(fa + c, 1u, fl.abort, fl.spin_ticks); flag_wait_ge(fb + c, 1u, fl.abort, fl.spin_ticks); }
        __syncthreads();   // (also: the previous chunk's fragments have been read)
        if (act) {
            // LDS[row][slot] holds the 16-B segment slot ^ (row & 7) of the row's 128-B chunk (the swizzle of gemm_core.h)
            // (six 16-byte agent-scope loads per thread, one wait: as 8-byte atomic loads they were twelve fabric reads)
            d2 va[4], vb[2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sidx = tid + 256 * u, row = sidx >> 3, slot = sidx & 7, seg = slot ^ (row & 7);
                ld_agent_x2_issue(A + (int64_t)row * ld + c * KC + 2 * seg, va[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int sidx = tid + 256 * u, row = sidx >> 3, slot = sidx & 7, seg = slot ^ (row & 7);
                ld_agent_x2_issue(B + (int64_t)row * ld + c * KC + 2 * seg, vb[u]);
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]), "+v"(vb[0]), "+v"(vb[1]) : : "memory");
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sidx = tid + 256 * u, row = sidx >> 3, slot = sidx & 7;
                *reinterpret_cast<d2*>(As + row * GL_ROW + 2 * slot) = va[u];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int sidx = tid + 256 * u, row = sidx >> 3, slot = sidx & 7;
                *reinterpret_cast<d2*>(Bs + row * GL_ROW + 2 * slot) = vb[u];
            }
        }
        __syncthreads();
        if (act) mma_chunk_swz<4>(As + a_frag, Bs + b_frag, oa, ob, acc);
    }
    if (act) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int r = acc_row(lane, wr, mi);
            // Neighbouring lanes hold neighbouring columns: the even lane of a pair takes both lanes' values of column group nj, the odd
            // lane both of group nj + 1 -- 16 stores of 16 bytes per thread instead of 32 of 8 (each one a fabric write of its own,
            // all of them drained before the tile's counter goes up: the hand-over to the next pivot block waits for exactly that)
            const bool odd = (lane & 1) != 0;
#pragma unroll
            for (int nj = 0; nj < 4; nj += 2) {
                const double a0 = -acc[mi][nj], a1 = -acc[mi][nj + 1];
                const double got = __shfl_xor(odd ? a0 : a1, 1);
                const int cc = odd ? acc_col<4>(lane, wc, nj + 1) - 1 : acc_col<4>(lane, wc, nj);   // first column of this lane's piece
                const double x = odd ? got : a0, y = odd ? a1 : got;
                if (col0 + cc > row0 + r) continue;   // the strict upper triangle stays zero
                if (col0 + cc + 1 > row0 + r) st_agent(C + (int64_t)r * ld + cc, x);
                else st_agent2(C + (int64_t)r * ld + cc, x, y);
            }
        }
        release_wg();
        if (lane == 0) atomicAdd(signal, 1u);
    }
    __syncthreads();
}

// waves of the first 128 columns of block k's far update (host: mt = T - k - 3 row tiles x 2 column tiles; every workgroup adds 8)
__device__ __forceinline__ unsigned farcol_want(int T, int k) { return 8u * 2u * (unsigned)(T - k - 3); }

// ---- role 3 (workgroups 4..7): the update of row k+2 with block k's panel -- the two tiles (k+2, k+1), (k+2, k+2) the
// chain needs next (the next owner waited 36 us per block for the launch-based form of this update).
__device__ __forceinline__ void gated_worker(double* __restrict__ Lmat, int64_t ld, const double* __restrict__ S, int T,
                                             const CholFlags& fl, double* sm, int tj) {
    for (int k = 0; k + 2 < T; ++k) {
        const double* A = S + (int64_t)(k + 2) * TILE * ld + (int64_t)k * TILE;
        const double* B = S + ((int64_t)(k + 1) * TILE + (int64_t)tj * CTILE) * ld + (int64_t)k * TILE;
        double* C = Lmat + (int64_t)(k + 2) * TILE * ld + (int64_t)(k + 1) * TILE + (int64_t)tj * CTILE;
        const unsigned* fa = xp_at(fl, k, k + 2);
        gated_tile(A, B, C, ld, (int64_t)(k + 2) * TILE, (int64_t)(k + 1) * TILE + (int64_t)tj * CTILE, fa,
                   tj < 2 ? xp_at(fl, k, k + 1) : fa, k >= 1 ? fl.rest + (k - 1) : nullptr, k >= 1 ? rest_want(fl, k - 1) : 0u,
                   fl.crit + k, fl, sm);
        if (threadIdx.x == 0 && tj == 0) CH_MARK(3300 + 4 * k + 2);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// mode2: one persistent workgroup of the chain kernel turns every finished diagonal block into its inverse W_kk (the inverse half of k_potf2_inv)
// and raises solved[k]: the launch that solves the rows >= k+3 of L(:, k) as a product with W_kk' waits for it.  The same
// blocks are the seeds of the triangular inverse W = L^-1, so k_inv128 is not needed afterwards.
// ------------------------------------------------------------------------------------------------------------------------
// ---- role 4 (workgroup 8, mode2 only; its upper four waves leave at once -- ended waves do not take part in barriers).
// INCREMENTAL: the first version waited for the block's last panel and then ran the recursive-doubling inverse of k_potf2_inv
// (25-30 us behind the pivot block -- on the critical path of every block: pivot -> inverse -> row solve -> update -> chain).
// W = L^-1 can follow the pivot chain row panel by row panel instead:
//     W[p, q] = -W16_p  sum_{q <= m < p} L[p, m] W[m, q]      (16-row panels p, q;  W[p, p] = W16_p, published by the pivot)
// T = L[p, 0:p] W[0:p, 0:p] needs only panels < p (the rows of L behind the pivot arrive with the earlier column panels), so
// it is computed BEFORE panel p's flag arrives; what is left behind the last panel is one 16 x 16 product and the stores:
// solved[k] rises ~3 us after the pivot block ends.  Image: Wimg[m][c] = W[m][c] for the rows m < 112 (zero above the diagonal).
constexpr int INV_WROWS = TILE - 16;
static_assert(INV_LDS_DOUBLES * 8 <= CH_LDS_BYTES, "inverter image must fit the chain's LDS");
// Round p, all 512 threads, thread (c = tid & 127, rq = tid >> 7) owns column c of the rows 4 rq .. 4 rq + 3 of a row panel:
//   [event p: panel p of L_kk and W16_p published]
//   ONE memory round trip: W16_p and the rows of L behind the NEXT pivot block, L[16(p+1) .., 0 .. 16p+15] (complete with event p)
//   W[p, 0:p] = -W16_p T(p)          (T(p) was computed in round p-1; 16 x 16 product per column, exchanged through LDS)
//   T(p+1)    =  L[p+1, 0:p+1] W[0:p+1, 0:p+1]   -- the long product, now BEHIND the publication of row panel p's inputs and
//                                                    before event p+1 arrives (the pivot needs 5-7 us per panel)
//   stores of row panel p (16-byte agent-scope pieces; the last panel straight from registers)
// so what is left behind the block's LAST event is one round trip, a 16 x 16 product and the stores.  (First incremental
// version: T(p) was computed after its own loads, 256 threads, one LDS read per FMA: it fell ~2 us behind per panel over the
// last panels and published the inverse 18-25 us after the pivot block -- no better than the recursive-doubling inverse.)
// Round 6: both products on the matrix pipe.  The round-4 form above (thread (column, row group), operands as LDS reads per multiply-add)
// needed ~49 us per 128-block: fine beside a pivot chain of 58 us per block, the new bound of everything once the chain took 41 -- the
// inverse fell 8 us further behind with every block until the row solves that wait for it stalled the chain (N = 3000: 26, 32, 40, 48, 56,
// 64 us behind the pivot over the first six blocks).  Now
//     W[p, 0:p] = -W16_p T(p)                     2 p column blocks of 8, 8 MFMA each
//     T(p+1)    =  L[p+1, 0:p+1] W[0:p+1, 0:p+1]  2 (p + 1) column blocks, contraction from the block's own rows on (W is lower triangular)
// with the column blocks dealt to the eight waves in pairs (cb, 15 - cb) of equal total contraction length.  Wimg and Tl are read as B
// operands (lane = (contraction index k, column)): their row stride is 128 and the column index is XOR-swizzled with 8 (k & 3), which puts
// the four contraction indices of an operand fetch on different bank groups.
#define INV_WIX(m, c) ((m) * TILE + ((c) ^ (((m) & 3) << 3)))
__device__ __forceinline__ void inverter_role(const double* __restrict__ Lmat, int64_t ld, double* __restrict__ W,
                                              double* __restrict__ WT, int64_t ldw, int T, const CholFlags& fl, double* sm) {
    double* Wimg = sm;                         // [112][128]  W_kk rows 0..111 (zero above the diagonal), swizzled
    double* Tl = Wimg + INV_WROWS * TILE;      // [16][128]   T(p), swizzled
    double* Lrow = Tl + 16 * TILE;             // [16][116]   rows of L_kk behind the next pivot block; at p = 7: the last row panel of W
    double* W16s = Lrow + 16 * INV_LS;         // [16][20]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, bb = (lane >> 2) & 3, t4 = lane & 3;
    const int ar = 4 * (bb >> 1) + t4, bc = 4 * (bb & 1) + t4, dr = 4 * (bb >> 1) + kq, dc = bc;
    for (int e = tid; e < INV_WROWS * TILE; e += CH_THREADS) Wimg[e] = 0.0;   // the strict upper triangle is never written again
    __syncthreads();
    for (int k = 0; k < T; ++k) {
        const int64_t off = (int64_t)k * TILE;
        const double* Lblk = Lmat + off * (ld + 1);
        double* Wb = W + off * (ldw + 1);
        double* WTb = WT + off * (ldw + 1);
        for (int p = 0; p < CH_PANELS; ++p) {
            const int R0 = 16 * p, R1 = R0 + 16;
            const bool last = p + 1 == CH_PANELS;
            if (tid == 0) flag_wait_ge(fl.panel + k * CH_PANELS + p, fl.panel_want, fl.abort, fl.spin_ticks);
            __syncthreads();   // (also: everybody is done with the previous round's W16s / Lrow / row panel)
            {   // ONE round trip of 16-byte agent-scope loads: W16_p and the rows of L behind the NEXT pivot block, columns < 16 (p + 1)
                d2 uw, ul[2];
                uw.x = uw.y = 0.0; ul[0] = uw; ul[1] = uw;
                const int r = tid >> 5, mm = tid & 31;
                const double* src = Lblk + (int64_t)(R1 + r) * ld;
                if (tid < 128) ld_agent_x2_issue(fl.w16_g + ((size_t)k * CH_PANELS + p) * 256 + 2 * tid, uw);
                if (!last) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        if (2 * mm + 64 * i < R1) ld_agent_x2_issue(src + 2 * mm + 64 * i, ul[i]);
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(uw), "+v"(ul[0]), "+v"(ul[1]) : : "memory");
                if (tid < 128) *reinterpret_cast<d2*>(W16s + (tid >> 3) * WK_S + 2 * (tid & 7)) = uw;
                if (!last) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        if (2 * mm + 64 * i < R1) *reinterpret_cast<d2*>(Lrow + r * INV_LS + 2 * mm + 64 * i) = ul[i];
                }
            }
            __syncthreads();
            // the new rows 16p .. 16p+15 of W, columns < 16p: -W16 T(p)
            for (int cbi = 0; cbi < 2; ++cbi) {
                const int cb = cbi == 0 ? wave : 15 - wave;
                if (cb >= 2 * p) continue;
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int kk = 4 * ks + kq;
                    const double b = Tl[INV_WIX(kk, 8 * cb + bc)];
                    a0 = mfma444(W16s[ar * WK_S + kk], b, a0);
                    a1 = mfma444(W16s[(8 + ar) * WK_S + kk], b, a1);
                }
                if (!last) {
                    Wimg[INV_WIX(R0 + dr, 8 * cb + dc)] = -a0;
                    Wimg[INV_WIX(R0 + 8 + dr, 8 * cb + dc)] = -a1;
                } else {
                    Lrow[dr * INV_LS + 8 * cb + dc] = -a0;
                    Lrow[(8 + dr) * INV_LS + 8 * cb + dc] = -a1;
                }
            }
            if (!last && tid < 256) Wimg[INV_WIX(R0 + (tid >> 4), R0 + (tid & 15))] = W16s[(tid >> 4) * WK_S + (tid & 15)];   // the diagonal block is W16 itself
            __syncthreads();
            if (!last) {
                // T(p+1), columns < 16 (p + 1): column block cb contracts over the rows m >= 8 cb
                for (int cbi = 0; cbi < 2; ++cbi) {
                    const int cb = cbi == 0 ? wave : 15 - wave;
                    if (cb >= 2 * p + 2) continue;
                    double a0 = 0.0, a1 = 0.0;
                    for (int ks = 2 * cb; ks < R1 / 4; ++ks) {
                        const int kk = 4 * ks + kq;
                        const double b = Wimg[INV_WIX(kk, 8 * cb + bc)];
                        a0 = mfma444(Lrow[ar * INV_LS + kk], b, a0);
                        a1 = mfma444(Lrow[(8 + ar) * INV_LS + kk], b, a1);
                    }
                    Tl[INV_WIX(dr, 8 * cb + dc)] = a0;
                    Tl[INV_WIX(8 + dr, 8 * cb + dc)] = a1;
                }
            }
            // row panel p out: W rows (16-byte pieces along the row) and the same entries as columns of W'
            {
                auto wval = [&](int rr, int c) -> double {   // W[16p + rr][c], c < 16 (p + 1)
                    if (!last) return Wimg[INV_WIX(R0 + rr, c)];
                    return c < R0 ? Lrow[rr * INV_LS + c] : W16s[rr * WK_S + (c - R0)];
                };
                for (int e = tid; e < 16 * (R1 / 2); e += CH_THREADS) {   // rows of W
                    const int rr = e / (R1 / 2), col = 2 * (e % (R1 / 2));
                    d2 v;
                    v.x = wval(rr, col); v.y = wval(rr, col + 1);
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(Wb + (int64_t)(R0 + rr) * ldw + col), "v"(v) : "memory");
                }
                for (int e = tid; e < 8 * R1; e += CH_THREADS) {   // columns of W': row c of W', entries 16p + 2 j, 16p + 2 j + 1
                    const int c = e % R1, j = e / R1;
                    d2 v;
                    v.x = wval(2 * j, c); v.y = wval(2 * j + 1, c);
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(WTb + (int64_t)c * ldw + R0 + 2 * j), "v"(v) : "memory");
                }
            }
        }
        release_wg();
        __syncthreads();
        if (tid == 0) { flag_set(fl.solved + k, 1u); CH_MARK(4096 + k); }
    }
}

// ---- role 5 (workgroups 9 .. 9 + nsf - 1, executor form only): the row solves of the tiles (k+3, k) .. (k+2+nsf, k), panel
// by panel behind the pivot chain like the critical followers, so that S(k+3, k) -- the operand of the three tiles the chain
// waits for next -- is complete ~3 us after the pivot block instead of one inverse (~20 us) + one solve task (~20-40 us) later.
// Follower f needs tile (k+3+f, k) final, i.e. a row of the executor's Late(k-1), which in turn needs S(k+3+f, k-1): for all
// but the last follower that is the NEXT follower's output of the block before -- also chain-internal and early -- so only the
// last follower's input hangs on the executor's inverse -> Solve -> Late path, and that one has nsf blocks of slack (a late
// start is caught up within the block: a panel takes a follower ~3 us, the pivot ~7 us).
constexpr int CH_NSF_MAX = 6;   // (the number in use is CholFlags::nsf, BOHIP_CHOL_NSF)
__device__ __forceinline__ void solve_follower(const double* __restrict__ Lmat, int64_t ld, double* __restrict__ S, int T,
                                               const CholFlags& fl, double* sm, int f) {
    const int tid = threadIdx.x;
    double ar[32], dd[18];
    for (int k = 0; k + 3 + f < T; ++k) {
        const int r = k + 3 + f;
        if (k >= 1) {   // ver(r, k) of kernels_exec.hip: every read-modify-write round of the tile is in memory
            const int nb = k / 4 - 1 > 0 ? k / 4 - 1 : 0;
            if (tid == 0) flag_wait_ge(fl.xp + ((size_t)(k - 1) * T + r) * CH_PANELS, 16u * (unsigned)(nb + 1), fl.abort, fl.spin_ticks);
            __syncthreads();
        }
        if (tid == 0 && f < 3) CH_MARK(7168 + 256 * f + k);
        load_row_piece(Lmat, ld, r, k, ar);
        follow_block<false, 0>(Lmat, ld, S, r, k, fl, sm, ar, dd, (f == 0 && fl.mode2 >= 200) ? fl.xp3 + (size_t)k * CH_PANELS : nullptr);
        release_wg();
        __syncthreads();
        if (tid == 0) { flag_set(fl.colr + (size_t)k * T + r, 16u); if (f == 0) CH_MARK(4608 + k); if (f < 3) CH_MARK(6144 + 256 * f + k); }   // sver(r, k)
    }
}

// ---- role 6 (workgroups 9 + nsf .. 9 + nsf + 5, executor form): the chunk-gated update of the three tiles of row k+3 --
// (k+3, k+1), which row k+3's critical follower solves during block k+1, and (k+3, k+2), (k+3, k+3), which block k+1's gated
// updates continue -- with block k, consumed 16 columns at a time as solve follower 0 publishes S(k+3, k): the chain's window
// spans three rows.  As executor tasks (Late(k): blocks k-1 and k, started when S(k+3, k) was complete) these tiles reached the
// chain 33-50 us after the pivot block on an idle chip and 80-250 us under load, and row k+3's follower, which needs ~6 us per
// panel against the pivot's ~7, never made a late start up: chain period = 69 us + that latency - 29.  Now the executor
// delivers the tiles with every block BEFORE k (pre3: one block of slack) and block k is added here, final ~3 us after the
// solve follower's last panel.
__device__ __forceinline__ void gated_worker3(double* __restrict__ Lmat, int64_t ld, const double* __restrict__ S, int T,
                                              const CholFlags& fl, double* sm, int tj) {
    const int j = tj >> 1, h = tj & 1;
    for (int k = 0; k + 3 < T; ++k) {
        const int c = k + 1 + j;
        const double* A = S + (int64_t)(k + 3) * TILE * ld + (int64_t)k * TILE;
        const double* B = S + ((int64_t)c * TILE + (int64_t)h * CTILE) * ld + (int64_t)k * TILE;
        double* C = Lmat + (int64_t)(k + 3) * TILE * ld + (int64_t)c * TILE + (int64_t)h * CTILE;
        const unsigned* fa = fl.xp3 + (size_t)k * CH_PANELS;
        gated_tile(A, B, C, ld, (int64_t)(k + 3) * TILE, (int64_t)c * TILE + (int64_t)h * CTILE, fa, j < 2 ? xp_at(fl, k, c) : fa,
                   k >= 1 ? fl.pre3 + 4 * k + j : nullptr, 16u, fl.rest + k, fl, sm);
        if (threadIdx.x == 0 && tj == 0) CH_MARK(4352 + 256 * 14 + k);   // [7936, 8192): row k+3's tiles carry block k
    }
}

// The persistent chain: 8 workgroups of 512 threads, one per CU (the owners' LDS image fills it).
__global__ __launch_bounds__(CH_THREADS, 1) void k_chol_chain(double* __restrict__ Lmat, int64_t ld, double* __restrict__ S,
                                                           int T, CholFlags fl, int* __restrict__ info, double* __restrict__ W,
                                                           double* __restrict__ WT) {
    extern __shared__ double sm[];
    const int b = blockIdx.x;
    if (threadIdx.x == 0) atomicAdd(fl.resident, 1u);
    if (b >= 9 + fl.nsf && fl.mode2 >= 200) {
        gated_worker3(Lmat, ld, S, T, fl, sm, b - 9 - fl.nsf);
    } else if (b >= 9) {
        solve_follower(Lmat, ld, S, T, fl, sm, b - 9);
    } else if (b == 8) {
        inverter_role(Lmat, ld, W, WT, ld, T, fl, sm);
    } else if (b < 2) {
        if ((threadIdx.x >> 8) == 0) chain_owner<0>(Lmat, ld, S, T, fl, info, sm, b);
        else chain_owner<1>(Lmat, ld, S, T, fl, info, sm, b);
    } else if (b < 4) {
        crit_follower(Lmat, ld, S, T, fl, sm, b - 2);
    } else {
        gated_worker(Lmat, ld, S, T, fl, sm, b - 4);
    }
}

// Rows >= 3 before they reach the chain.  One PERSISTENT workgroup per row i follows blocks k = 0 .. i-3 (tile (i, k) each:
// the same panel follower as the critical ones), and two persistent workgroups per row keep tile (i, k+1) -- the tile the
// row follows NEXT -- updated chunk by chunk (gated_tile).  A row's work for block k+1 can therefore start a few
// microseconds after its work for block k ends; with one launch per block for each of the two steps (the first version)
// the cycle follower -> launch gap -> K = 128 update -> launch gap was 80 us against a 58 us pivot block and paced everything.
__global__ __launch_bounds__(CH_THREADS, 1) void k_chol_rows(const double* __restrict__ Lmat, int64_t ld, double* __restrict__ S,
                                                          int T, CholFlags fl) {
    extern __shared__ double sm[];
    const int i_tile = 3 + blockIdx.x;
    if (threadIdx.x == 0) atomicAdd(fl.resident, 1u);
    double ar[32], d[18];
    for (int k = 0; k + 3 <= i_tile; ++k) {
        if (k >= 1) {   // tile (i, k) carries block k-1's update once its two column updaters have stored
            if (threadIdx.x == 0) flag_wait_ge(fl.colr + (size_t)(k - 1) * T + i_tile, 8u, fl.abort, fl.spin_ticks);
            __syncthreads();
        }
        load_row_piece(Lmat, ld, i_tile, k, ar);
        follow_block<false, 0>(Lmat, ld, S, i_tile, k, fl, sm, ar, d, xp_at(fl, k, i_tile));
    }
}
__global__ __launch_bounds__(GEMM_THREADS) void k_chol_cols(double* __restrict__ Lmat, int64_t ld, const double* __restrict__ S,
                                                         int T, CholFlags fl) {
    __shared__ __attribute__((aligned(16))) double sm[(TILE + CTILE) * GL_ROW];
    const int i_tile = 3 + (blockIdx.x >> 1), h = blockIdx.x & 1;
    if (threadIdx.x == 0) atomicAdd(fl.resident, 1u);
    for (int k = 0; k + 3 <= i_tile; ++k) {   // tile (i, k+1), columns [64 h, 64 h + 64): -= L(i, k) L(k+1, k)'
        const double* A = S + (int64_t)i_tile * TILE * ld + (int64_t)k * TILE;
        const double* B = S + ((int64_t)(k + 1) * TILE + (int64_t)h * CTILE) * ld + (int64_t)k * TILE;
        double* C = Lmat + (int64_t)i_tile * TILE * ld + (int64_t)(k + 1) * TILE + (int64_t)h * CTILE;
        // block k-1's far update wrote this tile too (column k+1 was its FIRST column, dispatched early, own counter)
        gated_tile(A, B, C, ld, (int64_t)i_tile * TILE, (int64_t)(k + 1) * TILE + (int64_t)h * CTILE, xp_at(fl, k, i_tile),
                   xp_at(fl, k, k + 1), k >= 1 ? fl.farall + (k - 1) : nullptr, k >= 1 ? farcol_want(T, k - 1) : 0u,
                   fl.colr + (size_t)k * T + i_tile, fl, sm);
    }
}

// Form 1's flagged launches (k_gemm_nt, one per block, hundreds of workgroups that spin inside the kernel) must not reach the chip before
// every PERSISTENT workgroup is resident: a panel follower holds 2 x 213 of a SIMD's 512 vector registers since round 6 (round 5: 2 x 160)
// and no longer fits beside even one k_gemm_nt workgroup, so when the first flagged launch was dispatched between the persistent
// kernels -- they sit on four streams released by one event -- some follower found no CU, the spinning workgroups waited for its rows,
// and the factorisation ended in the 200 ms time-out (4 of 8 fresh processes at 24 row tiles, none at 16).  One wave on the flagged
// launches' stream, in front of them, waits until the persistent workgroups have counted themselves in.
__global__ __launch_bounds__(64) void k_chol_gate(const unsigned* word, unsigned want, unsigned* abort, unsigned long long spin_ticks) {
    if (threadIdx.x == 0) flag_wait_ge(word, want, abort, spin_ticks);
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
