// kernels_exec.hip -- the tile-task executor of the blocked Cholesky factorisation AND of the triangular inverse W = L^-1 (row A2 of
// SURVEY.md section 8; reference call sites: update!/fit! at src/models/gp.jl:11-18, i.e. the LAPACK potrf behind ElasticPDMats).
//
// Why.  The second dataflow form (cholesky_dataflow2/3 in bohip.hip) feeds the persistent chain (k_chol_chain) from ~4 launches
// per 128-block on three streams: every flagged launch parks hundreds of workgroups that spin on flags inside the kernel (at
// N = 10^4 about 300 of the chip's 512 workgroup slots, which is why the K = 512 bulk updates beside them ran at 37 TF/s), the
// in-order streams put a 1 ms bulk launch in front of the 0.1 ms one the chain needs next, and the left-looking column update
// (K up to 1024, one workgroup per tile) sat on the critical path of every block: 11.7 ms for 333 GF.
//
// Here everything outside the chain is a TASK = one 128 x 64 half tile  C = C - A B' - P,  C = A B'  or  C = -A B'  (optionally
// stored a second time, transposed) on the contraction engine of gemm_core.h, described by an immutable 128-byte record the host
// writes once per (handle, T).  ONE persistent kernel (k_chol_exec, 512 workgroups = two per CU) executes them: a workgroup that
// is free looks at the heads of six in-order queues, highest priority first, and claims from the first one whose head is
// runnable with a fetch-and-add; a record it got a little ahead of its counters is HELD and polled (see k_chol_exec).  No launch
// boundary and no host event sits between dependent tasks, and progress does not depend on how many executor workgroups are
// resident.
//
//   queue 0  URGENT, per block k: Late(k) of the tiles the chain kernel itself reads next -- (k+4 .. k+2+nsf, k+1), which its solve
//            followers read, and the three tiles of row k+4 that its gated updates finish during block k+1 -- and the row steps of
//            the next two rows.  Their own queue and their own 32 workgroups: queued behind the far rows of the block before
//            (which wait for that block's inverse and row solves) they were claimed 40-80 us late, and a follower that starts
//            late never catches up
//   queue 1  per block k:  Solve(i, k) = A(i, k) W_kk'  for the rows the chain does not solve itself (W_kk: the chain's inverter
//            workgroup), then Late(k) of column k+1 for those rows: blocks k-1 and k (K = 256) and the pre-summed older blocks P
//   queue 2  Early(k): P(i, c) = sum over the window's blocks up to k of S(i, b) S(c, b)' for the tiles Late(k+2) will finish
//            (K = 128 .. 640; depends on S only, two blocks of slack) -- takes the long contraction off the critical path
//   queue 3  the triangular inverse W = L^-1 (what the scoring path contracts with), ROW CHAIN.  W(i, j) = W_ii Z(i, j),
//            Z(i, j) = -sum_{k=j}^{i-1} L(i, k) W(k, j): row i needs row i of L (final one block before pivot i) and the rows
//            < i of W, so its work (~ i^2) becomes available as the factorisation's own work (~ (T-k)^2 per block) dries up: the
//            two fill each other's idle time.  This queue holds what is serial from row to row: the last pieces of Z (the last
//            one also stores Z' to the mirror tile of W -- strictly upper, read by nobody else) and the product with W_ii, which
//            stores W(i, j) and the same entries as W'(j, i).  Little work: ABOVE the bulk, or it starts when the bulk ends
//   queue 4  bulk: group m (blocks 4m .. 4m+3, K = 512) applied to every tile of the columns >= 4m+8, column-major, so that
//            the four columns the chain reaches next are done first
//   queue 5  the inverse's WAVES: chunk m of the contraction (G blocks) pushed to every row below it as soon as the chunk's rows
//            of W are final -- most of the inverse's flops, thousands of independent tasks, lowest priority.  Nothing outside
//            the queues 3 and 5 ever waits for them
//
// Dependencies are counters in the flag area (one word per producer granule, each finished task adds 8 = its waves):
//   ver(i, c)   read-modify-write rounds completed on tile (i, c): bulk group m needs 16 m, Late needs 16 (number of groups)
//   pver(i, c)  P(i, c) is complete (16)
//   sver(i, k)  S(i, k) is complete (16)          [the words colr[k T + i] of CholFlags]
//   solved[k], xp[..][7]   raised by the chain (inverse of the diagonal block; rows k+1, k+2 of L(:, k))
//   pre3[4k+j]  16 = tile (k+3, k+1+j) carries every block before k (the executor's share): what the chain kernel's gated updates
//               of row k+3 wait for before they add block k; THEY raise rest[k], which the chain's followers of block k+1 wait for
//   zver / zt / wfin(i, j)   the inverse: rounds completed on Z(i, j) (16 each), Z' stored, W(i, j) and W'(j, i) final
// Every datum a running kernel reads is written with agent-scope (sc1, write-through) 16-byte stores behind an explicit
// s_waitcnt vmcnt(0); the read-modify-write operands (old tile value, P) are fetched with sc1 loads, which bypass the CU's
// vector L1 -- the executor lives for the whole factorisation, so no kernel boundary ever invalidates that cache.  Operand
// tiles that go through LDS-DMA (S, the finished A(i, k), W_kk, Z', W') are never written again once a task has read them that
// way (tests/test_exec_tasks.py asserts it for every record).
#include "gemm_core.h"

namespace bohip {

constexpr unsigned EX_NONE = 0xffffffffu;
constexpr int EX_NDEP = 6;
constexpr int EX_TRACE_CAP = 1 << 18;   // (trace build only)
constexpr int EX_NQ = 6;
constexpr int EX_QROWS = 3;     // the triangular inverse W = L^-1 grown behind the chain: its row-to-row chain (little work, but serial: ABOVE the bulk)
constexpr int EX_QBULK = 4;     // bulk updates
constexpr int EX_QWAVE = 5;     // the inverse's waves (most of its work, independent tasks: lowest priority, and what a waiting workgroup fills in)

struct ExTask {                 // 128 bytes, written by the host once per (handle, T), never modified on the device
    const double* A;            // [128][16 kc]  K-major, row stride ld
    const double* B;            // [64][16 kc]
    double* C;                  // [128][64]
    const double* P;            // optional: C -= P as well; with rmw bit 2: NOT an operand but the [64][128] block that receives the transposed result
    uint32_t dep_idx[EX_NDEP];  // word index into the flag area, EX_NONE = unused
    uint32_t dep_want[EX_NDEP];
    uint32_t sig_idx[2];        // counters every wave adds 1 to once its stores have landed
    int32_t kc;                 // 16-deep contraction chunks
    int32_t diag_h;             // -1, or this task is half `diag_h` of a DIAGONAL tile: entries with 64 h + c > r stay untouched
    int32_t rmw;                // bits 0-1: 1: C = C - A B' - P;  0: C = A B';  2: C = -A B'.   bit 2: the result also goes out transposed, to P
    int32_t prio;               // s_setprio of the task's waves: the chain waits for queue 0, so its tasks win the matrix pipe of a shared CU
    int32_t kc_split;           // > 0: the contraction runs in two pieces, chunks [0, kc_split) at once and [kc_split, kc) once the
    uint32_t dep2_idx[2];       //      counters dep2 have arrived (waited for INSIDE the task: the three tiles the chain waits for start on
    uint32_t dep2_want[2];      //      the previous block's half of their K = 256 while the row solve of the current block is still running)
    int32_t pad[1];
};
static_assert(sizeof(ExTask) == 128, "task record layout");

struct ExQueues {
    const ExTask* tasks;
    int qbeg[EX_NQ + 1];        // queue q = tasks[qbeg[q] .. qbeg[q+1])
    unsigned* flags;            // the flag area of the factorisation (CholFlags arrays live in it)
    unsigned* heads;            // [EX_NQ] queue cursors (zeroed with the flags)
    unsigned* abort;
    int64_t ld;
    unsigned long long spin_ticks;   // a workgroup that finds no runnable task for this long gives up (see flag_wait_ge)
    int nurgent;                // workgroups 0 .. nurgent-1 serve the urgent queue (and nothing else until it is exhausted)
    int second_from;            // > 0: workgroups from this index on take only early sums, bulk and wave tasks (see k_chol_exec)
    int nfast;                  // the next nfast workgroups never take bulk or wave tasks: whatever the chain will need soon (queues 1-3) finds
                                // one of them free.  Chain-paced sizes only: with every general workgroup inside a 60-100 us bulk / wave task
                                // right after a group's release, the row steps waited that long and the pivot chain with them
    int fill_inv;               // ... and inverse-wave work (1), see k_chol_exec
    unsigned patience_ticks;    // ... but only once the held record has been waited for this long (wall-clock ticks of 10 ns)
    int fill;                   // a workgroup that holds a claimed task whose counters are not in yet takes bulk work meanwhile:
                                // 1 = if the held task is an Early sum (queue 2: two blocks of slack), 2 = also for queue 1, 0 = never
    int stride[EX_NQ];          // records per claim: 1, or 2 = both halves of a tile run back to back by one workgroup (the look and
                                // claim between two tasks cost ~12 us against ~70 us of work: queues 1 and 2 are claimed in pairs)
};

// Claim the next task.  Run by the 64 lanes of wave 0: lane 8 q + d looks at dependency d of the head of queue q, so the heads
// of all queues are examined in one round of parallel loads (a one-lane version walked ~20 dependent memory round trips
// per look, and as every free workgroup looked at the SAME head and only one compare-and-swap could win, the claims were
// serialised at ~0.2 per microsecond: first light of this kernel ran at 1.2 TF/s).  A queue whose head is runnable is claimed
// with ONE fetch-and-add; under contention the claimer gets a task a little behind the head it looked at and then waits for
// THAT task's counters (tasks are claimed in queue order, so everything a claimed task depends on is already claimed by a
// running workgroup or belongs to the chain: the wait is bounded and cannot dead-lock).  Returns -1 when every queue is
// exhausted, or on abort.
__device__ __forceinline__ bool ex_dep_pending(const ExTask* t, int d, const unsigned* flags) {
    const uint32_t idx = t->dep_idx[d];
    return idx != EX_NONE && __hip_atomic_load(flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < t->dep_want[d];
}
// are the counters of a claim (`ntile` tiles = 2 ntile consecutive records; the two halves of a tile share their counters) all in?
// wave 0: lane 8 j + d looks at counter d of tile j
__device__ __forceinline__ bool ex_ready_once(const ExQueues& q, const ExTask* t, int ntile, int lane) {
    const bool p2 = (lane >> 3) < ntile && (lane & 7) < EX_NDEP && ex_dep_pending(t + 2 * (lane >> 3), lane & 7, q.flags);
    return __ballot(p2) == 0ull;
}
__device__ __forceinline__ int ex_ntile(const ExQueues& q, int qi, unsigned c) {
    const unsigned nq = (unsigned)(q.qbeg[qi + 1] - q.qbeg[qi]), st = (unsigned)q.stride[qi];
    return st <= 2u ? 1 : (int)(((nq - c < st ? nq - c : st) + 1u) / 2u);
}
// wait until every counter of a claim has arrived (wave 0).  false: abort / time-out.
__device__ __forceinline__ bool ex_wait_task(const ExQueues& q, const ExTask* t, int ntile, int lane) {
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        if (ex_ready_once(q, t, ntile, lane)) return true;
        if (__hip_atomic_load(q.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        if (wall_clock64() - t0 > q.spin_ticks) {
            if (lane == 0) atomicCAS(q.abort, 0u, 1u);
            return false;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}
// Claim from queue qi (its head was seen runnable) with ONE fetch-and-add: the first record of the claim, or -2 if the queue ran dry
// meanwhile.  Under contention the claimer gets a task a little behind the head it looked at; `ready` tells whether THAT task's
// counters are in (looked at once, not waited for: see the caller).
__device__ __forceinline__ int ex_claim(const ExQueues& q, int qi, unsigned h_seen, int lane, bool& ready) {
    unsigned c = 0;
    if (lane == 0) c = atomicAdd(q.heads + qi, (unsigned)q.stride[qi]);
    c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
    if (c >= (unsigned)(q.qbeg[qi + 1] - q.qbeg[qi])) return -2;
    const int ntile = ex_ntile(q, qi, c);
    ready = (c == h_seen && ntile == 1) || ex_ready_once(q, q.tasks + q.qbeg[qi] + c, ntile, lane);
    return q.qbeg[qi] + (int)c;
}
// Look at the heads of the queues in `qmask` (lane 8 q + d: counter d of queue q's head record) and claim from the first one
// whose head is runnable.  wait = true: stay until something could be claimed; -1 = every queue exhausted (or abort).
// wait = false: one look; -3 = nothing runnable right now.
__device__ __forceinline__ int ex_pick(const ExQueues& q, int lane, bool& ready, unsigned qmask, bool wait) {
    const int qi = lane >> 3, d = lane & 7;
    const unsigned long long t_idle = wall_clock64();
    int backoff = 1;
    for (;;) {
        unsigned h = 0, n = 0;
        if (qi < EX_NQ) {
            n = (unsigned)(q.qbeg[qi + 1] - q.qbeg[qi]);
            h = __hip_atomic_load(q.heads + qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool live = qi < EX_NQ && ((qmask >> qi) & 1u) != 0u && h < n;
        const bool pending = live && d < EX_NDEP && ex_dep_pending(q.tasks + q.qbeg[qi] + h, d, q.flags);
        const unsigned long long lv = __ballot(live), pd = __ballot(pending);
        if (lv == 0ull) return wait ? -1 : -3;
        int pickq = -1;
#pragma unroll
        for (int c = EX_NQ - 1; c >= 0; --c)
            if (((lv >> (8 * c)) & 1ull) && ((pd >> (8 * c)) & 0xffull) == 0ull) pickq = c;
        if (pickq >= 0) {
            const unsigned hs = (unsigned)__builtin_amdgcn_readlane((int)h, 8 * pickq);
            const int r = ex_claim(q, pickq, hs, lane, ready);
            if (r != -2) return r;
            continue;   // the queue ran dry between the look and the claim
        }
        if (!wait) return -3;
        if (__hip_atomic_load(q.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return -1;
        if (wall_clock64() - t_idle > q.spin_ticks) {   // no runnable task for this long: something upstream never arrived
            if (lane == 0) atomicCAS(q.abort, 0u, 1u);
            return -1;
        }
        for (int s_ = 0; s_ < backoff; ++s_) __builtin_amdgcn_s_sleep(16);
        if (backoff < 8) backoff *= 2;
    }
}

#if BOHIP_CHOL_TRACE
__device__ unsigned long long g_ex_trace[EX_TRACE_CAP * 8];   // per task: looking since | claimed and runnable | finished | workgroup | second-stage counters in | main loop done
#endif
// The look for the NEXT task rides on this task's epilogue (do_look): wave 0 reads the queue heads while the tile goes through
// LDS, the head records' counters' names while the old tile value is fetched, the counters while the stores drain -- three
// dependent memory round trips that used to sit between two tasks.  What comes out is a few microseconds stale, which is
// harmless: a head seen runnable stays runnable, and one that became runnable meanwhile is seen by the next free workgroup
// (one frees up every ~0.1 us).  Returns 8 * queue + ... packed as (queue << 24) | head, or -1 if no head was runnable.
__device__ __forceinline__ int ex_run(const ExTask& t, const ExQueues& q, double* smem, int tid, bool do_look, int trace_slot = -1) {
    const int64_t ld = q.ld;
    unsigned* flags = q.flags;
    unsigned* abort = q.abort;
    const unsigned long long spin_ticks = q.spin_ticks;
    double acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    {
        const int pr = t.prio;   // (s_setprio takes an immediate)
        if (pr >= 2) __builtin_amdgcn_s_setprio(3);
        else if (pr == 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
    const int ksp = t.kc_split;
    if (ksp > 0) {
        gemm_tile_loop_glds3_ks<4, 0, true, false>(t.A, ld, t.B, ld, 0, ksp, smem, acc, TILE, 1 << 30, tid);
        if (tid == 0) {   // (tasks are claimed in queue order: what this waits for is already claimed by a running workgroup, or the chain's)
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int d = 0; d < 2; ++d)
                    if (t.dep2_idx[d] != EX_NONE && __hip_atomic_load(flags + t.dep2_idx[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < t.dep2_want[d]) ok = false;
                if (ok || __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (wall_clock64() - t0 > spin_ticks) { atomicCAS(abort, 0u, 1u); break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
#if BOHIP_CHOL_TRACE
        if (tid == 0 && trace_slot >= 0 && trace_slot < EX_TRACE_CAP) g_ex_trace[8 * trace_slot + 4] = wall_clock64();
#endif
    }
    gemm_tile_loop_glds3_ks<4, 0, true>(t.A, ld, t.B, ld, ksp, t.kc, smem, acc, TILE, 1 << 30, tid);
#if BOHIP_CHOL_TRACE
    if (tid == 0 && trace_slot >= 0 && trace_slot < EX_TRACE_CAP) g_ex_trace[8 * trace_slot + 5] = wall_clock64();
#endif
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = (wave & 3) >> 1, wc = wave & 1;
    const bool looker = do_look && tid < 64;
    const int lqi = tid >> 3, ldp = tid & 7;
    unsigned lk_h = 0, lk_n = 0, lk_f = 0;
    uint32_t lk_idx = EX_NONE, lk_want = 0;
    bool lk_live = false;
    if (looker && lqi < EX_NQ) {
        lk_n = (unsigned)(q.qbeg[lqi + 1] - q.qbeg[lqi]);
        lk_h = __hip_atomic_load(q.heads + lqi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the tile takes a turn through LDS and leaves as 16-byte agent-scope pieces, 1 KB contiguous per wave instruction
    constexpr int TS = CTILE + 2;
    double* Tl = smem;   // [128][66]: the staging buffers are free (the loop ended on a barrier)
    if (wave < 4) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) Tl[acc_row(lane, wr, mi) * TS + acc_col<4>(lane, wc, nj)] = acc[mi][nj];
    }
    __syncthreads();
    if (looker) {
        lk_live = lqi < EX_NQ && lk_h < lk_n;
        if (lk_live && ldp < EX_NDEP) {
            const ExTask* nt = q.tasks + q.qbeg[lqi] + lk_h;
            lk_idx = nt->dep_idx[ldp];
            lk_want = nt->dep_want[ldp];
        }
    }
    constexpr int NP = (TILE * CTILE / 2) / GEMM_THREADS_8;   // 8 pieces per thread
    const int rmw = t.rmw & 3, dh = t.diag_h;
    const bool has_ct = (t.rmw & 4) != 0;
    const double* P = has_ct ? nullptr : t.P;
    double* C = t.C;
    d2 oldv[NP], pv[NP];
    if (rmw == 1) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int piece = tid + GEMM_THREADS_8 * u, r = piece >> 5, c = (piece & 31) * 2;
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(oldv[u]) : "v"(C + (int64_t)r * ld + c) : "memory");
        }
        if (P) {
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int piece = tid + GEMM_THREADS_8 * u, r = piece >> 5, c = (piece & 31) * 2;
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pv[u]) : "v"(P + (int64_t)r * ld + c) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7])
                         :
                         : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(oldv[0]), "+v"(oldv[1]), "+v"(oldv[2]), "+v"(oldv[3]), "+v"(oldv[4]), "+v"(oldv[5]), "+v"(oldv[6]), "+v"(oldv[7])
                     :
                     : "memory");
    }
    if (looker && lk_idx != EX_NONE) lk_f = __hip_atomic_load(flags + lk_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int piece = tid + GEMM_THREADS_8 * u, r = piece >> 5, c = (piece & 31) * 2;
        double* dst = C + (int64_t)r * ld + c;
        const bool k0 = !(dh >= 0 && CTILE * dh + c > r), k1 = !(dh >= 0 && CTILE * dh + c + 1 > r);
        if (!k0) continue;   // (k1 implies k0): the strict upper triangle of a diagonal tile stays zero
        d2 v = *reinterpret_cast<const d2*>(Tl + r * TS + c);
        if (rmw == 1) {
            v.x = oldv[u].x - v.x;
            v.y = oldv[u].y - v.y;
            if (P) {
                v.x -= pv[u].x;
                v.y -= pv[u].y;
            }
        } else if (rmw == 2) {
            v.x = -v.x;
            v.y = -v.y;
        }
        oldv[u] = v;   // (kept for the transposed copy)
        if (k1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
        else __hip_atomic_store(dst, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (has_ct) {
        // the same values once more as a [64][128] block (rows = the tile's columns): through LDS again, so that they leave as
        // 16-byte pieces, 1 KB contiguous per wave instruction (never a diagonal tile)
        constexpr int TST = TILE + 2;
        __syncthreads();   // every thread has read its pieces of Tl
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int piece = tid + GEMM_THREADS_8 * u, r = piece >> 5, c = (piece & 31) * 2;
            Tl[c * TST + r] = oldv[u].x;
            Tl[(c + 1) * TST + r] = oldv[u].y;
        }
        __syncthreads();
        double* CTp = const_cast<double*>(t.P);
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int piece = tid + GEMM_THREADS_8 * u, cr = piece >> 6, rr = (piece & 63) * 2;
            const d2 v = *reinterpret_cast<const d2*>(Tl + cr * TST + rr);
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(CTp + (int64_t)cr * ld + rr), "v"(v) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // landed (a workgroup-scope fence emits no such wait)
    if (lane == 0) {
        if (t.sig_idx[0] != EX_NONE) atomicAdd(flags + t.sig_idx[0], 1u);
        if (t.sig_idx[1] != EX_NONE) atomicAdd(flags + t.sig_idx[1], 1u);
    }
    int res = -1;
    if (looker) {
        const bool pending = lk_live && lk_idx != EX_NONE && lk_f < lk_want;
        const unsigned long long lv = __ballot(lk_live), pd = __ballot(pending);
        int pickq = -1;
#pragma unroll
        for (int c = EX_NQ - 1; c >= 1; --c)   // (queue 0 has its own workgroups: see k_chol_exec)
            if (((lv >> (8 * c)) & 1ull) && ((pd >> (8 * c)) & 0xffull) == 0ull) pickq = c;
        if (pickq >= 0) res = (pickq << 24) | (int)((unsigned)__builtin_amdgcn_readlane((int)lk_h, 8 * pickq) & 0xffffffu);
    }
    return res;
}

// What a free workgroup does (wave 0 decides, the others wait at the barrier):
//   nothing in hand   claim -- from the queue the look during the last epilogue found runnable, else after a fresh look.  The
//                     claim is a fetch-and-add, so under contention it lands a little behind the head that was seen runnable,
//                     on a task whose counters may not be in yet.
//   counters in       run it.
//   counters not in   HOLD it: one slot per kind -- `pend` (queues 1, 2), `pend2` (bulk), `pend3` (inverse).  An urgent task
//                     (queue 0) is simply waited for.  A held record is usually microseconds from runnable (its producers were
//                     claimed just before it), so for `patience` the workgroup only polls it; after that it takes what its free
//                     slots allow -- a held inverse record restricts nothing but other inverse claims (nothing outside the
//                     inverse ever waits for one); with `pend` occupied only fill-in work: inverse waves, bulk if asked for.
//                     (Taking a 60-100 us task at once made held records start that much late: on a mostly idle chip the
//                     inverse's rows fell 150-340 us behind the pivots, N = 6000 cost 5.4 instead of 4.9 ms.)  A workgroup NEVER
//                     blocks on one claim while it holds another: the first version of this waited for the bulk claim and
//                     dead-locked -- a bulk task of the next group waits for blocks the chain can only reach once the Early sum
//                     held in the other slot has run.  Held tasks are never given back; what a held task waits for is running,
//                     held by a workgroup that keeps polling it, or the chain's.
__global__ __launch_bounds__(GEMM_THREADS_8, 4) void k_chol_exec(ExQueues q) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int s_task, s_cnt, s_cut;
    // the workgroup's scheduling state lives in LDS (wave 0 reads and writes it between tasks): in registers it would be alive
    // across the contraction loop, which has none to spare
    __shared__ int s_look;     // what the look during the previous task's epilogue found ((queue << 24) | head), -1 nothing
    __shared__ int s_pend;     // a claimed task of the queues 1, 2 whose counters were not in when last looked at
    __shared__ int s_pend2;    // the same for a bulk claim (queue EX_QBULK)
    __shared__ int s_pend2n;   //   records of that bulk claim still to run (0: the whole claim)
    __shared__ int s_pend3;    // a claimed record of the inverse queue whose counters are not in: held in a slot of its own that
                               // restricts nothing -- no other queue and not the chain ever waits for an inverse record, so a
                               // workgroup holding one goes on claiming everywhere else
    __shared__ unsigned s_tp, s_tp3;   // when the records in `pend` / `pend3` were claimed (low word of the wall clock)
    __shared__ int s_urgent;
    // queues this workgroup may ever claim from.  The workgroups from `second_from` on are (as dispatched) the SECOND ones of their CUs: they
    // take throughput work only (early sums, bulk, waves) and leave when that is exhausted, so that what remains at the end -- the inverse's
    // row-to-row chain -- runs one workgroup per CU
    const unsigned lane_mask = ((int)blockIdx.x >= q.nurgent && (int)blockIdx.x < q.nurgent + q.nfast)
                                   ? ~((1u << EX_QBULK) | (1u << EX_QWAVE))
                                   : (q.second_from > 0 && (int)blockIdx.x >= q.second_from ? ((1u << 2) | (1u << EX_QBULK) | (1u << EX_QWAVE)) : ~0u);
    if (threadIdx.x == 0) { s_look = -1; s_pend = -1; s_pend2 = -1; s_pend2n = 0; s_pend3 = -1; s_tp = 0u; s_tp3 = 0u; s_urgent = (int)blockIdx.x < q.nurgent; }
    __syncthreads();
    // The urgent queue (~10 tasks per block: what the chain kernel reads next) has its own workgroups: each takes the next urgent
    // task with a fetch-and-add AHEAD of time and waits for its counters, so the task starts the moment they arrive -- no look, no
    // claim on the critical path -- and nobody else touches that queue.  (Claimed by everybody it was either raced -- when its
    // head turned runnable every free workgroup saw it at once and walked away with an urgent task of a block further down, which
    // nobody may work beside: ~250 of 490 workgroups parked -- or, claimed exactly by compare-and-swap, serialised at one claim
    // per look, ~5 us each.)
    for (;;) {
#if BOHIP_CHOL_TRACE
        const unsigned long long tr0 = wall_clock64();
#endif
        if (threadIdx.x < 64) {
            int pl = threadIdx.x;
            asm volatile("" : "+v"(pl));   // opaque: the look's lane arithmetic must not be kept alive across the task
            int run = -1, run_n = 0;   // run_n > 0: only that many records (the rest of a bulk claim that was interrupted)
            int look = s_look, pend = s_pend, pend2 = s_pend2, pend2_n = s_pend2n, pend3 = s_pend3;
            unsigned tp = s_tp, tp3 = s_tp3;
            bool urgent_wg = s_urgent != 0;
            const unsigned long long t_wait = wall_clock64();
            auto queue_of = [&](int t) {
                int qq = 0;
#pragma unroll
                for (int c = 1; c < EX_NQ; ++c) qq = t >= q.qbeg[c] ? c : qq;
                return qq;
            };
            auto in_hand_ready = [&](int t) { const int qt = queue_of(t); return ex_ready_once(q, q.tasks + t, ex_ntile(q, qt, (unsigned)(t - q.qbeg[qt])), pl); };
            if (urgent_wg) {
                unsigned c = 0;
                if (pl == 0) c = atomicAdd(q.heads, 1u);
                c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
                if (c < (unsigned)(q.qbeg[1] - q.qbeg[0])) {
                    run = ex_wait_task(q, q.tasks + q.qbeg[0] + c, 1, pl) ? q.qbeg[0] + (int)c : -1;
                } else {
                    urgent_wg = false;   // the urgent queue is exhausted: an ordinary workgroup from here on
                }
            }
            if (!urgent_wg && run < 0)
            for (;;) {
                if (pend >= 0 && in_hand_ready(pend)) { run = pend; pend = -1; break; }
                if (pend2 >= 0 && (pend2_n > 0 ? ex_ready_once(q, q.tasks + pend2, 1, pl) : in_hand_ready(pend2))) {
                    run = pend2; run_n = pend2_n; pend2 = -1; pend2_n = 0; break;
                }
                if (pend3 >= 0 && in_hand_ready(pend3)) { run = pend3; pend3 = -1; break; }
                bool rdy = true;
                int tk = -3;
                // which queues may this workgroup claim from right now?  one held record per slot: `pend` (queues 1, 2), `pend2`
                // (bulk), `pend3` (inverse: rows or waves).  While `pend` is occupied only fill-in work is taken: inverse waves
                // (nothing ever waits for those), bulk if asked for (q.fill).
                // A record that was claimed a little ahead of its counters is usually microseconds from runnable (its producers
                // were claimed just before it): for `patience` the workgroup only polls -- picking up a 60-100 us task meanwhile
                // made row-chain records and row steps start that much late on a mostly idle chip (N = 3000: the inverse's rows
                // fell 150-340 us behind the pivots) -- and takes other work only when the wait turns out to be a long one.
                const unsigned now = (unsigned)wall_clock64();
                const bool patient = (pend >= 0 && now - tp < q.patience_ticks) ||
                                     (pend3 >= 0 && pend3 < q.qbeg[EX_QROWS + 1] && now - tp3 < q.patience_ticks);
                unsigned qmask = 0u;
                if (patient) {
                } else if (pend < 0) {
                    qmask = 6u;
                    if (pend3 < 0) qmask |= 1u << EX_QROWS;
                    if (pend2 < 0) qmask |= 1u << EX_QBULK;
                    if (pend3 < 0 && pend2 < 0) qmask |= 1u << EX_QWAVE;
                } else {
                    if (pend2 < 0 && (q.fill >= 2 || (q.fill == 1 && queue_of(pend) == 2))) qmask |= 1u << EX_QBULK;
                    if (pend3 < 0 && q.fill_inv) qmask |= 1u << EX_QWAVE;
                }
                qmask &= lane_mask;
                const bool nothing_held = pend < 0 && pend2 < 0 && pend3 < 0;
                if (qmask != 0u) {
                    if (look >= 0 && ((qmask >> (look >> 24)) & 1u)) { tk = ex_claim(q, look >> 24, (unsigned)(look & 0xffffff), pl, rdy); if (tk == -2) tk = -3; }
                    look = -1;
                    if (tk == -3) tk = ex_pick(q, pl, rdy, qmask, nothing_held);
                    if (tk == -1) { run = -1; break; }     // every queue exhausted (nothing held either), or abort
                }
                if (tk >= 0) {
                    if (rdy) { run = tk; break; }
                    const int qt = queue_of(tk);
                    if (qt == EX_QROWS || qt == EX_QWAVE) { pend3 = tk; tp3 = (unsigned)wall_clock64(); }
                    else if (qt == EX_QBULK) pend2 = tk;
                    else { pend = tk; tp = (unsigned)wall_clock64(); }
                    continue;
                }
                if (__hip_atomic_load(q.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { run = -1; break; }
                if (wall_clock64() - t_wait > q.spin_ticks) {
                    if (pl == 0) atomicCAS(q.abort, 0u, 1u);
                    run = -1;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
            if (pl == 0) { s_task = run; s_cnt = run_n; s_look = -1; s_pend = pend; s_pend2 = pend2; s_pend2n = pend2_n; s_pend3 = pend3; s_tp = tp; s_tp3 = tp3; s_urgent = urgent_wg ? 1 : 0; }
        }
        __syncthreads();
        const int ti = __builtin_amdgcn_readfirstlane(s_task);
        const int tn = __builtin_amdgcn_readfirstlane(s_cnt);
        __syncthreads();
        if (ti < 0) break;
        int qi_ = 0;
#pragma unroll
        for (int c = 1; c < EX_NQ; ++c) qi_ = ti >= q.qbeg[c] ? c : qi_;
        const int pair = tn > 0 ? tn : min(q.stride[qi_], q.qbeg[qi_ + 1] - ti);
        for (int u = 0; u < pair; ++u) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));   // opaque per task: nothing lane-dependent is hoisted out of this loop
#if BOHIP_CHOL_TRACE
            const unsigned long long tr1 = wall_clock64();
#endif
            const int look = ex_run(q.tasks[ti + u], q, smem, tid, u == pair - 1, ti + u);
            if (threadIdx.x == 0) s_look = look;
            // between the records of a bulk claim: has the task held in the first slot become runnable?  then it goes first and
            // the rest of the claim waits in the second slot (a held task is never delayed by more than one record, ~65 us)
            int cut = 0;
            if (u + 1 < pair && qi_ == EX_QBULK && threadIdx.x < 64) {
                int pl = threadIdx.x;
                asm volatile("" : "+v"(pl));
                const int pend = s_pend;
                if (pend >= 0 && s_pend2 < 0) {
                    const int qp = pend >= q.qbeg[2] ? 2 : 1;
                    if (ex_ready_once(q, q.tasks + pend, ex_ntile(q, qp, (unsigned)(pend - q.qbeg[qp])), pl)) cut = 1;
                }
                if (pl == 0) {
                    s_cut = cut;
                    if (cut) { s_pend2 = ti + u + 1; s_pend2n = pair - (u + 1); s_look = -1; }
                }
            }
            __syncthreads();   // the LDS tile is rewritten by the next task's DMA
            if (u + 1 < pair && qi_ == EX_QBULK) {
                const int c_ = __builtin_amdgcn_readfirstlane(s_cut);
                __syncthreads();
                if (c_) break;
            }
#if BOHIP_CHOL_TRACE
            if (threadIdx.x == 0 && ti + u < EX_TRACE_CAP) {
                g_ex_trace[8 * (ti + u)] = u == 0 ? tr0 : tr1; g_ex_trace[8 * (ti + u) + 1] = tr1; g_ex_trace[8 * (ti + u) + 2] = wall_clock64();
                g_ex_trace[8 * (ti + u) + 3] = blockIdx.x;
            }
#endif
        }
    }
}

}  // namespace bohip
