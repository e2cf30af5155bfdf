// bohip.hip -- libbohip.so: C ABI (include/bohip.h) + host orchestration of the gfx950 kernels.
// Single translation unit (the kernel files are included) so one hipcc invocation builds the library.
//
// HBM layout per handle (capacity C observations, ld = round_up(C + 1, 128)):
//   dX  [C][d]      observations, row = one observation (= Julia's d x N column-major)
//   dy  [C]
//   dL  [ld][ld]    row-major; lower 128-tiles hold cK then its Cholesky factor L (= Julia's upper U
//                   read column-major); strict-upper tiles stay zero; rows/cols >= N identity padding
//   dW  [ld][ld]    W = L^-1 (lower).  Row N (first padding row) carries alpha' so the scoring
//                   contraction V = W K* also produces mu - beta.
//   dWT [ld][ld]    W' (upper): lets every contraction run in the K-major x K-major form of the MFMA engine
//   dS  [ld][ld]    scratch: solved panels during the factorisation, S' blocks during the recursive inverse,
//                   cK^-1 for the marginal-likelihood gradient
//   dKsT [Rc][ld]   cross-covariance chunk, candidate-major
// There is NO CPU fallback: every entry point that computes fails with BOHIP_E_NODEVICE / BOHIP_E_HIP
// when the GPU is unavailable.
#include "../../include/bohip.h"
#include "kernels_linalg.hip"
#include "kernels_chol.hip"
#include "kernels_exec.hip"
#include "kernels_score.hip"
#include "kernels_ascent.hip"
#include "kernels_small.hip"   // (after the ascent: k_small_u's last workgroup runs its step, asc_step_one<true>)
#include "direct_l.h"          // host bookkeeping of :GN_DIRECT_L (ask / tell)

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>
#include <string>
#include <vector>
#include <queue>

using namespace bohip;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(BOHIP_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ \
                                         ":" + std::to_string(__LINE__) + ")");                        \
    } while (0)
#define CHK(expr)              \
    do {                       \
        int rc_ = (expr);      \
        if (rc_ != 0) return rc_; \
    } while (0)

static_assert(sizeof(bohip_best) == sizeof(Best), "record layout");
static constexpr int SMALL_MAX = 256;                      // most candidates the row-wise path takes in one go
static constexpr int APP_UT_ROW0 = SMALL_MAX;              // rows [0, 256): V' / L21;  rows [256, 512): U' / T = L21 W11
static constexpr int APP_ROWS = 2 * SMALL_MAX;
static_assert(APPEND_PMAX <= SMALL_MAX && SMALL_R <= SMALL_MAX, "the append shares the small-batch row area");

static constexpr int CHOL_DF_TCAP = 96;   // the dataflow factorisation's flag storage is sized for this many row tiles
struct bohip_gp {
    int device = 0, d = 0, kern = 0;
    int64_t n = 0, cap = 0, ld = 0;
    double *dX = nullptr, *dy = nullptr, *dL = nullptr, *dW = nullptr, *dWT = nullptr, *dS = nullptr;
    double *dalpha = nullptr, *dr = nullptr, *dt = nullptr, *dmll = nullptr;
    double* dApp = nullptr;  // [APP_ROWS][ld] scratch of the incremental append and of the row-wise (small-batch) posterior
    int* dinfo = nullptr;
    unsigned* dchol_flags = nullptr;   // dataflow factorisation (kernels_chol.hip): panel[T*8] | solved[T] | crit[T] | abort[1]
    double* dchol_idl = nullptr;       // [T*8][16][16] published inverses of the 16 x 16 pivot blocks
    std::vector<double> hX, hy;
    double loglen[DMAX], logsig = 0.0, lognoise = -2.0, beta = 0.0;
    bool stale = true;
    int64_t n_factored = 0;  // observations covered by the current factor
    hipStream_t stream = nullptr, own_stream = nullptr;
    hipStream_t side_stream = nullptr;            // bulk trailing updates of the factorisation run here (look-ahead)
    hipEvent_t ev_panels = nullptr, ev_bulk = nullptr;
    hipStream_t col_stream = nullptr;             // dataflow factorisation: bulk followers + column updates (high priority)
    hipStream_t inv_stream = nullptr;             // W = L^-1 grows block by block beside the factorisation's diagonal chain
    hipEvent_t ev_blk = nullptr, ev_inv = nullptr;
    hipEvent_t ev_gate = nullptr;                 // a diagonal-block kernel is about to start: release one piece of the pending bulk update
    std::vector<hipEvent_t> ev_tier;              // its cross-stream hand-overs (two per group of four blocks)
    bool w_seeded = false;                        // the diagonal blocks of W were produced during the factorisation (k_chol_inverter)
    ExTask* dex_tasks = nullptr;                  // task records of the executor form (kernels_exec.hip), built once per (buffers, T)
    size_t ex_cap = 0;
    bool ex_bulk4_ok = false;   // the bulk queue may be claimed two tiles at a time (exec_bulk_stride_ok)
    int ex_T = 0, ex_nsf = 0, ex_inv_g = -1, ex_grp_min = -1, ex_qbeg[EX_NQ + 1] = {0};
    bool w_done = false;       // the last factorisation also produced W = L^-1 (executor form with its inverse queue)
    // scoring scratch
    double* dKsT = nullptr;
    unsigned long long* dclk = nullptr;   // [2] core-clock / wall-clock ticks of sampled k_trigemm_sq workgroups (timing runs)
    int* dpieces = nullptr;      // k_trigemm_sq's row pieces, heaviest first (trigemm_pieces)
    std::vector<int> hpieces;
    int n_pieces = 0, pieces_T = -1;
    int64_t pieces_cap = 0;
    int64_t pieces_alpha = -1;
    int64_t kst_rows = 0;        // rows the K*' chunk buffer holds
    int64_t chunk_now = 0;       // candidates per K*' chunk of the current call (chunk_rows(R) <= kst_rows)
    int64_t score_launches = 0;  // k_trigemm_sq launches of the last posterior pass (BOHIP_INFO_SCORE_LAUNCHES)
    double *dVT = nullptr, *dUT = nullptr;  // gradient path: V' and U' = V' W chunks, candidate-major
    int64_t vt_rows = 0;
    double *dq = nullptr, *dmu_raw = nullptr, *dXs = nullptr, *dmu = nullptr, *dvar = nullptr, *dscore = nullptr;
    int64_t q_cap = 0, r_cap = 0, xs_cap = 0;
    Best *dblock_best = nullptr, *dbest = nullptr;
    int64_t bb_cap = 0;
    unsigned* dfz_cnt = nullptr;   // fused finish of k_trigemm_sq: [tiles] arrivals per candidate tile + [1] finished tiles (kept at zero)
    Best* dfz_best = nullptr;      // [tiles] per-tile arg-max records
    int64_t fz_cap = 0;
    bool fz_dirty = false;         // a fused call failed between its launches: clear the counters before the next one
    double* dgrad = nullptr;   // d x R gradient staging of the host-pointer entry point
    double asc_ftol_abs = 0.0, asc_xtol_rel = 0.0, asc_stopval = INFINITY;   // bohip_gp_set_ascent_stop (NLopt's ftol_abs / xtol_rel / stopval)
    double asc_maxtime = 0.0;    // bohip_gp_set_maxtime: wall-clock budget of one acquire_max call in seconds (NLopt maxtime), 0 = none
    int64_t batch_hint = 0;      // bohip_gp_set_batch_hint: choose the scoring path as if the batch had at least this many candidates
    double* dsplit = nullptr;    // split-K partial planes (batches of a few hundred candidates)
    int64_t split_cap = 0;
    const SmallFold* asc_fold = nullptr;   // set around a pass of the free-running ascent: the step this pass's gradient kernel may run in its last workgroup
    bool asc_fold_done = false;            // ... and whether it took it (small_pass_mfma: one pass of <= 16 candidates, d <= 16); else k_asc_step follows
    const unsigned* asc_go = nullptr;   // set around the passes of the free-running ascent: the pass's big kernels return at once when the word is 0
    double* dgparts = nullptr;   // [SMALL_MAX][16][2 DMAX] split partial sums of k_grad_finish (small batches)
    unsigned* dgcount = nullptr; // per-candidate arrival counters (left at zero by the kernel)
    // round 5: the small-batch pass as two MFMA kernels (kernels_small.hip): one scratch block [part | v16 | qpart | mupart | post | fstash | gpart]
    double* dsm = nullptr;
    size_t sm_bytes = 0;
    unsigned* dsm_cnt = nullptr;
    int sm_cnt_T = 0;
    // pinned, device-visible host block the kernels of a small batch (R <= SMALL_R) write their results into: the
    // 2-3 device-to-host copies of a call cost more than its kernels (each ~8 us of API + DMA set-up)
    double* hpin = nullptr;      // [score 32 | mu 32 | var 32 | best 2 | grad 32 DMAX | candidates of a small host call 32 DMAX]
    double* dschur = nullptr;    // [APPEND_PMAX][APPEND_PMAX] Schur complement of an append of >= 3 rows (k_schur_dots)
    // lock-step L-BFGS ascent of acquire_max (kernels_ascent.hip)
    AscentState asc{};
    double* asc_block = nullptr;   // one allocation behind all double arrays of asc
    int* asc_ints = nullptr;       // active, accepted
    double* asc_hio = nullptr;     // pinned staging of the one-launch ascent: [lb | ub | starts] in, the packed result out
    double* asc_dio = nullptr;     // its device mirror
    size_t asc_io_cap = 0;
    int* asc_hints = nullptr;      // pinned: h_accepted, h_active
    double* asc_bounds = nullptr;  // lb, ub, best_x (3 d doubles) + starts staging is dXs
    Best* asc_best = nullptr;
    int64_t asc_cap = 0;
    int64_t grad_cap = 0;
    Best* dthompson = nullptr; // S arg-max records
    double* ddmll_parts = nullptr;  // per-block partial sums of the marginal-likelihood gradient
    double *dVV = nullptr, *dcov = nullptr;  // [Rp][Rp] V'V (lower tiles) and the full posterior covariance
    int64_t cov_cap = 0;
    int64_t dmll_cap = 0;
    int64_t thompson_cap = 0;
    // one-process-per-device exchange (multigpu.hip): communicator attached by bohip_gp_comm_init
    void* comm = nullptr;
    int comm_rank = 0, comm_n = 0;
    int64_t comm_exchanges = 0;        // BOHIP_INFO_COMM_EXCHANGES: all-gathers this handle has issued on its communicator
    Best *csend = nullptr, *crecv = nullptr, *cfinal = nullptr;
    int64_t crec_cap = 0;
    // bookkeeping
    int64_t pivot = 0, refits = 0, appends = 0;
    int alpha_inc_run = 0;         // appends since alpha was last computed in full (compute_alpha); the incremental form re-synchronises every 256
    bool mirror_dirty = false;     // a staged append failed after the host mirrors advanced: the next refit uploads X and y again
    int64_t chol_lock_skips = 0;   // refits that took the launch-chained form because another process held the refit lock
    int chol_form_last = 0;            // BOHIP_INFO_CHOL_FORM: 0 launch chain, 1 dataflow form 1, 2 form 2, 3 form 2 left-looking, 4 executor
    int64_t chol_fallbacks = 0;        // BOHIP_INFO_CHOL_FALLBACKS: refits of this handle that timed out on a dependency and were redone launch-chained
    int chol_abort_T = 0;              // BOHIP_INFO_CHOL_ABORT_TILES: row tiles of the last factorisation that timed out (0: never)
    // jitter escalation on a failed factorisation (GaussianProcesses.jl make_posdef!, UPSTREAM-UNVERIFIED: off by default)
    double jitter_rel = 0.0;           // first jitter = jitter_rel x mean(diag cK), x10 per further try
    int jitter_tries = 0;              // 0 = report BOHIP_E_NOTPD at once (the default)
    int jitter_steps_last = 0;         // BOHIP_INFO_JITTER_STEPS: tries the last refit needed (0: none)
    double jitter_last = 0.0;          // what it added to the diagonal
    int q_tiles = 0;  // row tiles behind dq after the last posterior pass (two partial sums each); 0: dq[r] is the finished sum (row-wise / split-K paths)
    bool timing = false;
    bool t_open = false;
    bool timing_dominant_only = false;   // enable_timing(2): only the dominant kernel (k_trigemm_sq) is bracketed by events
    bool timing_accumulate = false;      // enable_timing(3): as 2, and the event pairs of successive calls pile up unread until
                                         // bohip_gp_get_timing (no hipEventSynchronize / hipEventElapsedTime inside a timed loop)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> tpool;  // pooled event pairs
    size_t tused = 0;
    std::vector<const char*> tlabel;
    std::vector<std::string> tnames;
    std::vector<double> tms;
};

static KernelHyper make_hyper(const bohip_gp* g) {
    KernelHyper h;
    h.kern = g->kern;
    h.d = g->d;
    h.sigma2 = std::exp(2.0 * g->logsig);
    for (int k = 0; k < DMAX; ++k) h.il2[k] = 0.0;
    for (int k = 0; k < g->d; ++k) h.il2[k] = std::exp(-2.0 * g->loglen[g->kern == KERN_SEISO ? 0 : k]);
    return h;
}

// ---- stage timing (HIP events on the handle's stream; the event pairs are pooled, not re-created per call) -------
static bool t_skip(const bohip_gp* g, const char* name) {
    return g->timing_dominant_only && std::strncmp(name, "trigemm_sq", 10) != 0;
}
static void t_begin(bohip_gp* g, const char* name) {
    if (!g->timing) return;
    g->t_open = !t_skip(g, name);
    if (!g->t_open) return;
    if (g->timing_accumulate && g->tused >= 8192) {   // nobody reads them: keep the newest (mode 3 piles pairs up until get_timing)
        std::rotate(g->tpool.begin(), g->tpool.begin() + 4096, g->tpool.begin() + g->tused);
        g->tlabel.erase(g->tlabel.begin(), g->tlabel.begin() + 4096);
        g->tused -= 4096;
    }
    if (g->tused == g->tpool.size()) {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        g->tpool.push_back({a, b});
    }
    hipEventRecord(g->tpool[g->tused].first, g->stream);
    g->tlabel.push_back(name);
    ++g->tused;
}
static void t_end(bohip_gp* g) {
    if (!g->timing || g->tused == 0 || !g->t_open) return;
    g->t_open = false;
    hipEventRecord(g->tpool[g->tused - 1].second, g->stream);
}
static void t_reset(bohip_gp* g) {
    if (g->timing_accumulate) return;
    g->tused = 0;
    g->tlabel.clear();
}
static void t_collect(bohip_gp* g, bool force = false) {
    if (!g->timing) return;
    if (g->timing_accumulate && !force) return;
    g->tnames.clear();
    g->tms.clear();
    for (size_t i = 0; i < g->tused; ++i) {
        hipEventSynchronize(g->tpool[i].second);
        float ms = 0.f;
        hipEventElapsedTime(&ms, g->tpool[i].first, g->tpool[i].second);
        g->tnames.push_back(g->tlabel[i]);
        g->tms.push_back(ms);
    }
    g->tused = 0;
    g->tlabel.clear();
}

// ---- allocation -----------------------------------------------------------------------------------
static int free_model(bohip_gp* g) {
    for (double** p : {&g->dX, &g->dy, &g->dL, &g->dW, &g->dWT, &g->dS, &g->dalpha, &g->dr, &g->dt, &g->dApp, &g->dchol_idl})
        if (*p) { hipFree(*p); *p = nullptr; }
    if (g->dchol_flags) { hipFree(g->dchol_flags); g->dchol_flags = nullptr; }
    g->ex_T = 0;   // the task records hold pointers into the buffers just freed
    return 0;
}
static size_t chol_flag_words(int T);
static size_t chol_abort_word(int T);
static int alloc_model(bohip_gp* g, int64_t cap) {
    free_model(g);
    // chunk buffers are sized by ld: drop them, they are re-made on demand
    for (double** p : {&g->dKsT, &g->dVT, &g->dUT, &g->dq})
        if (*p) { hipFree(*p); *p = nullptr; }
    g->kst_rows = g->vt_rows = g->q_cap = 0;
    g->cap = cap;
    // +16 doubles: row stride is an ODD multiple of 128 B, so the 128 rows of a tile spread over the L2/HBM
    // channels instead of camping on one (a power-of-two stride sends every row of a tile to the same channel).
    g->ld = round_up(cap + 1, TILE) + 16;
    const size_t mat = (size_t)g->ld * g->ld * sizeof(double);
    HIPCHK(hipMalloc(&g->dX, std::max<size_t>(8, (size_t)cap * g->d * 8)));
    HIPCHK(hipMalloc(&g->dy, std::max<size_t>(8, (size_t)cap * 8)));
    HIPCHK(hipMalloc(&g->dL, mat));
    HIPCHK(hipMalloc(&g->dW, mat));
    HIPCHK(hipMalloc(&g->dWT, mat));
    HIPCHK(hipMalloc(&g->dS, mat));
    // (+256: k_trimv_stream reads its right-hand sides in whole chunks of 256, masked by index)
    HIPCHK(hipMalloc(&g->dalpha, (g->ld + 256) * 8));
    HIPCHK(hipMalloc(&g->dr, (g->ld + 256) * 8));
    HIPCHK(hipMalloc(&g->dt, (g->ld + 256) * 8));
    HIPCHK(hipMalloc(&g->dApp, (size_t)APP_ROWS * g->ld * 8));
    {
        const size_t Tm = (size_t)(g->ld / TILE) + 1;
        HIPCHK(hipMalloc(&g->dchol_flags, chol_flag_words((int)std::min<size_t>(Tm, CHOL_DF_TCAP)) * sizeof(unsigned)));
        HIPCHK(hipMalloc(&g->dchol_idl, Tm * CH_PANELS * 256 * 8));   // W16 of every pivot block
    }
    HIPCHK(hipMemsetAsync(g->dL, 0, mat, g->stream));
    HIPCHK(hipMemsetAsync(g->dW, 0, mat, g->stream));
    HIPCHK(hipMemsetAsync(g->dWT, 0, mat, g->stream));
    HIPCHK(hipMemsetAsync(g->dS, 0, mat, g->stream));
    if (g->n > 0) {
        HIPCHK(hipMemcpyAsync(g->dX, g->hX.data(), (size_t)g->n * g->d * 8, hipMemcpyHostToDevice, g->stream));
        HIPCHK(hipMemcpyAsync(g->dy, g->hy.data(), (size_t)g->n * 8, hipMemcpyHostToDevice, g->stream));
    }
    g->stale = true;
    return 0;
}

// ---- GEMM launcher --------------------------------------------------------------------------------
static int g_inv_overlap = 0;  // BOHIP_INV_OVERLAP=1: grow W = L^-1 block by block beside the factorisation instead of after it.
                               // Measured: refit 3.67 -> 3.56 ms (N=3000), 21.4 -> 19.9 ms (N=10000), but the factorisation itself slows
                               // down under the competition (2.90 -> 3.08 ms, 13.9 -> 17.7 ms), so it stays opt-in.
static int g_bulk_pieces = 4;   // gated pieces of the side-stream bulk update per outer block (BOHIP_BULK_PIECES; 0/1: one launch)
static int g_split = 1;   // split-K path for batches of a few hundred candidates (BOHIP_SPLIT=0 disables)
static int g_asc_wg_nmax = 256;  // BOHIP_ASC_WG_NMAX: models up to this many observations run acquire_max as ONE launch, one workgroup per start point
                                 // (kernels_ascent.hip k_ascent_wg); 0: never
static const double g_asc_first_step = 0.1;   // the first step is never shorter than this fraction of the smallest box side (kernels_ascent.hip asc_direction_one)
static int g_asc_fold = 1;       // the free-running form's step inside k_small_u's last workgroup (BOHIP_ASC_LOCKSTEP=2: as a launch of its own, until round 6)
static int g_asc_lockstep = 0;   // BOHIP_ASC_LOCKSTEP=1: the lock-step driver of the device ascent (five launches + a stream synchronisation per
                                 // evaluation pass) instead of the free-running one (k_asc_step)
static int g_small_mfma = 1;   // BOHIP_SMALL_MFMA: the small-batch pass as two MFMA kernels (kernels_small.hip); 0 = round 4's five kernels
static int g_small_m = 0;      // BOHIP_SMALL_M: 128-chunks of the contraction index per tile of those kernels; 0 = by size
static int g_small_r = -1;  // batches up to this size (<= 256) take the row-wise path; -1: min(256, 90 + 300000 / N), the measured
                            // break-even with the MFMA path (N=500: > 256, N=3000: ~190, N=10000: ~110); BOHIP_SMALL_R overrides: below that
                            // candidates one 64-wide MFMA tile column leaves single workgroups walking the whole K extent
static int g_inv_hi_h = 16;            // levels of the triangular inverse with half-size >= this many tiles run as throughput launches (BOHIP_INV_HI_H)
static int g_chol_df2_ll = 1;          // left-looking window updates (cholesky_dataflow3) instead of cholesky_dataflow2's K = 128 ones: N=8000 7.46 vs
                                       // 7.63 ms, N=10000 11.7 vs 11.9, N=12000 16.5 vs 18.9 (BOHIP_CHOL_DF2_LL=0: the latter)
static int g_chol_df2_coal = 1;        // its K = 512 launches store through LDS in 16-byte pieces (BOHIP_CHOL_DF2_COAL=0: from the MFMA layout)
static int g_chol_df2_hi = 1;          // its K = 512 launches with two workgroups per CU (BOHIP_CHOL_DF2_HI=0: one)
static int g_chol_df2_win = 6;         // its window: block k's flagged update reaches column 4 (k / 4) + win (BOHIP_CHOL_DF2_WIN, 6..10; 6 is the least that keeps the chain's next tiles inside)
static int g_chol_df2_min = 47;  // cholesky_dataflow2 (large-T form) from this many row tiles on (BOHIP_CHOL_DF2_MIN): N=6000 5.13 vs 5.3 ms, N=8000 7.8
                                 // vs 8.65, N=10000 12.1 vs 13.06, N=11000 14.6 vs 15.4; below (N=5600) the first form is faster (4.58 vs 4.78)
static int g_chol_df = 1;  // dataflow factorisation (kernels_chol.hip) for 3 <= T <= g_chol_df_tmax row tiles: N=3000 1.66 vs 2.71 ms, N=1000 0.57
                            // vs 0.82 ms.  Far beyond that its one-tier K=128 bulk updates lose to the two-tier launch chain, below 3 there is
                            // nothing to pipeline.  BOHIP_CHOL_DATAFLOW=0 disables, =2 forces it for every 2 <= T <= CHOL_DF_TCAP.
static bool g_chol_df_strict = false;   // BOHIP_CHOL_DF_STRICT: a timed-out dependency is an error instead of a fall-back (tests, tools)
static bool g_chol_df_dump = false;     // measurement builds: the flags of a timed-out factorisation on stderr
static int g_chol_df_tmax = 96;   // = CHOL_DF_TCAP, N <= ~12200 (N=12000: 16.5 vs 18.5 ms for the launch chain)
// A dataflow factorisation that timed out on a dependency (another process holds the CUs, two streams share a hardware queue, ...)
// switches the PROCESS to the launch chain -- for the next `skip` refits, not for good: the cause is usually transient, a
// time-out costs one bounded wait (200 ms), and every further time-out doubles the pause (8, 24, 56, ... up to 1024 refits).
static std::atomic<int> g_chol_df_skip{0}, g_chol_df_backoff{0};
static unsigned long long g_chol_spin_ticks = CH_SPIN_TICKS_DEFAULT;   // BOHIP_CHOL_SPIN_US: bound of every in-kernel wait of the dataflow forms
static int g_chol_exec = 1;       // executor form (kernels_exec.hip, cholesky_exec) from g_chol_exec_min row tiles on (BOHIP_CHOL_EXEC=0: the stream-based second form)
static int g_chol_exec_min = -1;  // BOHIP_CHOL_EXEC_MIN, row tiles.  Default: 4 with the executor's inverse queues (factorisation + inverse: N=500 0.41 ms against
                                  // 0.51 for the first dataflow form + the inverse behind it, N=1000 0.70 / 0.87, N=3000 2.02 / 2.49, N=4000 2.96 / 3.50), 32 without
                                  // them (the factorisation alone: N=3000 1.75 against 1.67 for the first form, N=4000 2.42 / 2.49, N=5000 3.23 / 3.75)
static int g_chol_exec_patience_us = 1000;   // BOHIP_CHOL_EXEC_PATIENCE_US: how long a workgroup only polls a held record before it takes other work meanwhile
static int g_chol_exec_fill_inv = 1;    // a workgroup waiting for the counters of a claimed task runs inverse-wave tasks meanwhile (BOHIP_CHOL_EXEC_FILL_INV)
static int g_chol_exec_inv_pairs = 0;   // inverse queue claimed one record (0) or one tile = two records (1) at a time (BOHIP_CHOL_EXEC_INV_PAIRS)
static std::atomic<int> g_chol_inv_grp_min{28};   // row tiles from which the inverse queues take the GROUP form (exec_task_list); BOHIP_CHOL_INV_GRP_MIN.  40 until the end of
                                                  // round 6: with that round's faster chain the group form wins from 28 row tiles on (28 ... 39 row tiles: -3 ... -7 %, 26: +3 %, 24: +4 %)
static std::atomic<int> g_chol_inv_g{8};      // executor form: W = L^-1 is grown behind the chain by a fifth task queue, in pieces of this many 128-blocks
                                   // of contraction (BOHIP_CHOL_INV_G; 0 = off: the level-by-level inverse runs after the factorisation)
static int g_chol_nsf = 3;         // solve-follower workgroups of the chain kernel in the executor form (BOHIP_CHOL_NSF, 1..6)
static int g_chol_copy_early = 1;   // BOHIP_CHOL_COPY_EARLY=0: the copy S -> L behind the executor instead of behind the chain kernel (cholesky_exec)
static int g_chol_exec_urgent = -1; // executor workgroups that serve the urgent queue only (BOHIP_CHOL_EXEC_URGENT); -1: 32, and 16 from 56 row tiles on
                                    // (N = 10^4: 14.0-14.2 against 14.3 ms; 8 starve the chain at N = 6000: 4.75 against 3.95; profiles/r04_inverse_group_form.txt)
static int g_chol_exec_fill = 0;   // BOHIP_CHOL_EXEC_FILL=1/2: a workgroup that holds a claimed task whose counters are not in takes bulk work meanwhile (1: Early sums only, 2: also row solves / updates).  Measured without effect on the total (N=10^4: 9.6-9.9 ms in every mode): more workgroups are busy, but the factorisation is paced by the per-block row steps, not by throughput -- so the default stays the simple rule
static int g_chol_exec_second = 1;   // BOHIP_CHOL_EXEC_SECOND=0: every executor workgroup serves every queue (until round 4).  1: the workgroups beyond one per CU take
                                     // throughput work only (early sums, bulk, waves) and leave when it is exhausted: N = 6000 4.60 -> 4.38 ms, N = 5000 3.27 -> 3.18
static int g_chol_exec_excl = 0;   // measurement build: more than half a CU's LDS per executor workgroup where the rule says one per CU (no effect measured: the dispatcher places them so already)
static int g_small_zero_copy = 1;          // small host calls: candidates read from the pinned block (0 in the measurement build: copied to HBM first)
static int g_chol_exec_early_tail = 0;    // the factorisation ALONE beyond 56 row tiles: two-piece Early sums for the last this-many blocks (the chain-bound tail)
static int g_chol_exec_early_split = 1;   // Early sums in two pieces (BOHIP_CHOL_EXEC_EARLY_SPLIT=0 in the measurement build: one piece, until round 6)
static int g_chol_exec_nbu = 2;      // BOHIP_CHOL_EXEC_NBU: rows behind the solve followers whose row step (Solve, Late) sits in the urgent queue
static int g_chol_exec_fast = -1;    // BOHIP_CHOL_EXEC_FAST: executor workgroups that never take bulk / wave tasks (-1: where CUs hold two executor workgroups and the chain paces, 33 ... 48 row tiles: up to 112)
static int g_chol_exec_bulk_edf = 0;   // BOHIP_CHOL_EXEC_BULK_EDF=1: bulk queue in earliest-deadline order from a host-side simulation (round-4 experiment: same total, see exec_task_list)
static int g_chol_exec_pairs = -1; // early sums and bulk updates are claimed two records (= both halves of a tile) at a time (BOHIP_CHOL_EXEC_PAIRS=1; 0: one); p >= 2: the bulk 2 p
                                    // records (p tiles) per claim.  -1: 1, and 2 from 72 row tiles on (N = 10^4 14.05 -> 13.8 ms, N = 12000 23.1 -> 22.6; N = 8000 the same, N = 6000 4.0 -> 4.25)
static int g_chol_exec_wgs = -1;  // executor workgroups (BOHIP_CHOL_EXEC_WGS); -1: by size -- ONE per free CU up to 32 row tiles, two from 45 on (see cholesky_exec)
static int g_append_alpha_inc = 1;   // incremental alpha on append (BOHIP_APPEND_ALPHA_INC=0: recomputed as W'(W r))
static int g_fuse_finish = 1;  // sigma^2 + acquisition + arg-max in k_trigemm_sq's epilogue (BOHIP_FUSE_FINISH=0: k_score + k_argmax_final)
static int64_t g_chunk_rows_forced = 0;   // BOHIP_CHUNK_ROWS: candidates per K*' chunk (tools: chunk-size sweeps), 0 = the rule in chunk_rows
static int g_halve_lo = -1, g_halve_hi = -1;   // BOHIP_TRIGEMM_HALVE (see trigemm_pieces)
static int g_ks8 = 1;  // 8-wave k_trigemm_sq (contraction index split inside the workgroup); BOHIP_KS8=0 selects the 4-wave loop
static int launch_gemm_nt(bohip_gp* g, const GemmNTParams& p, int batch = 1, hipStream_t st = nullptr, bool hi = false) {
    if (p.mt <= 0 || p.nt64 <= 0 || p.kc <= 0 || batch <= 0) return 0;
    if (hi) hipLaunchKernelGGL(k_gemm_nt_hi, dim3(p.mt * p.nt64, batch), dim3(GEMM_THREADS_8), glds3_lds_bytes<4>(), st ? st : g->stream, p);
    else hipLaunchKernelGGL(k_gemm_nt, dim3(p.mt * p.nt64, batch), dim3(GEMM_THREADS_8), glds3_lds_bytes<4>(), st ? st : g->stream, p);
    HIPCHK(hipGetLastError());
    return 0;
}

static int device_cus();
// Measurement builds only (make abl/libbohip_dev.so, -DBOHIP_DEV_KNOBS=1): the constants the sweeps under tools/ vary, by
// name.  The shipped library does not read them -- each has a measured default recorded beside its declaration above.
static void read_dev_knobs() {
#if BOHIP_DEV_KNOBS
    struct Knob { const char* name; int* v; int lo, hi; };
    static int chunk_rows = 0, dump = 0;
    const Knob knobs[] = {
        {"BOHIP_CHOL_DF2_HI", &g_chol_df2_hi, 0, 1}, {"BOHIP_CHOL_DF2_COAL", &g_chol_df2_coal, 0, 1}, {"BOHIP_CHOL_DF2_WIN", &g_chol_df2_win, 6, 10},
        {"BOHIP_CHOL_DF_TMAX", &g_chol_df_tmax, 0, CHOL_DF_TCAP}, {"BOHIP_INV_HI_H", &g_inv_hi_h, 0, 1 << 20}, {"BOHIP_INV_OVERLAP", &g_inv_overlap, 0, 1},
        {"BOHIP_CHOL_EXEC_PAIRS", &g_chol_exec_pairs, -1, 64}, {"BOHIP_CHOL_EXEC_FILL", &g_chol_exec_fill, 0, 2}, {"BOHIP_CHOL_COPY_EARLY", &g_chol_copy_early, 0, 1},
        {"BOHIP_CHOL_EXEC_URGENT", &g_chol_exec_urgent, 1, 1 << 20}, {"BOHIP_CHOL_NSF", &g_chol_nsf, 1, CH_NSF_MAX},
        {"BOHIP_CHOL_EXEC_PATIENCE_US", &g_chol_exec_patience_us, 0, 1 << 30}, {"BOHIP_CHOL_EXEC_FILL_INV", &g_chol_exec_fill_inv, 0, 1},
        {"BOHIP_CHOL_EXEC_INV_PAIRS", &g_chol_exec_inv_pairs, 0, 1}, {"BOHIP_CHOL_EXEC_WGS", &g_chol_exec_wgs, 1, 1 << 20}, {"BOHIP_KS8", &g_ks8, 0, 1},
        {"BOHIP_CHOL_EXEC_BULK_EDF", &g_chol_exec_bulk_edf, 0, 1}, {"BOHIP_CHOL_EXEC_FAST", &g_chol_exec_fast, -1, 1 << 20},
        {"BOHIP_CHOL_EXEC_SECOND", &g_chol_exec_second, 0, 1}, {"BOHIP_CHOL_EXEC_EARLY_SPLIT", &g_chol_exec_early_split, 0, 2}, {"BOHIP_CHOL_EXEC_EARLY_TAIL", &g_chol_exec_early_tail, 0, 1 << 20}, {"BOHIP_SMALL_ZERO_COPY", &g_small_zero_copy, 0, 1}, {"BOHIP_CHOL_EXEC_EXCL", &g_chol_exec_excl, 0, 1}, {"BOHIP_CHOL_EXEC_NBU", &g_chol_exec_nbu, 0, 16}, {"BOHIP_CHUNK_ROWS", &chunk_rows, 0, 1 << 30},
        {"BOHIP_TRIGEMM_HALVE_LO", &g_halve_lo, 0, 1 << 20}, {"BOHIP_TRIGEMM_HALVE_HI", &g_halve_hi, 0, 1 << 20}, {"BOHIP_FUSE_FINISH", &g_fuse_finish, 0, 1},
        {"BOHIP_APPEND_ALPHA_INC", &g_append_alpha_inc, 0, 1}, {"BOHIP_BULK_PIECES", &g_bulk_pieces, 0, 8}, {"BOHIP_SPLIT", &g_split, 0, 1},
        {"BOHIP_SMALL_R", &g_small_r, 0, SMALL_MAX}, {"BOHIP_SMALL_M", &g_small_m, 0, 1 << 20}, {"BOHIP_CHOL_DF_DUMP", &dump, 0, 1},
    };
    for (const Knob& k : knobs)
        if (const char* e = getenv(k.name)) *k.v = std::min(k.hi, std::max(k.lo, atoi(e)));
    g_chunk_rows_forced = chunk_rows;
    g_chol_df_dump = dump != 0;
#endif
}

static int one_time_kernel_setup() {
    // the >64 KB dynamic-LDS opt-in is a per-device function attribute
    static bool done_dev[64] = {false};
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    bool& done = done_dev[dev & 63];
    if (done) return 0;
    HIPCHK(hipFuncSetAttribute((const void*)k_potf2_inv, hipFuncAttributeMaxDynamicSharedMemorySize, POTF2_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)k_gemm_nt, hipFuncAttributeMaxDynamicSharedMemorySize, glds3_lds_bytes<4>()));
    HIPCHK(hipFuncSetAttribute((const void*)k_trigemm_sq<1>, hipFuncAttributeMaxDynamicSharedMemorySize, glds3_lds_bytes<4>()));
    HIPCHK(hipFuncSetAttribute((const void*)k_trigemm_sq<2>, hipFuncAttributeMaxDynamicSharedMemorySize, glds3_lds_bytes<4>()));
    HIPCHK(hipFuncSetAttribute((const void*)k_chol_chain, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)k_inv128, hipFuncAttributeMaxDynamicSharedMemorySize, POTF2_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)k_gemm_nt_pair, hipFuncAttributeMaxDynamicSharedMemorySize, glds3_lds_bytes<4>()));
    HIPCHK(hipFuncSetAttribute((const void*)k_gemm_nt_hi, hipFuncAttributeMaxDynamicSharedMemorySize, glds3_lds_bytes<4>()));
    HIPCHK(hipFuncSetAttribute((const void*)k_gemm_nt_quad, hipFuncAttributeMaxDynamicSharedMemorySize, glds3_lds_bytes<4>()));
    HIPCHK(hipFuncSetAttribute((const void*)k_chol_exec, hipFuncAttributeMaxDynamicSharedMemorySize, std::max<size_t>(glds3_lds_bytes<4>(), 84 * 1024)));
    HIPCHK(hipFuncSetAttribute((const void*)k_trimv_stream<TRIMV_D>, hipFuncAttributeMaxDynamicSharedMemorySize, trimv_lds_bytes(TRIMV_D)));
    // The library's switches (README "Environment"): the forms of the factorisation the tests select, the bound of its
    // waits, the ascent drivers, the small-batch pass.  Everything else that used to be read here is a constant now; the
    // sweeps under tools/ that vary those constants run against csrc/abl/libbohip_dev.so (read_dev_knobs below).
    if (const char* e = getenv("BOHIP_CHOL_DATAFLOW")) g_chol_df = atoi(e);
    if (const char* e = getenv("BOHIP_CHOL_DF2_MIN")) g_chol_df2_min = atoi(e);
    if (const char* e = getenv("BOHIP_CHOL_DF2_LL")) g_chol_df2_ll = atoi(e);
    if (const char* e = getenv("BOHIP_CHOL_SPIN_US")) g_chol_spin_ticks = 100ull * (unsigned long long)std::max(1, atoi(e));
    if (const char* e = getenv("BOHIP_CHOL_EXEC")) g_chol_exec = atoi(e);
    if (const char* e = getenv("BOHIP_CHOL_EXEC_MIN")) g_chol_exec_min = atoi(e);
    if (const char* e = getenv("BOHIP_CHOL_INV_G")) g_chol_inv_g = std::min(64, std::max(0, atoi(e)));
    if (const char* e = getenv("BOHIP_CHOL_INV_GRP_MIN")) g_chol_inv_grp_min = std::max(0, atoi(e));
    if (const char* e = getenv("BOHIP_ASC_WG_NMAX")) g_asc_wg_nmax = std::max(0, atoi(e));
    if (const char* e = getenv("BOHIP_ASC_LOCKSTEP")) { g_asc_lockstep = atoi(e) == 1; g_asc_fold = atoi(e) != 2; }
    if (const char* e = getenv("BOHIP_SMALL_MFMA")) g_small_mfma = atoi(e);
    g_chol_df_strict = getenv("BOHIP_CHOL_DF_STRICT") != nullptr;
    read_dev_knobs();
    done = true;
    return 0;
}

// ---- A1-A3: full rebuild --------------------------------------------------------------------------
static int launch_rows_trimv(bohip_gp* g, const double* W, int64_t N0, const double* rows, int P, double* out, int upper);
static int compute_alpha(bohip_gp* g) {
    // alpha = W'(W (y - beta)): two passes of the row-wise kernel (W, then the resident W' as an upper-triangular
    // K-major matrix) with one right-hand side
    const int64_t N = g->n;
    g->alpha_inc_run = 0;
    // (dr = y - beta and dt = W (y - beta) stay behind for the incremental update of the appends: nothing else may write them between refits)
    hipLaunchKernelGGL(k_sub_mean, dim3((N + 255) / 256), dim3(256), 0, g->stream, g->dy, g->beta, N, g->dr);
    HIPCHK(hipGetLastError());
    CHK(launch_rows_trimv(g, g->dW, N, g->dr, 1, g->dt, 0));
    CHK(launch_rows_trimv(g, g->dWT, N, g->dt, 1, g->dalpha, 1));
    // alpha' into the first padding row of W (cols < N); W[N][N] = 1 meets K*[N] = 0.
    HIPCHK(hipMemcpyAsync(g->dW + N * g->ld, g->dalpha, (size_t)N * 8, hipMemcpyDeviceToDevice, g->stream));
    return 0;
}

static int check_info(bohip_gp* g) {
    int info = 0;
    if (g->hpin) {     // through the pinned block: a plain DMA command (a pageable destination is staged by the runtime)
        int* pw = reinterpret_cast<int*>(g->hpin + 3 * SMALL_R) + 1;
        HIPCHK(hipMemcpyAsync(pw, g->dinfo, sizeof(int), hipMemcpyDeviceToHost, g->stream));
        HIPCHK(hipStreamSynchronize(g->stream));
        info = *pw;
    } else {
        HIPCHK(hipMemcpyAsync(&info, g->dinfo, sizeof(int), hipMemcpyDeviceToHost, g->stream));
        HIPCHK(hipStreamSynchronize(g->stream));
    }
    if (info != 0) {
        g->pivot = info;
        g->stale = true;
        return fail(BOHIP_E_NOTPD, "kernel matrix not positive definite at pivot " + std::to_string(info));
    }
    g->pivot = 0;
    return 0;
}

// ---- W = L^-1 by blocks:  [L11 0; L21 L22]^-1 = [W11 0; -W22 L21 W11, W22].  With W' kept beside W both products are
// K-major x K-major:   S'[c][i]  =  sum_k W11'[c][k] L21[i][k]     (A = W' block, upper-triangular: k >= c)
//                      W21[i][c] = -sum_k W22[i][k]  S'[c][k]      (A = W22, lower-triangular: k <= i); W21' goes to W' too
// S' lives in the (otherwise unused) upper-right part of the scratch matrix dS, whose lower part holds the solved panels
// (= L21; dL receives them only with the final copy).
// inverse_level: all pairs of half-size h (tiles) inside the diagonal region [t0, t0 + nt), one batched launch each.
static int inverse_level(bohip_gp* g, hipStream_t st, int t0, int nt, int h) {
    const int64_t ld = g->ld, base = (int64_t)t0 * TILE * (ld + 1);
    const int pairs = (nt + 2 * h - 1) / (2 * h);
    const int64_t zs = (int64_t)2 * h * TILE * (ld + 1);
    const int64_t off21 = (int64_t)h * TILE * ld, off12 = (int64_t)h * TILE, off22 = (int64_t)h * TILE * (ld + 1);
    GemmNTParams a{};
    a.A = g->dWT + base; a.lda = ld; a.B = g->dS + base + off21; a.ldb = ld; a.C = g->dS + base + off12; a.ldc = ld;
    a.zA = a.zB = a.zC = zs;
    a.mt = h; a.nt64 = 2 * h; a.kc = h * (TILE / KC); a.alpha = 1.0; a.beta = 0.0; a.klo_from_m = 1;
    a.row_t0 = 0; a.row_ts = 2 * h; a.col_t0 = h; a.col_ts = 2 * h; a.total_t = nt;
    const bool big = h >= g_inv_hi_h;   // large levels are throughput launches: two workgroups per CU, stores through LDS
    a.coalesced = big ? 1 : 0;
    CHK(launch_gemm_nt(g, a, pairs, st, big));
    GemmNTParams b{};
    b.A = g->dW + base + off22; b.lda = ld; b.B = g->dS + base + off12; b.ldb = ld; b.C = g->dW + base + off21; b.ldc = ld;
    b.CT = g->dWT + base + off12; b.ldct = ld;
    b.zA = b.zB = b.zC = b.zCT = zs;
    b.mt = h; b.nt64 = 2 * h; b.kc = h * (TILE / KC); b.alpha = -1.0; b.beta = 0.0; b.khi_from_m = 1;
    b.row_t0 = h; b.row_ts = 2 * h; b.col_t0 = 0; b.col_ts = 2 * h; b.total_t = nt;
    b.coalesced = big ? 1 : 0;
    CHK(launch_gemm_nt(g, b, pairs, st, big));
    return 0;
}
// inverse_join: the leading P tiles are inverted, so are the nb tiles behind them: fill W[P:P+nb, 0:P] (and its transpose).
static int inverse_join(bohip_gp* g, hipStream_t st, int P, int nb) {
    if (P <= 0 || nb <= 0) return 0;
    const int64_t ld = g->ld;
    const int64_t off21 = (int64_t)P * TILE * ld, off12 = (int64_t)P * TILE, off22 = (int64_t)P * TILE * (ld + 1);
    GemmNTParams a{};
    a.A = g->dWT; a.lda = ld; a.B = g->dS + off21; a.ldb = ld; a.C = g->dS + off12; a.ldc = ld;
    a.mt = P; a.nt64 = 2 * nb; a.kc = P * (TILE / KC); a.alpha = 1.0; a.beta = 0.0; a.klo_from_m = 1;
    CHK(launch_gemm_nt(g, a, 1, st));
    GemmNTParams b{};
    b.A = g->dW + off22; b.lda = ld; b.B = g->dS + off12; b.ldb = ld; b.C = g->dW + off21; b.ldc = ld;
    b.CT = g->dWT + off12; b.ldct = ld;
    b.mt = nb; b.nt64 = 2 * P; b.kc = nb * (TILE / KC); b.alpha = -1.0; b.beta = 0.0; b.khi_from_m = 1;
    CHK(launch_gemm_nt(g, b, 1, st));
    return 0;
}

// ---- A2, dataflow form (kernels_chol.hip): ONE persistent chain launch on the critical stream; the panel followers and the
// flag-gated trailing updates of every block are enqueued up front on the side stream.  No host events inside the factorisation.
static size_t chol_abort_word(int T) { return (size_t)T * (CH_PANELS + 7) + (size_t)T * T * (CH_PANELS + 1); }   // see the layout in cholesky_dataflow
static size_t chol_inv_word(int T) { return chol_abort_word(T) + 8; }      // the inverse queue's counters: 4 words per tile (i, j)
static size_t chol_xp3_word(int T) { return chol_inv_word(T) + (size_t)4 * T * T; }    // per-panel flags of S(k+3, k) (8 per block), then pre3 (4 per block)
static size_t chol_flag_words(int T) { return chol_xp3_word(T) + (size_t)12 * T; }       // abort word + the executor's queue cursors + those
static CholFlags chol_flags_layout_at(unsigned* base, double* idl, int T);
static CholFlags chol_flags_layout(bohip_gp* g, int T) { return chol_flags_layout_at(g->dchol_flags, g->dchol_idl, T); }
static CholFlags chol_flags_layout_at(unsigned* base, double* idl, int T) {
    CholFlags fl{};
    fl.T = T;
    fl.panel = base;
    fl.solved = fl.panel + (size_t)T * CH_PANELS;
    fl.crit = fl.solved + T;
    fl.rest = fl.crit + T;
    fl.col = fl.rest + T;
    fl.farall = fl.col + T;
    fl.fol = fl.farall + T;
    fl.colall = fl.fol + T;
    fl.colr = fl.colall + T;
    fl.xp = fl.colr + (size_t)T * T;
    fl.abort = fl.xp + (size_t)T * T * CH_PANELS;
    fl.xp3 = base + chol_xp3_word(T);
    fl.pre3 = fl.xp3 + (size_t)T * CH_PANELS;
    fl.w16_g = idl;
    fl.resident = fl.abort + 7;   // (the word between the executor's six queue cursors and the inverse queues' counters)
    fl.spin_ticks = g_chol_spin_ticks;
    fl.crit_want = 16u;   // the row-(k+2) update: 4 workgroups x 4 storing waves
    fl.panel_want = 3u;   // three publishing waves per panel
    return fl;
}
static int cholesky_dataflow(bohip_gp* g, int T) {
    const int64_t ld = g->ld;
    CholFlags fl = chol_flags_layout(g, T);
    g->w_seeded = false;
    HIPCHK(hipMemsetAsync(g->dchol_flags, 0, chol_flag_words(T) * sizeof(unsigned), g->stream));
    HIPCHK(hipEventRecord(g->ev_panels, g->stream));           // K and the cleared flags are in place
    // Four launches that live for the whole factorisation and talk through flags:
    //   critical stream: the chain (row owners, critical followers of rows k+1 / k+2, gated update of row k+2)
    //   two more       : one follower workgroup per row >= 3, two column-updater workgroups per row >= 3
    hipLaunchKernelGGL(k_chol_chain, dim3(T > 1 ? 8 : 1), dim3(CH_THREADS), CH_LDS_BYTES, g->stream, g->dL, ld, g->dS, T, fl, g->dinfo, g->dW, g->dWT);
    HIPCHK(hipGetLastError());
    if (T > 3) {
        HIPCHK(hipStreamWaitEvent(g->col_stream, g->ev_panels, 0));
        HIPCHK(hipStreamWaitEvent(g->side_stream, g->ev_panels, 0));
        HIPCHK(hipStreamWaitEvent(g->inv_stream, g->ev_panels, 0));
        hipLaunchKernelGGL(k_chol_rows, dim3(T - 3), dim3(CH_THREADS), WK_LDS_DOUBLES * 8, g->col_stream, g->dL, ld, g->dS, T, fl);
        hipLaunchKernelGGL(k_chol_cols, dim3(2 * (T - 3)), dim3(GEMM_THREADS), 0, g->side_stream, g->dL, ld, g->dS, T, fl);
        HIPCHK(hipGetLastError());
    }
    if (T > 3) {   // the flagged launches below stay behind the persistent workgroups (k_chol_gate)
        hipLaunchKernelGGL(k_chol_gate, dim3(1), dim3(64), 0, g->inv_stream, fl.resident, 8u + 3u * (unsigned)(T - 3), fl.abort, fl.spin_ticks);
        HIPCHK(hipGetLastError());
    }
    for (int k = 0; k + 3 < T; ++k) {
        // fourth stream: the rest of block k's update -- rows >= k+3, columns >= k+2 -- on the MFMA engine; each workgroup
        // waits for the last panel of the two rows of L(:, k) it reads
        GemmNTParams f{};
        f.A = g->dS + (int64_t)(k + 3) * TILE * ld + (int64_t)k * TILE; f.lda = ld;
        f.B = g->dS + (int64_t)(k + 2) * TILE * ld + (int64_t)k * TILE; f.ldb = ld;
        f.C = g->dL + (int64_t)(k + 3) * TILE * ld + (int64_t)(k + 2) * TILE; f.ldc = ld;
        f.mt = T - (k + 3); f.nt64 = 2 * (T - (k + 2)); f.kc = TILE / KC; f.alpha = -1.0; f.beta = 1.0;
        f.diag_skip = 1; f.row0 = (int64_t)(k + 3) * TILE; f.col0 = (int64_t)(k + 2) * TILE;
        f.wait_flag = fl.xp + ((size_t)k * T + (k + 3)) * CH_PANELS + (CH_PANELS - 1); f.wait_val = 1u; f.wait_stride_ti = CH_PANELS;
        f.wait_flag2 = fl.xp + ((size_t)k * T + (k + 2)) * CH_PANELS + (CH_PANELS - 1); f.wait_val2 = 1u; f.wait_stride_tj2 = CH_PANELS;
        f.signal = fl.colall + k;          // (selects the agent-scope stores; nobody waits for the whole launch)
        f.signal_row0 = fl.rest + k;       // row k+3: what the gated update of block k+1 starts from
        f.signal_col0 = fl.farall + k;     // column k+2: what block k+1's column updaters write next
        f.first_row_col = 1; f.abort_flag = fl.abort; f.spin_ticks = g_chol_spin_ticks;
        CHK(launch_gemm_nt(g, f, 1, g->inv_stream));
    }
    if (T > 3) {
        HIPCHK(hipEventRecord(g->ev_bulk, g->col_stream));
        HIPCHK(hipStreamWaitEvent(g->stream, g->ev_bulk, 0));
        HIPCHK(hipEventRecord(g->ev_gate, g->side_stream));
        HIPCHK(hipStreamWaitEvent(g->stream, g->ev_gate, 0));
        HIPCHK(hipEventRecord(g->ev_inv, g->inv_stream));
        HIPCHK(hipStreamWaitEvent(g->stream, g->ev_inv, 0));
    }
    if (T > 1) {
        hipLaunchKernelGGL(k_copy_offdiag_tiles, dim3(T * (T - 1) / 2), dim3(256), 0, g->stream, g->dS, g->dL, ld, T);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// ---- A2, dataflow form for LARGE T.  The form above keeps T - 3 row followers and 2 (T - 3) column updaters resident for the
// whole factorisation: at T = 63 they sit on ~180 of the 256 CUs and halve the rate of every GEMM beside them, and its far
// updates are one K = 128 launch per block over the whole trailing matrix (~16 TF/s).  Here only the chain (8 workgroups)
// and one inverter workgroup are persistent:
//   rows >= k+3 of L(:, k)      one launch: product with W_kk' (k_chol_inverter raises solved[k] a few us after the pivot block)
//   columns k+1 .. 4m+6, m=k/4  one flagged K = 128 launch (first row = what the chain reads next, counted in rest[k])
//   columns 4m+7 .. 4m+10       K = 512, plain, once group m (four blocks) is solved; the next group's flagged update waits for it
//   columns >= 4m+11            K = 512, plain, behind it on the same low-priority stream
// Every launch with an in-kernel wait is on ONE in-order stream and waits only for the chain or for launches before it on that
// stream; the plain launches are released by host events: no spinning workgroup can keep a producer off the chip.
static int cholesky_dataflow2(bohip_gp* g, int T) {
    const int64_t ld = g->ld;
    // the handle's four streams and no more: a fifth and sixth stream ended up sharing a hardware queue on the first handle of a
    // process (the long plain launches then sat in front of the flagged ones: 14.2 instead of 12.2 ms at N = 10000)
    hipStream_t flagged_stream = g->col_stream, bulk_stream = g->side_stream;
    CholFlags fl = chol_flags_layout(g, T);
    const int win = g_chol_df2_win;
    fl.mode2 = win;
    HIPCHK(hipMemsetAsync(g->dchol_flags, 0, chol_flag_words(T) * sizeof(unsigned), g->stream));
    HIPCHK(hipEventRecord(g->ev_panels, g->stream));           // K and the cleared flags are in place
    hipLaunchKernelGGL(k_chol_chain, dim3(9), dim3(CH_THREADS), CH_LDS_BYTES, g->stream, g->dL, ld, g->dS, T, fl, g->dinfo, g->dW, g->dWT);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamWaitEvent(flagged_stream, g->ev_panels, 0));
    HIPCHK(hipStreamWaitEvent(bulk_stream, g->ev_panels, 0));
    while ((int)g->ev_tier.size() < 2 * (T / 4 + 1)) {
        hipEvent_t ev;
        HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        g->ev_tier.push_back(ev);
    }
    // ONE in-order stream of flagged launches.  Launch k carries Update(k) and Solve(k+1) in one grid (k_gemm_nt_pair), so that
    // the row solve is already resident when the inverse of its diagonal block arrives:
    //   Solve(k)   row i waits for solved[k] and for tile (i, k) from Update(k-1); counts S(i, k) into colr[k T + i] (16 = done)
    //   Update(k)  tile (i, j) waits for S(i, k) and S(j, k) (rows k+1, k+2: the chain's last-panel flags); its column k+1
    //              counts into xp[(k T + i) 8] (rows >= k+3 have no other use for that word), its first row into rest[k]
    hipStream_t ss = flagged_stream;
    auto solve_params = [&](int k) {
        GemmNTParams sv{};   // S(i, k) = A(i, k) W_kk' for i >= k+3
        sv.A = g->dL + (int64_t)(k + 3) * TILE * ld + (int64_t)k * TILE; sv.lda = ld;
        sv.B = g->dW + (int64_t)k * TILE * (ld + 1); sv.ldb = ld;
        sv.C = g->dS + (int64_t)(k + 3) * TILE * ld + (int64_t)k * TILE; sv.ldc = ld;
        sv.mt = T - (k + 3); sv.nt64 = 2; sv.kc = TILE / KC; sv.alpha = 1.0; sv.beta = 0.0;
        if (k == 0) {
            sv.wait_flag = fl.solved + k; sv.wait_val = 1u; sv.wait_stride_ti = 0;
        } else {
            sv.wait_flag = fl.xp + ((size_t)(k - 1) * T + (k + 3)) * CH_PANELS; sv.wait_val = 16u; sv.wait_stride_ti = CH_PANELS;
            sv.wait_flag2 = fl.solved + k; sv.wait_val2 = 1u; sv.wait_stride_tj2 = 0;
        }
        sv.abort_flag = fl.abort; sv.spin_ticks = g_chol_spin_ticks;
        sv.signal = fl.farall + k;         // (selects the agent-scope stores: the update beside it reads these tiles)
        sv.signal_rows = fl.colr + (size_t)k * T + (k + 3); sv.signal_rows_ntj = 2; sv.signal_rows_stride = 1;
        return sv;
    };
    auto update_params = [&](int k) {
        const int c_hi = near_last_col(T, k, win);
        GemmNTParams f{};    // tiles (i, j), i >= k+3, k+1 <= j <= c_hi:  -= S(i, k) S(j, k)'
        f.A = g->dS + (int64_t)(k + 3) * TILE * ld + (int64_t)k * TILE; f.lda = ld;
        f.B = g->dS + (int64_t)(k + 1) * TILE * ld + (int64_t)k * TILE; f.ldb = ld;
        f.C = g->dL + (int64_t)(k + 3) * TILE * ld + (int64_t)(k + 1) * TILE; f.ldc = ld;
        f.mt = T - (k + 3); f.nt64 = 2 * (c_hi - k); f.kc = TILE / KC; f.alpha = -1.0; f.beta = 1.0;
        f.diag_skip = 1; f.row0 = (int64_t)(k + 3) * TILE; f.col0 = (int64_t)(k + 1) * TILE;
        f.wait_flag = fl.colr + (size_t)k * T + (k + 3); f.wait_val = 16u; f.wait_stride_ti = 1;
        f.wait_flag2 = fl.xp + ((size_t)k * T + (k + 1)) * CH_PANELS + (CH_PANELS - 1); f.wait_val2 = 1u; f.wait_stride_tj2 = CH_PANELS;
        f.wait2_tj2_max = 2; f.wait2_rows = 1;
        f.signal = fl.colall + k;          // (selects the agent-scope stores; nobody waits for the whole launch)
        f.signal_row0 = fl.rest + k;       // row k+3: what the chain's followers and gated updates of block k+1 start from
        f.signal_rows = fl.xp + ((size_t)k * T + (k + 3)) * CH_PANELS; f.signal_rows_ntj = 2; f.signal_rows_stride = CH_PANELS;
        f.first_row_col = 1; f.abort_flag = fl.abort; f.spin_ticks = g_chol_spin_ticks;
        return f;
    };
    if (T > 3) CHK(launch_gemm_nt(g, solve_params(0), 1, ss, true));
    for (int k = 0; k + 3 < T; ++k) {
        const int m = k / 4;
        if (k % 4 == 0 && m >= 1 && 4 * (m - 1) + win + 1 <= T - 1)   // columns 4m+3 .. 4m+6 carry group m-1 only after its first K = 512 launch
            HIPCHK(hipStreamWaitEvent(ss, g->ev_tier[2 * (m - 1) + 1], 0));
        const GemmNTParams f = update_params(k);
        if (k + 4 < T) {
            const GemmNTParams sv = solve_params(k + 1);
            hipLaunchKernelGGL(k_gemm_nt_pair, dim3(f.mt * f.nt64 + sv.mt * sv.nt64), dim3(GEMM_THREADS_8), glds3_lds_bytes<4>(), ss, f, sv,
                               f.mt * f.nt64);
            HIPCHK(hipGetLastError());
        } else {
            CHK(launch_gemm_nt(g, f, 1, ss, true));
        }
        if (k % 4 == 3 && 4 * m + win + 1 <= T - 1) {
            // group m is solved for every row once this launch has finished (the update waited for every row of S(:, k))
            HIPCHK(hipEventRecord(g->ev_tier[2 * m], ss));
            HIPCHK(hipStreamWaitEvent(bulk_stream, g->ev_tier[2 * m], 0));
            for (int part = 0; part < 2; ++part) {
                const int c0 = part == 0 ? 4 * m + win + 1 : 4 * m + win + 5, c1 = part == 0 ? std::min(T - 1, 4 * m + win + 4) : T - 1;
                if (c0 <= c1) {
                    GemmNTParams q{};
                    q.A = g->dS + (int64_t)c0 * TILE * ld + (int64_t)(4 * m) * TILE; q.lda = ld;
                    q.B = q.A; q.ldb = ld;
                    q.C = g->dL + (int64_t)c0 * TILE * (ld + 1); q.ldc = ld;
                    q.mt = T - c0; q.nt64 = 2 * (c1 - c0 + 1); q.kc = 4 * (TILE / KC); q.alpha = -1.0; q.beta = 1.0;
                    q.diag_skip = 1; q.row0 = (int64_t)c0 * TILE; q.col0 = (int64_t)c0 * TILE;
                    q.coalesced = g_chol_df2_coal;
                    CHK(launch_gemm_nt(g, q, 1, bulk_stream, g_chol_df2_hi != 0));
                }
                if (part == 0) HIPCHK(hipEventRecord(g->ev_tier[2 * m + 1], bulk_stream));
            }
        }
    }
    HIPCHK(hipEventRecord(g->ev_inv, ss));
    HIPCHK(hipStreamWaitEvent(g->stream, g->ev_inv, 0));
    HIPCHK(hipEventRecord(g->ev_blk, bulk_stream));
    HIPCHK(hipStreamWaitEvent(g->stream, g->ev_blk, 0));
    hipLaunchKernelGGL(k_copy_offdiag_tiles, dim3(T * (T - 1) / 2), dim3(256), 0, g->stream, g->dS, g->dL, ld, T);
    HIPCHK(hipGetLastError());
    g->w_seeded = true;
    return 0;
}

// ---- the same with LEFT-LOOKING window updates (BOHIP_CHOL_DF2_LL=1).  In cholesky_dataflow2 a quarter of the flops run as
// K = 128 read-modify-writes of the window columns (~15 TF/s).  Here a column receives its updates just in time and all at once:
//   bulk      group m (blocks 4m..4m+3), K = 512, goes to the columns >= 4m+8 only: Qa(m) = columns 4m+8..4m+11 (counted into
//             col[m]: the flagged launches that touch those columns wait for it in-kernel), Qb(m) = the rest
//   block k   column c = k+1, rows >= k+3:  -= S(i, ks..k) S(c, ks..k)'  with ks = first block of the group BEFORE c's -- K up to 1024,
//             one read-modify-write per tile; the two tiles (k+3, k+2), (k+3, k+3) the chain reads next the same way; Solve(k+1)
//             -- all four in one grid (k_gemm_nt_quad) on the one flagged stream.
static int cholesky_dataflow3(bohip_gp* g, int T) {
    const int64_t ld = g->ld;
    hipStream_t ss = g->col_stream, bulk_stream = g->side_stream;
    CholFlags fl = chol_flags_layout(g, T);
    fl.mode2 = 100;
    HIPCHK(hipMemsetAsync(g->dchol_flags, 0, chol_flag_words(T) * sizeof(unsigned), g->stream));
    HIPCHK(hipEventRecord(g->ev_panels, g->stream));           // K and the cleared flags are in place
    hipLaunchKernelGGL(k_chol_chain, dim3(9), dim3(CH_THREADS), CH_LDS_BYTES, g->stream, g->dL, ld, g->dS, T, fl, g->dinfo, g->dW, g->dWT);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamWaitEvent(ss, g->ev_panels, 0));
    HIPCHK(hipStreamWaitEvent(bulk_stream, g->ev_panels, 0));
    while ((int)g->ev_tier.size() < 2 * (T / 4 + 1)) {
        hipEvent_t ev;
        HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        g->ev_tier.push_back(ev);
    }
    unsigned* qa_cnt = fl.col;   // [T] words, one per group used
    auto qa_dims = [&](int m, int& mt, int& nt64) {   // Qa(m): columns 4m+8 .. 4m+11, rows >= 4m+8
        const int c0 = 4 * m + 8, c1 = std::min(T - 1, 4 * m + 11);
        mt = T - c0; nt64 = 2 * (c1 - c0 + 1);
    };
    auto bulk_wait = [&](GemmNTParams& f, int c) {    // column c carries the groups <= c/4 - 2 once Qa(c/4 - 2) has stored
        const int m = c / 4 - 2;
        if (m < 0) return;
        int mt, nt64; qa_dims(m, mt, nt64);
        f.wait_flag3 = qa_cnt + m; f.wait_val3 = 8u * (unsigned)(mt * nt64);
    };
    auto solve_params = [&](int k) {
        GemmNTParams sv{};   // S(i, k) = A(i, k) W_kk' for i >= k+3
        sv.A = g->dL + (int64_t)(k + 3) * TILE * ld + (int64_t)k * TILE; sv.lda = ld;
        sv.B = g->dW + (int64_t)k * TILE * (ld + 1); sv.ldb = ld;
        sv.C = g->dS + (int64_t)(k + 3) * TILE * ld + (int64_t)k * TILE; sv.ldc = ld;
        sv.mt = T - (k + 3); sv.nt64 = 2; sv.kc = TILE / KC; sv.alpha = 1.0; sv.beta = 0.0;
        if (k == 0) {
            sv.wait_flag = fl.solved + k; sv.wait_val = 1u; sv.wait_stride_ti = 0;
        } else {
            sv.wait_flag = fl.xp + ((size_t)(k - 1) * T + (k + 3)) * CH_PANELS; sv.wait_val = 16u; sv.wait_stride_ti = CH_PANELS;
            sv.wait_flag2 = fl.solved + k; sv.wait_val2 = 1u; sv.wait_stride_tj2 = 0;
        }
        sv.abort_flag = fl.abort; sv.spin_ticks = g_chol_spin_ticks;
        sv.signal = fl.farall + k;
        sv.signal_rows = fl.colr + (size_t)k * T + (k + 3); sv.signal_rows_ntj = 2; sv.signal_rows_stride = 1;
        return sv;
    };
    // tiles (i, c), i = r0 .. r0 + mt - 1, left-looking over the blocks ks .. k
    auto ll_params = [&](int k, int c, int r0, int mt) {
        const int ks = 4 * std::max(c / 4 - 1, 0), nb = k - ks + 1;
        GemmNTParams f{};
        f.A = g->dS + (int64_t)r0 * TILE * ld + (int64_t)ks * TILE; f.lda = ld;
        f.B = g->dS + (int64_t)c * TILE * ld + (int64_t)ks * TILE; f.ldb = ld;
        f.C = g->dL + (int64_t)r0 * TILE * ld + (int64_t)c * TILE; f.ldc = ld;
        f.mt = mt; f.nt64 = 2; f.kc = nb * (TILE / KC); f.alpha = -1.0; f.beta = 1.0;
        f.diag_skip = 1; f.row0 = (int64_t)r0 * TILE; f.col0 = (int64_t)c * TILE;
        f.wait_flag = fl.colr + (size_t)k * T + r0; f.wait_val = 16u; f.wait_stride_ti = 1;   // S(i, k) from Solve(k)
        if (c <= k + 2) {   // operand row c is one of the chain's two: its last-panel flag
            f.wait_flag2 = fl.xp + ((size_t)k * T + c) * CH_PANELS + (CH_PANELS - 1); f.wait_val2 = 1u; f.wait_stride_tj2 = 0;
        }
        bulk_wait(f, c);
        f.signal = fl.colall + k;          // (selects the agent-scope stores)
        f.signal_row0 = fl.rest + k;       // only the launches whose FIRST row is row k+3 count there (all three per block do)
        f.abort_flag = fl.abort; f.spin_ticks = g_chol_spin_ticks;
        return f;
    };
    if (T > 3) CHK(launch_gemm_nt(g, solve_params(0), 1, ss, true));
    for (int k = 0; k + 3 < T; ++k) {
        const int m = k / 4;
        GemmNTParams r2 = ll_params(k, k + 2, k + 3, 1), r3 = ll_params(k, k + 3, k + 3, 1);
        GemmNTParams col = ll_params(k, k + 1, k + 3, T - (k + 3));
        col.signal_rows = fl.xp + ((size_t)k * T + (k + 3)) * CH_PANELS; col.signal_rows_ntj = 2; col.signal_rows_stride = CH_PANELS;
        GemmNTParams sv{};
        int n3 = 0;
        if (k + 4 < T) { sv = solve_params(k + 1); n3 = sv.mt * sv.nt64; }
        const int n0 = r2.mt * r2.nt64, n1 = r3.mt * r3.nt64, n2 = col.mt * col.nt64;
        hipLaunchKernelGGL(k_gemm_nt_quad, dim3(n0 + n1 + n2 + n3), dim3(GEMM_THREADS_8), glds3_lds_bytes<4>(), ss, r2, r3, col, sv, n0, n1, n2);
        HIPCHK(hipGetLastError());
        if (k % 4 == 2 && 4 * m + 8 <= T - 1) {
            // this grid held Solve(4m+3): group m is solved for every row, its K = 512 update of the columns >= 4m+8 can go
            HIPCHK(hipEventRecord(g->ev_tier[2 * m], ss));
            HIPCHK(hipStreamWaitEvent(bulk_stream, g->ev_tier[2 * m], 0));
            for (int part = 0; part < 2; ++part) {
                const int c0 = part == 0 ? 4 * m + 8 : 4 * m + 12, c1 = part == 0 ? std::min(T - 1, 4 * m + 11) : T - 1;
                if (c0 > c1) continue;
                GemmNTParams q{};
                q.A = g->dS + (int64_t)c0 * TILE * ld + (int64_t)(4 * m) * TILE; q.lda = ld;
                q.B = q.A; q.ldb = ld;
                q.C = g->dL + (int64_t)c0 * TILE * (ld + 1); q.ldc = ld;
                q.mt = T - c0; q.nt64 = 2 * (c1 - c0 + 1); q.kc = 4 * (TILE / KC); q.alpha = -1.0; q.beta = 1.0;
                q.diag_skip = 1; q.row0 = (int64_t)c0 * TILE; q.col0 = (int64_t)c0 * TILE;
                if (part == 0) q.signal = qa_cnt + m;   // agent-scope stores + the counter the flagged launches wait for
                else q.coalesced = g_chol_df2_coal;
                CHK(launch_gemm_nt(g, q, 1, bulk_stream, g_chol_df2_hi != 0));
            }
        }
    }
    HIPCHK(hipEventRecord(g->ev_inv, ss));
    HIPCHK(hipStreamWaitEvent(g->stream, g->ev_inv, 0));
    HIPCHK(hipEventRecord(g->ev_blk, bulk_stream));
    HIPCHK(hipStreamWaitEvent(g->stream, g->ev_blk, 0));
    hipLaunchKernelGGL(k_copy_offdiag_tiles, dim3(T * (T - 1) / 2), dim3(256), 0, g->stream, g->dS, g->dL, ld, T);
    HIPCHK(hipGetLastError());
    g->w_seeded = true;
    return 0;
}

// ---- A2, executor form (kernels_exec.hip): the chain + ONE persistent kernel that pulls tile tasks from six in-order queues
// (four feed the factorisation's pivot chain, two grow W = L^-1 behind it).
// The records are a pure function of (buffer addresses, ld, T): built on the host at the first factorisation with this T,
// then resident.  Layout of the counters inside the flag area (all zeroed per factorisation):
//   tile (i, c), i >= c+2:  ver = xp[((c-1) T + i) 8 + 0], pver = xp[.. + 1]   (the chain uses xp[(k T + i) 8 + p] for i = k+1, k+2 only)
//   tile (c+1, c):          ver = farall[c], pver = fol[c];      tile (c, c):  ver = colall[c], pver = col[c]
//   sver(i, k) = colr[k T + i];   queue cursors = the EX_NQ = 6 words behind the abort word
static void exec_task_list(double* dL, double* dS, double* dW, double* dWT, unsigned* flag_base, int64_t ld, int T, int CH_NSF, int inv_g,
                           std::vector<ExTask>& all, int* qbeg, int inv_grp_min = -1) {
    if (inv_grp_min < 0) inv_grp_min = g_chol_inv_grp_min;   // (the test hook passes its own: no process-wide state is swapped)
    const CholFlags fl = chol_flags_layout_at(flag_base, nullptr, T);
    auto widx = [&](const unsigned* p) { return (uint32_t)(p - flag_base); };
    auto nb = [](int c) { return std::max(c / 4 - 1, 0); };         // bulk groups that touch column c
    auto ks = [](int c) { return 4 * std::max(c / 4 - 1, 0); };     // first block of column c's window (= 4 nb)
    auto ver = [&](int i, int c) {
        return i >= c + 2 ? widx(fl.xp + ((size_t)(c - 1) * T + i) * CH_PANELS) : (i == c + 1 ? widx(fl.farall + c) : widx(fl.colall + c));
    };
    auto pver = [&](int i, int c) {
        return i >= c + 2 ? widx(fl.xp + ((size_t)(c - 1) * T + i) * CH_PANELS + 1) : (i == c + 1 ? widx(fl.fol + c) : widx(fl.col + c));
    };
    auto sver = [&](int i, int k) { return widx(fl.colr + (size_t)k * T + i); };
    auto Sp = [&](int i, int kb) { return dS + (int64_t)i * TILE * ld + (int64_t)kb * TILE; };
    auto Ap = [&](int i, int c) { return dL + (int64_t)i * TILE * ld + (int64_t)c * TILE; };
    // P(i, c): the mirror tile of the scratch matrix (its upper triangle and diagonal tiles are free during the factorisation)
    auto Pp = [&](int i, int c) { return dS + (int64_t)c * TILE * ld + (int64_t)i * TILE; };
    // last block the Early sum of tile (i, c) contains; Late adds the two blocks behind it
    // (the three tiles of a row that the chain kernel finishes itself -- i <= c + 2 -- get their last block from its gated updates:
    // the executor's share ends one block earlier)
    auto e_of = [](int i, int c) { return i <= c + 2 ? i - 6 : c - 3; };
    std::vector<ExTask> q[EX_NQ];
    {   // (rough upper bounds: growing the vectors record by record was a third of the 32 ms this takes at T = 79)
        const size_t t1 = (size_t)T, t2 = t1 * t1, t3 = t2 * t1;
        const size_t cap[EX_NQ] = {32 * t1, 4 * t2, 2 * t2, 6 * t2 + 64, t3 / 10 + t2 + 64, inv_g > 0 ? t3 / (2 * (size_t)inv_g) + t2 + 64 : 0};
        for (int qi = 0; qi < EX_NQ; ++qi) q[qi].reserve(cap[qi]);
    }
    struct Dep { uint32_t idx, want; };
    auto add = [&](int qi, const double* A, const double* B, double* C, const double* P, int kc_h0, int kc_h1, bool diag, int rmw,
                   std::initializer_list<Dep> deps, uint32_t s0, uint32_t s1, int kc_split = 0, Dep d2a = Dep{EX_NONE, 0},
                   Dep d2b = Dep{EX_NONE, 0}) {
        for (int h = 0; h < 2; ++h) {
            ExTask t{};
            t.A = A; t.B = B + (int64_t)h * CTILE * ld; t.C = C + h * CTILE; t.P = P ? P + h * CTILE : nullptr;
            int nd = 0;
            for (int d = 0; d < EX_NDEP; ++d) { t.dep_idx[d] = EX_NONE; t.dep_want[d] = 0; }
            for (const Dep& d : deps) {
                if (d.idx == EX_NONE || d.want == 0) continue;
                t.dep_idx[nd] = d.idx; t.dep_want[nd] = d.want; ++nd;
            }
            t.sig_idx[0] = s0; t.sig_idx[1] = s1;
            t.kc = h == 0 ? kc_h0 : kc_h1; t.diag_h = diag ? h : -1; t.rmw = rmw; t.prio = qi <= 1 ? 2 : (qi == 2 ? 1 : 0);
            t.kc_split = kc_split;
            t.dep2_idx[0] = d2a.idx; t.dep2_want[0] = d2a.want; t.dep2_idx[1] = d2b.idx; t.dep2_want[1] = d2b.want;
            q[qi].push_back(t);
        }
    };
    const int CPB = TILE / KC;   // contraction chunks per 128-block
    for (int k = 0; k + 3 < T; ++k) {
        // Solve(i, k) = A(i, k) W_kk'  (W_kk lower-triangular: the left half of the columns needs the first 64 contraction indices only)
        const double* Wkk = dW + (int64_t)k * TILE * (ld + 1);
        // The rows k+3 .. k+2+CH_NSF are solved panel by panel inside the chain kernel (solve_follower).  The next NBU rows -- the
        // ones that become follower rows within NBU blocks -- have their row step (Solve, then Late of column k+1) in the URGENT
        // queue: what paces the whole factorisation is the latency pivot k -> inverse -> Solve -> Late of the row that enters the
        // follower window next (chain period ~ 69 us + that latency - ~29 us), and in the ordinary queue that row's two steps wait
        // for a free workgroup twice and run beside bulk work.  Everything further out is queue 1.
        const int NBU = g_chol_exec_nbu, rb = k + 3 + CH_NSF;   // first row solved by the executor
        auto solve_row = [&](int qi, int i) {
            add(qi, Ap(i, k), Wkk, Sp(i, k), nullptr, CPB / 2, CPB, false, 0,
                {{widx(fl.solved + k), 1u}, {k >= 1 ? ver(i, k) : EX_NONE, 16u * (unsigned)(nb(k) + 1)}}, sver(i, k), EX_NONE);
        };
        // Late(k): blocks max(k-1, 0) .. k into the tiles read next
        const int kb0 = std::max(k - 1, 0);
        auto late = [&](int qi, int i, int c, uint32_t sig2) {
            const bool has_p = e_of(i, c) >= ks(c);
            auto chain_row = [&](int kk, int r) { return Dep{widx(fl.xp + ((size_t)kk * T + r) * CH_PANELS + (CH_PANELS - 1)), 1u}; };
            const Dep p_dep{has_p ? pver(i, c) : EX_NONE, 16u}, v_dep{ver(i, c), 16u * (unsigned)nb(c)};
            // (a diagonal tile whose bulk groups already reach block k-1 -- c = k+4 a multiple of 4 -- gets block k only)
            const int kb = (i == c && ks(c) >= k) ? k : kb0, kcl = (k - kb + 1) * CPB;
            if (kb < k) {
                // TWO PIECES: the block k-1 half of the contraction needs nothing of block k, so the task is claimed and starts before
                // S(i, k) exists -- for the follower rows while block k is still being factored, for the others while their row
                // solve is running -- and waits for S(i, k) (and the chain's row c) inside, half a contraction from its end
                const Dep b_prev = c == k + 1 ? chain_row(k - 1, c) : Dep{sver(c, k - 1), 16u};
                add(qi, Sp(i, kb), Sp(c, kb), Ap(i, c), has_p ? Pp(i, c) : nullptr, kcl, kcl, i == c, 1,
                    {{sver(i, k - 1), 16u}, b_prev, p_dep, v_dep}, ver(i, c), sig2, CPB,
                    Dep{sver(i, k), 16u}, c <= k + 2 ? chain_row(k, c) : (c != i ? Dep{sver(c, k), 16u} : Dep{EX_NONE, 0}));
                return;
            }
            // one block, one piece.  rows of S on the B side: row c of block k
            const Dep b0 = c <= k + 2 ? chain_row(k, c) : Dep{sver(c, k), 16u};
            add(qi, Sp(i, kb), Sp(c, kb), Ap(i, c), has_p ? Pp(i, c) : nullptr, kcl, kcl, i == c, 1,
                {{sver(i, k), 16u}, b0, p_dep, v_dep}, ver(i, c), sig2);
        };
        // urgent, in the order they are needed: the tiles the chain's solve followers read in block k+1, then the three tiles of row
        // k+4 that its gated updates finish during block k+1 (pre3; the blocks up to k: one block of slack), then the row steps of
        // the rows about to enter the follower window
        for (int i = k + 4; i < std::min(T, rb); ++i) late(0, i, k + 1, EX_NONE);
        if (k + 4 < T)
            for (int j = 0; j < 3; ++j) late(0, k + 4, k + 2 + j, widx(fl.pre3 + 4 * (k + 1) + j));
        for (int i = rb; i < std::min(T, rb + NBU); ++i) solve_row(0, i);
        for (int i = rb; i < std::min(T, rb + NBU); ++i) late(0, i, k + 1, EX_NONE);
        for (int i = rb + NBU; i < T; ++i) solve_row(1, i);
        for (int i = rb + NBU; i < T; ++i) late(1, i, k + 1, EX_NONE);
    }
    for (int kp = 0; kp + 5 < T; ++kp) {
        // Early(kp): P(i, c) = sum_{b = ks(c)}^{kp} S(i, b) S(c, b)'  for the tiles Late(kp + 2) finishes
        auto early = [&](int i, int c) {
            if (i >= T || c >= T || ks(c) > kp) return;
            const int nbk = kp - ks(c) + 1;   // blocks in the sum: 1 ... 5, growing with c inside a group of four columns
            if (((g_chol_exec_early_split == 1 && (T <= (inv_g == 0 ? 56 : 23) || (inv_g == 0 && T - kp <= g_chol_exec_early_tail))) ||
                 g_chol_exec_early_split == 2) && nbk >= 2) {
                // (round 6) TWO PIECES like Late: the blocks before kp need nothing of block kp, so the task starts a block earlier and waits for
                // S(i, kp) / S(c, kp) inside, one K = 128 piece from its end.  In one piece it started only when the LAST block's rows were solved and
                // then ran its whole window -- up to 50 us on the path S(i, kp) -> Early -> Late -> follower, growing over every group of four columns:
                // the drift behind the owner's 30-50 us waits for crit[k-2] at every fourth block (profiles/r06_exec_trace_N3000.txt).
                // The factorisation ALONE up to 56 row tiles: N=3000 1.14 -> 1.09-1.11 ms, N=3500 1.37 -> 1.31, N=6000 3.03 -> 2.90; N=8000 the same,
                // N=10^4 9.0 -> 9.2 (a waiting task holds a workgroup the bulk work could use); with the inverse queues the same at N <= 3000 and
                // worse beyond (N=4000 2.00 -> 2.05, N=10^4 13.65 -> 14.15): one piece there; up to 23 row tiles two pieces with them too
                // (N=1600 0.64 -> 0.615 ms, N=2000 0.78 -> 0.76, N=2300 0.93 -> 0.90, N=2800 1.147 -> 1.114).
                add(2, Sp(i, ks(c)), Sp(c, ks(c)), const_cast<double*>(Pp(i, c)), nullptr, nbk * CPB, nbk * CPB, i == c, 0,
                    {{sver(i, kp - 1), 16u}, {sver(c, kp - 1), 16u}}, pver(i, c), EX_NONE, (nbk - 1) * CPB, Dep{sver(i, kp), 16u},
                    c != i ? Dep{sver(c, kp), 16u} : Dep{EX_NONE, 0});
                return;
            }
            add(2, Sp(i, ks(c)), Sp(c, ks(c)), const_cast<double*>(Pp(i, c)), nullptr, nbk * CPB,
                nbk * CPB, i == c, 0, {{sver(i, kp), 16u}, {sver(c, kp), 16u}}, pver(i, c), EX_NONE);
        };
        early(kp + 6, kp + 4);   // the three tiles of row kp+6: blocks up to kp (e_of), then Late(kp+2) adds two, the chain the last
        early(kp + 6, kp + 5);
        early(kp + 6, kp + 6);
        for (int i = kp + 6; i < T; ++i) early(i, kp + 3);
    }
    auto add_bulk = [&](int m, int c, int i) {
        add(EX_QBULK, Sp(i, 4 * m), Sp(c, 4 * m), Ap(i, c), nullptr, 4 * CPB, 4 * CPB, i == c, 1,
            {{sver(i, 4 * m + 3), 16u}, {sver(c, 4 * m + 3), 16u}, {ver(i, c), 16u * (unsigned)m}}, ver(i, c), EX_NONE);
    };
    if (!g_chol_exec_bulk_edf) {
        // round 3: group after group, column-major inside a group
        for (int m = 0; 4 * m + 8 <= T - 1; ++m)
            for (int c = 4 * m + 8; c < T; ++c)
                for (int i = c; i < T; ++i) add_bulk(m, c, i);
    } else {
        // Round-4 experiment (BOHIP_CHOL_EXEC_BULK_EDF=1, off by default): the bulk queue in EARLIEST-DEADLINE order.  A workgroup
        // claims from a queue only when its HEAD is runnable, and group after group the head of group m -- the four columns the pivot
        // chain reaches within the next blocks -- sits behind the far columns of group m-1, which nobody needs for another 50 blocks;
        // the round-3 trace at N = 10^4 shows the chain stalling for up to 0.8 ms at group boundaries (gap between pivots: mean 64 us,
        // median 18).  The deadline of a bulk tile is its COLUMN (the chain needs column c complete when it gets there), its release
        // the moment its group's panels are solved.  The order here is produced by simulating the factorisation on the host: a chain
        // that needs `period` per block and waits for its next column, `workers` workgroups that each take the released tile of the
        // smallest column (rounds on one tile in group order), a group released `lag` behind its fourth pivot.  The counters the
        // records wait for are the same, so any order is CORRECT (tests/test_exec_tasks.py replays this list too).
        // MEASURED (profiles/r04_exec_bulk_edf.txt): the chain then advances evenly -- 3.3 blocks per 500 us from start to end, no stall
        // above 100 us -- and the factorisation takes exactly as long: 9.75-9.83 ms against 9.80-9.86 at N = 10^4, 15.3 against 14.9 at
        // N = 12000.  The executor is THROUGHPUT-bound, not order-bound: 350 of its 476 workgroups are busy whatever the order (bulk
        // tasks lose 9 us of look + claim per 61 us of work, the row steps 45 us of waiting per 37 us), and an evenly paced chain only
        // moves the idle time from the group boundaries to every block.
        const int NG = std::max(0, (T - 1 - 8) / 4 + 1);
        const double period = 76.0, lag = 90.0, tile_us = 125.0, col_lag = 140.0;   // us: pivot block; pivot -> S rows solved; one tile (two records); last bulk tile of a column -> its pivot can start
        const int workers = 400;
        struct Tile { int c, m, i; };
        auto later = [](const Tile& a, const Tile& b) { return a.c != b.c ? a.c > b.c : (a.m != b.m ? a.m > b.m : a.i > b.i); };
        std::priority_queue<Tile, std::vector<Tile>, decltype(later)> ready(later);
        std::vector<int> left_in_col(T, 0);                 // bulk tiles of column c not yet finished
        std::vector<double> col_done(T, 0.0), pivot_end(T, 0.0);
        for (int m = 0; m < NG; ++m)
            for (int c = 4 * m + 8; c < T; ++c) left_in_col[c] += T - c;
        // (i, c) -> finish time of its last round, next round to run
        std::vector<int> next_m((size_t)T * T, 0);
        struct Run { double end; Tile t; };
        auto run_later = [](const Run& a, const Run& b) { return a.end > b.end; };
        std::priority_queue<Run, std::vector<Run>, decltype(run_later)> running(run_later);
        int released = 0, pivots = 0;          // groups released, pivot blocks finished
        double now = 0.0;
        size_t emitted = 0, total = 0;
        for (int c = 8; c < T; ++c) total += (size_t)left_in_col[c];
        auto pivot_ready_at = [&](int k) {     // when can pivot block k start?  its column must be complete
            double t = k == 0 ? 0.0 : pivot_end[k - 1];
            if (k >= 8 && left_in_col[k] > 0) return 1e300;
            if (k >= 8) t = std::max(t, col_done[k] + col_lag);
            return t;
        };
        std::vector<char> busy((size_t)T * T, 0);   // a round of this tile is running
        while (emitted < total) {
            // advance the chain as far as it can go at `now`
            while (pivots < T) {
                const double st = pivot_ready_at(pivots);
                if (st + period > now) break;
                pivot_end[pivots] = st + period;
                ++pivots;
            }
            // release the groups whose fourth pivot is `lag` old: the tiles whose earlier rounds are all done become ready now, the
            // others when their running (or still queued) round finishes
            while (released < NG && 4 * released + 3 < pivots && pivot_end[4 * released + 3] + lag <= now) {
                const int m = released++;
                for (int c = 4 * m + 8; c < T; ++c)
                    for (int i = c; i < T; ++i)
                        if (next_m[(size_t)i * T + c] == m && !busy[(size_t)i * T + c]) ready.push(Tile{c, m, i});
            }
            // start work on free workers, smallest column first
            while ((int)running.size() < workers && !ready.empty()) {
                const Tile t = ready.top(); ready.pop();
                add_bulk(t.m, t.c, t.i);
                ++emitted;
                next_m[(size_t)t.i * T + t.c] = t.m + 1;
                busy[(size_t)t.i * T + t.c] = 1;
                running.push(Run{now + tile_us, t});
            }
            // next event: a tile finishes, a pivot ends, a group is released
            double nxt = 1e300;
            if (!running.empty()) nxt = std::min(nxt, running.top().end);
            if (pivots < T) { const double st = pivot_ready_at(pivots); if (st < 1e299) nxt = std::min(nxt, st + period); }
            if (released < NG && 4 * released + 3 < pivots) nxt = std::min(nxt, pivot_end[4 * released + 3] + lag);
            if (nxt >= 1e299) break;       // (cannot happen: something is always in flight until everything is emitted)
            now = std::max(now, nxt);
            while (!running.empty() && running.top().end <= now) {
                const Tile t = running.top().t;
                const double e = running.top().end;
                running.pop();
                busy[(size_t)t.i * T + t.c] = 0;
                if (--left_in_col[t.c] == 0) col_done[t.c] = e;
                // the tile's next round, if its group is out already
                if (t.m + 1 < released && 4 * (t.m + 1) + 8 <= t.c) ready.push(Tile{t.c, t.m + 1, t.i});
            }
        }
        // safety net: whatever the simulation did not emit (it always emits everything; a change of the model must not lose tiles)
        if (emitted < total)
            for (int m = 0; m < NG; ++m)
                for (int c = 4 * m + 8; c < T; ++c)
                    for (int i = c; i < T; ++i)
                        if (next_m[(size_t)i * T + c] <= m) { add_bulk(m, c, i); next_m[(size_t)i * T + c] = m + 1; }
    }
    // Queues 3 (EX_QROWS) and 5 (EX_QWAVE): W = L^-1 behind the chain (inv_g > 0: blocks per piece of the long contraction = chunk size G).
    //   Z(i, j) = -sum_{k=j}^{i-1} L(i, k) W(k, j)   accumulated in place at W(i, j), in k order, from three kinds of pieces:
    //       wave m     chunk m = blocks [G m, G (m+1)), pushed to EVERY row i >= G (m+1) + 1 as soon as the chunk's rows of W are
    //                  final (right-looking: queue 5, lowest priority -- thousands of independent K = 128 G tasks per wave)
    //       partial    what is left of row i's own chunk, [G m_i, i-1), m_i = (i-1) / G
    //       last       {i-1}: the only piece that needs row i-1 of W; it also stores Z' to the mirror tile (j, i) of W
    //   W(i, j) = W_ii Z(i, j)                        A = W_ii, B = Z' (K-major), result to W(i, j) and transposed to W'(j, i)
    // partial, last and the product form the row-to-row chain (queue 3, ABOVE the bulk updates: per row one K = 128 task + the product,
    // little work but serial -- starved behind the bulk it started when the bulk ended and the waves with it;
    // in ONE queue with the waves every row waited behind a whole wave).  Counters, 4 words per tile behind the queue cursors:
    // [0] zver = rounds on Z (16 each), [1] zt = Z' complete, [2] wfin.  Everything a record waits for is earlier in its queue, in
    // another queue or the chain's; nothing outside these two queues ever waits for them.
    // GROUP FORM of the two inverse queues (T >= g_chol_inv_grp_min).  Above, every row of W waits for the row before it: last(i) needs
    // W(i-1, j), product(i) needs last(i) -- two dependent K = 128 records per row, ~117 us under load, T - 1 rows one after the other.  At
    // N = 10^4 that chain IS the second half of the refit: starved behind the bulk until ~7 ms, then 67 rows x 117 us = 7.8 ms during which
    // the waves fill in but the end is the chain's (profiles/r04_exec_trace_N10000.txt).  Here the rows of a GROUP of G blocks do not wait
    // for each other.  With g0 = G g the first block of group g and D_g = L_gg^-1 the group's own triangular block of W:
    //   D_g                      by the row chain above, restricted to the columns j >= g0 (queue 3: at most G - 1 short row steps, right
    //                            behind the pivots, nothing else waits inside them)
    //   P(i, j) = -sum_{k=j}^{g0-1} L(i, k) W(k, j),  j < g0      the waves as above, one round per earlier group m (blocks of group m, pushed
    //                            to every later row as soon as group m's rows of W are final); the round of group g-1 completes P and
    //                            also stores P' to the mirror tile (j, i)
    //   W(i, j) = sum_{k=g0}^{i} W(i, k) P(k, j)                  ONE record per tile half, K = 128 (i - g0 + 1): A = row i of D_g (K-major in W),
    //                            B = the mirror tiles (j, g0 .. i)
    // so the serial part is one K = 128 G round + one product per GROUP (~250 us per G rows instead of ~117 us per row), and what runs last
    // is a full-width product, not a row chain.  Queue 5 order: rounds of group m for the rows of group m+1 first, then the first half of the
    // far rows, then the products of group m+1 (their inputs were claimed a thousand records earlier), then the other half (which separates
    // the products from the rounds of group m+1 that need them).  Counters: word 3 of tile (g0, j) counts the completed P' tiles of group g
    // and column j, word 2 of the same tile its finished products, word 3 of the diagonal tile (g0, g0) the finished tiles of D_g.
    const bool inv_groups = inv_g >= 2 && T >= inv_grp_min;   // (a group of one row has no D_g: nothing would mark W as grown)
    if (inv_groups) {
        const int G = inv_g, NG = (T + G - 1) / G;
        const uint32_t ivb = (uint32_t)chol_inv_word(T);
        auto iw = [&](int i, int j, int w) { return ivb + (uint32_t)(((size_t)i * T + j) * 4 + w); };
        auto chain_row = [&](int kk, int r) { return Dep{widx(fl.xp + ((size_t)kk * T + r) * CH_PANELS + (CH_PANELS - 1)), 1u}; };
        auto l_final = [&](int i, int k) { return i <= k + 2 ? chain_row(k, i) : Dep{sver(i, k), 16u}; };
        auto w_final = [&](int k, int j) { return k > j ? Dep{iw(k, j, 2), 16u} : Dep{widx(fl.solved + j), 1u}; };   // inside a group
        auto add_inv = [&](int qi, const double* A, const double* B, double* C, double* CTb, int kc, int mode, std::initializer_list<Dep> deps,
                           uint32_t s0, uint32_t s1) {
            for (int h = 0; h < 2; ++h) {
                ExTask t{};
                t.A = A; t.B = B + (int64_t)h * CTILE * ld; t.C = C + h * CTILE; t.P = CTb ? CTb + (int64_t)h * CTILE * ld : nullptr;
                int nd = 0;
                for (int d = 0; d < EX_NDEP; ++d) { t.dep_idx[d] = EX_NONE; t.dep_want[d] = 0; }
                for (const Dep& d : deps) {
                    if (d.idx == EX_NONE || d.want == 0) continue;
                    t.dep_idx[nd] = d.idx; t.dep_want[nd] = d.want; ++nd;
                }
                t.sig_idx[0] = s0; t.sig_idx[1] = s1;
                t.kc = kc; t.diag_h = -1; t.rmw = mode | (CTb ? 4 : 0); t.prio = 0; t.kc_split = 0;
                t.dep2_idx[0] = t.dep2_idx[1] = EX_NONE;
                q[qi].push_back(t);
            }
        };
        auto Wp = [&](int i, int j) { return dW + (int64_t)i * TILE * ld + (int64_t)j * TILE; };
        auto WTp = [&](int j, int i) { return dWT + (int64_t)j * TILE * ld + (int64_t)i * TILE; };
        auto rows_of = [&](int g) { return std::min(G, T - G * g); };
        auto tiles_of = [&](int g) { const int r = rows_of(g); return r * (r - 1) / 2; };
        auto GP = [&](int g, int j) { return iw(G * g, j, 3); };
        auto GW = [&](int g, int j) { return iw(G * g, j, 2); };
        auto GD = [&](int g) { return iw(G * g, G * g, 3); };
        // D_g: row i's tiles j = g0 .. i-1 -- what is below row i-1 in one piece (needs row i-2), then {i-1}, then the product.
        // Z(i, j) is summed in the scratch matrix's mirror tile (free since Late consumed its Early sum, long before row i was solved), NOT
        // in place: the products below read the tiles of D_g through LDS-DMA, and a tile that was written THREE times by workgroups on
        // different XCDs (Z, Z, then W) is served stale from the reader's L2 every now and then -- nothing else in these queues reads a
        // location that is written more than once (first version: 1 % errors in W at N = 3000 whenever eight refits shared the device).
        auto Zp = [&](int i, int j) { return const_cast<double*>(Pp(i, j)); };
        auto d_partial = [&](int i) {
            if (i >= T) return;
            const int g0 = G * (i / G);
            for (int j = g0; j < i - 1; ++j)
                add_inv(EX_QROWS, Sp(i, j), WTp(j, j), Zp(i, j), nullptr, (i - 1 - j) * CPB, 2, {l_final(i, i - 2), w_final(i - 2, j)}, iw(i, j, 0), EX_NONE);
        };
        auto d_last = [&](int i) {
            const int g0 = G * (i / G);
            for (int j = g0; j < i; ++j) {
                const bool hp = j < i - 1;
                add_inv(EX_QROWS, Sp(i, i - 1), WTp(j, i - 1), Zp(i, j), Wp(j, i), CPB, hp ? 1 : 2,
                        {l_final(i, i - 1), w_final(i - 1, j), Dep{hp ? iw(i, j, 0) : EX_NONE, 16u}}, iw(i, j, 0), iw(i, j, 1));
            }
        };
        auto d_product = [&](int i) {
            const int g = i / G, g0 = G * g;
            for (int j = g0; j < i; ++j)
                add_inv(EX_QROWS, Wp(i, i), Wp(j, i), Wp(i, j), WTp(j, i), CPB, 0, {Dep{iw(i, j, 1), 16u}, Dep{widx(fl.solved + i), 1u}}, iw(i, j, 2), GD(g));
        };
        for (int i = 1; i < T; ++i) {
            if ((i + 1) / G == i / G) d_partial(i + 1);
            if (i % G != 0) { d_last(i); d_product(i); }
        }
        // one round: the blocks of group m into P(i, j)
        auto round = [&](int m, int i, int j) {
            const int g = i / G, m0 = G * m, m1 = G * (m + 1), a = std::max(j, m0), nbf = j >= m0 ? 0 : m - j / G;
            const bool fin = m == g - 1;
            const Dep src = j < m0 ? Dep{GW(m, j), 16u * (unsigned)rows_of(m)} : Dep{GD(m), 16u * (unsigned)tiles_of(m)};
            add_inv(EX_QWAVE, Sp(i, a), WTp(j, a), Wp(i, j), fin ? Wp(j, i) : nullptr, (m1 - a) * CPB, nbf == 0 ? 2 : 1,
                    {l_final(i, m1 - 1), src, Dep{widx(fl.solved + m1 - 1), 1u}, Dep{nbf ? iw(i, j, 0) : EX_NONE, 16u * (unsigned)nbf}},
                    iw(i, j, 0), fin ? GP(g, j) : EX_NONE);
        };
        auto products = [&](int g) {
            const int g0 = G * g, r = rows_of(g);
            for (int i = g0; i < g0 + r; ++i)
                for (int j = 0; j < g0; ++j)
                    add_inv(EX_QWAVE, Wp(i, g0), Wp(j, g0), Wp(i, j), WTp(j, i), (i - g0 + 1) * CPB, 0,
                            {Dep{GP(g, j), 16u * (unsigned)r}, Dep{GD(g), 16u * (unsigned)tiles_of(g)}, Dep{widx(fl.solved + g0 + r - 1), 1u}}, GW(g, j), EX_NONE);
        };
        for (int m = 0; m + 1 < NG; ++m) {
            const int m1 = G * (m + 1), near_end = std::min(T, m1 + G), far_mid = near_end + (T - near_end + 1) / 2;
            for (int i = m1; i < near_end; ++i)
                for (int j = 0; j < m1; ++j) round(m, i, j);
            for (int i = near_end; i < far_mid; ++i)
                for (int j = 0; j < m1; ++j) round(m, i, j);
            products(m + 1);
            for (int i = far_mid; i < T; ++i)
                for (int j = 0; j < m1; ++j) round(m, i, j);
        }
    } else if (inv_g > 0) {
        const int G = inv_g;
        const uint32_t ivb = (uint32_t)chol_inv_word(T);
        auto iw = [&](int i, int j, int w) { return ivb + (uint32_t)(((size_t)i * T + j) * 4 + w); };
        auto chain_row = [&](int kk, int r) { return Dep{widx(fl.xp + ((size_t)kk * T + r) * CH_PANELS + (CH_PANELS - 1)), 1u}; };
        auto l_final = [&](int i, int k) {   // row i of L complete up to block k (rows are solved in block order)
            return i <= k + 2 ? chain_row(k, i) : Dep{sver(i, k), 16u};
        };
        auto w_final = [&](int k, int j) { return k > j ? Dep{iw(k, j, 2), 16u} : Dep{widx(fl.solved + j), 1u}; };   // W(k, j); the diagonal tile is the inverter's
        auto add_inv = [&](int qi, const double* A, const double* B, double* C, double* CTb, int kc, int mode, std::initializer_list<Dep> deps,
                           uint32_t s0, uint32_t s1) {
            for (int h = 0; h < 2; ++h) {
                ExTask t{};
                t.A = A; t.B = B + (int64_t)h * CTILE * ld; t.C = C + h * CTILE; t.P = CTb ? CTb + (int64_t)h * CTILE * ld : nullptr;
                int nd = 0;
                for (int d = 0; d < EX_NDEP; ++d) { t.dep_idx[d] = EX_NONE; t.dep_want[d] = 0; }
                for (const Dep& d : deps) {
                    if (d.idx == EX_NONE || d.want == 0) continue;
                    t.dep_idx[nd] = d.idx; t.dep_want[nd] = d.want; ++nd;
                }
                t.sig_idx[0] = s0; t.sig_idx[1] = s1;
                t.kc = kc; t.diag_h = -1; t.rmw = mode | (CTb ? 4 : 0); t.prio = 0; t.kc_split = 0;
                t.dep2_idx[0] = t.dep2_idx[1] = EX_NONE;
                q[qi].push_back(t);
            }
        };
        auto Wp = [&](int i, int j) { return dW + (int64_t)i * TILE * ld + (int64_t)j * TILE; };
        auto WTp = [&](int j, int i) { return dWT + (int64_t)j * TILE * ld + (int64_t)i * TILE; };
        auto m_of = [&](int i) { return (i - 1) / G; };                                       // row i's own (incomplete) chunk
        auto n_waves = [&](int i, int j) { return std::max(0, m_of(i) - j / G); };           // wave pieces tile (i, j) receives
        // what is left of row i's own chunk before its last block goes in two pieces, so that none of them has less than two rows
        // of slack behind the product it waits for: A = [G m_i, i-2) (needs row i-3 of W), B = {i-2} (row i-2).  (One piece
        // [G m_i, i-1): up to G-1 blocks of contraction, ~100 us at G = 8, between the products of two consecutive rows -- the
        // rows fell behind the pivots by that much every G rows.)
        auto has_a = [&](int i, int j) { return std::max(j, G * m_of(i)) < i - 2; };
        auto has_b = [&](int i, int j) { return i - 2 >= G * m_of(i) && j <= i - 2; };
        auto zdep = [&](int i, int j, int n) { return Dep{n ? iw(i, j, 0) : EX_NONE, 16u * (unsigned)n}; };
        auto wave = [&](int m) {   // after the product of row G (m+1) - 1
            const int k1 = G * (m + 1);
            for (int i = k1 + 1; i < T; ++i)
                for (int j = 0; j < k1; ++j) {
                    const int a = std::max(j, G * m), nbf = j >= G * m ? 0 : m - j / G;
                    add_inv(EX_QWAVE, Sp(i, a), WTp(j, a), Wp(i, j), nullptr, (k1 - a) * CPB, nbf == 0 ? 2 : 1,
                            {l_final(i, k1 - 1), w_final(k1 - 1, j), zdep(i, j, nbf)}, iw(i, j, 0), EX_NONE);
                }
        };
        auto partial_a = [&](int i) {
            if (i >= T || i < 3) return;
            for (int j = 0; j < i - 2; ++j) {
                if (!has_a(i, j)) continue;
                const int a = std::max(j, G * m_of(i)), nbf = n_waves(i, j);
                add_inv(EX_QROWS, Sp(i, a), WTp(j, a), Wp(i, j), nullptr, (i - 2 - a) * CPB, nbf == 0 ? 2 : 1,
                        {l_final(i, i - 3), w_final(i - 3, j), zdep(i, j, nbf)}, iw(i, j, 0), EX_NONE);
            }
        };
        auto partial_b = [&](int i) {
            if (i >= T || i < 2) return;
            for (int j = 0; j <= i - 2; ++j) {
                if (!has_b(i, j)) continue;
                const int nbf = n_waves(i, j) + (has_a(i, j) ? 1 : 0);
                add_inv(EX_QROWS, Sp(i, i - 2), WTp(j, i - 2), Wp(i, j), nullptr, CPB, nbf == 0 ? 2 : 1,
                        {l_final(i, i - 2), w_final(i - 2, j), zdep(i, j, nbf)}, iw(i, j, 0), EX_NONE);
            }
        };
        auto last = [&](int i) {
            for (int j = 0; j < i; ++j) {
                const int nbf = n_waves(i, j) + (has_a(i, j) ? 1 : 0) + (has_b(i, j) ? 1 : 0);
                add_inv(EX_QROWS, Sp(i, i - 1), WTp(j, i - 1), Wp(i, j), Wp(j, i), CPB, nbf == 0 ? 2 : 1,
                        {l_final(i, i - 1), w_final(i - 1, j), zdep(i, j, nbf)}, iw(i, j, 0), iw(i, j, 1));
            }
        };
        auto product = [&](int i) {
            for (int j = 0; j < i; ++j)
                add_inv(EX_QROWS, Wp(i, i), Wp(j, i), Wp(i, j), WTp(j, i), CPB, 0, {Dep{iw(i, j, 1), 16u}, Dep{widx(fl.solved + i), 1u}}, iw(i, j, 2), EX_NONE);
        };
        for (int i = 1; i < T; ++i) {   // what follows needs the product of row i-1 at most, except product(i): pivot i
            partial_a(i + 2);
            partial_b(i + 1);
            last(i);
            product(i);
            if ((i + 1) % G == 0) wave((i + 1) / G - 1);
        }
    }
    all.clear();
    {
        size_t total = 0;
        for (int qi = 0; qi < EX_NQ; ++qi) total += q[qi].size();
        all.reserve(total);
    }
    qbeg[0] = 0;
    for (int qi = 0; qi < EX_NQ; ++qi) {
        all.insert(all.end(), q[qi].begin(), q[qi].end());
        qbeg[qi + 1] = (int)all.size();
    }
}
// May the bulk queue be claimed `stride` records at a time?  A claim starts when ALL its records' counters are in and runs them back to back,
// and claims are the consecutive blocks of `stride` records from the queue's start (one fetch-and-add each) -- so no record of a block may wait
// for another record of the SAME block: that claim would wait for itself.  In the bulk queue that is two rounds on one tile next to each other,
// which happens where a group has a single tile left (T = 4 m + 9: the last round of the last diagonal tile directly behind the round before).
static bool exec_bulk_stride_ok(const std::vector<ExTask>& all, const int* qbeg, int stride) {
    for (int b = qbeg[EX_QBULK]; b < qbeg[EX_QBULK + 1]; b += stride) {
        const int e = std::min(b + stride, qbeg[EX_QBULK + 1]);
        for (int x = b; x < e; ++x)
            for (int y = x + 1; y < e; ++y)
                if (all[x].C == all[y].C) return false;
    }
    return true;
}
static int build_exec_tasks(bohip_gp* g, int T) {
    if (g->ex_T == T && g->dex_tasks && g->ex_nsf == g_chol_nsf && g->ex_inv_g == g_chol_inv_g && g->ex_grp_min == g_chol_inv_grp_min) return 0;
    std::vector<ExTask> all;
    exec_task_list(g->dL, g->dS, g->dW, g->dWT, g->dchol_flags, g->ld, T, g_chol_nsf, g_chol_inv_g, all, g->ex_qbeg);
    if (all.size() > g->ex_cap) {
        if (g->dex_tasks) hipFree(g->dex_tasks);
        g->dex_tasks = nullptr; g->ex_cap = 0;
        HIPCHK(hipMalloc(&g->dex_tasks, all.size() * sizeof(ExTask)));
        g->ex_cap = all.size();
    }
    HIPCHK(hipMemcpyAsync(g->dex_tasks, all.data(), all.size() * sizeof(ExTask), hipMemcpyHostToDevice, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));   // `all` is pageable host memory that dies with this frame
    g->ex_bulk4_ok = exec_bulk_stride_ok(all, g->ex_qbeg, 4);
    g->ex_T = T;
    g->ex_nsf = g_chol_nsf;
    g->ex_inv_g = g_chol_inv_g;
    g->ex_grp_min = g_chol_inv_grp_min;
    return 0;
}

static std::mutex g_df_mutex[64];   // per device: see refit_once
// ... and the same between PROCESSES that share a device (two ranks on one GPU in the sharded bench's test mode, several BO loops per GPU): an
// advisory file lock named after the device's PCI address, taken inside the host lock and released with it (the kernel drops it if the process
// dies).  Round 5 (advisor): the lock file lives in a PER-USER directory ($XDG_RUNTIME_DIR, else /tmp/bohip-<uid> created 0700 and checked to be
// ours), is opened O_NOFOLLOW | O_CLOEXEC with mode 0600 and never chmod'ed (a planted symlink or file cannot redirect it); the lock is taken
// NON-blocking with a bounded retry (BOHIP_DF_FILE_LOCK_MS, 300 ms) -- a stopped or hung holder no longer blocks every refit on the GPU: the
// refit then goes ahead unlocked and the flags' 200 ms time-out + launch-chain fall-back stay the safety net; the descriptor is re-opened in a
// forked child (parent and child would share ONE open file description and not exclude each other).  No lock file (read-only directory,
// BOHIP_DF_FILE_LOCK=0): the refit goes ahead as before.  Processes of DIFFERENT users on one device do not see each other's lock.
struct DfFileLock {
    int fd = -1;
    static int open_lock_file(int device) {
        char bus[64] = "dev";
        if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) snprintf(bus, sizeof bus, "dev%d", device);
        for (char* c = bus; *c; ++c) if (*c == ':' || *c == '.' || *c == '/') *c = '_';
        char dir[128];
        const char* xdg = getenv("XDG_RUNTIME_DIR");
        struct stat st;
        if (xdg && xdg[0] == '/' && strlen(xdg) < 100 && stat(xdg, &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == geteuid()) {
            snprintf(dir, sizeof dir, "%s", xdg);
        } else {
            snprintf(dir, sizeof dir, "/tmp/bohip-%u", (unsigned)geteuid());
            if (mkdir(dir, 0700) != 0 && errno != EEXIST) return -1;
            if (lstat(dir, &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & 077)) return -1;   // not ours / a link / open to others
        }
        char path[256];
        snprintf(path, sizeof path, "%s/bohip_refit_%s.lock", dir, bus);
        return open(path, O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0600);
    }
    bool contended = false;   // another process held the lock for the whole wait: this refit takes the launch-chained form (nothing in it spins)
    DfFileLock() = default;
    // T: row tiles of the refit -- the wait covers a dozen refits of this size by the other process (300 ms up to ~50 row tiles, T^2 / 8 ms beyond)
    void acquire(int device, int T) {
        // BOHIP_DF_FILE_LOCK: 0 = off, 1 = on with the default wait, n > 1 = on, waiting up to n ms for the other process
        static int setting = [] { const char* e = getenv("BOHIP_DF_FILE_LOCK"); return e ? std::max(0, atoi(e)) : 1; }();
        const int enabled = setting != 0, wait_ms = setting > 1 ? setting : std::max(300, T * T / 8);
        if (!enabled) return;
        static int fds[64];
        static pid_t owner[64];
        static std::mutex open_mutex;
        const int dv = device & 63;
        {
            std::lock_guard<std::mutex> lk(open_mutex);
            const pid_t me = getpid();
            if (owner[dv] != me) {                        // first use in this process, or we are a forked child: an own open file description
                if (owner[dv] != 0 && fds[dv] >= 0) close(fds[dv]);
                fds[dv] = open_lock_file(device);
                owner[dv] = me;
            }
            fd = fds[dv];
        }
        if (fd < 0) return;                               // no lock file (directory not ours, ...): unlocked, the time-out fall-back is the safety net
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            if (flock(fd, LOCK_EX | LOCK_NB) == 0) return;
            if (errno != EWOULDBLOCK && errno != EINTR) break;
            if (std::chrono::steady_clock::now() - t0 >= std::chrono::milliseconds(wait_ms)) { contended = true; break; }
            usleep(200);
        }
        fd = -1;
    }
    void release() { if (fd >= 0) { flock(fd, LOCK_UN); fd = -1; } }
    ~DfFileLock() { release(); }
};
static int device_cus();
static int cholesky_exec(bohip_gp* g, int T) {
    const int64_t ld = g->ld;
    CHK(build_exec_tasks(g, T));
    CholFlags fl = chol_flags_layout(g, T);
    fl.mode2 = 200;   // the chain's view: row k+3's three tiles carry block k (its own gated updates: rest[k] = 24), inverter workgroup on
    HIPCHK(hipMemsetAsync(g->dchol_flags, 0, chol_flag_words(T) * sizeof(unsigned), g->stream));
    HIPCHK(hipEventRecord(g->ev_panels, g->stream));           // K and the cleared flags are in place
    fl.nsf = g_chol_nsf;
    hipLaunchKernelGGL(k_chol_chain, dim3(9 + g_chol_nsf + 6), dim3(CH_THREADS), CH_LDS_BYTES, g->stream, g->dL, ld, g->dS, T, fl, g->dinfo, g->dW, g->dWT);
    HIPCHK(hipGetLastError());
    // The solved panels go home (S -> L) right behind the CHAIN kernel, not behind the executor: every S(i, k) feeds the diagonal tile of its
    // row, so all of them are final -- and the tiles of L they replace have been read for the last time -- when the last pivot is done, while the
    // executor still has the second half of W = L^-1 to grow (N = 10^4: 4.4 ms).  The copy runs beside that on the CUs the chain has left.
    if (g_chol_copy_early) {
        hipLaunchKernelGGL(k_copy_offdiag_tiles, dim3(T * (T - 1) / 2), dim3(256), 0, g->stream, g->dS, g->dL, ld, T);
        HIPCHK(hipGetLastError());
    }
    if (T > 3 || g->ex_qbeg[EX_NQ] > 0) {   // (T = 2, 3: no factorisation task, but the inverse's rows)
        ExQueues q{};
        q.tasks = g->dex_tasks;
        for (int i = 0; i <= EX_NQ; ++i) q.qbeg[i] = g->ex_qbeg[i];
        q.flags = g->dchol_flags;
        q.abort = g->dchol_flags + chol_abort_word(T);
        q.heads = q.abort + 1;
        q.ld = ld;
        q.spin_ticks = g_chol_spin_ticks;
        q.fill = g_chol_exec_fill;
        // executor workgroups: two per CU that the chain kernel leaves free (a small partition must not fill its slots with the
        // urgent queue's own workgroups: nobody would serve the other queues until the time-out)
        const int cus_free = std::max(1, device_cus() - (9 + g_chol_nsf + 6));
        // How many: two per CU share the CU's matrix pipe -- more throughput (N = 10^4: 14.5 against 15.6 ms for the bare task list), but every
        // task takes ~1.6x as long beside a busy neighbour, and at chain-paced sizes that is what the pivot chain waits for (the row steps
        // Solve -> Late of the rows about to enter its window): with ONE workgroup per CU the refit at N = 3000 takes 1.43 instead of 1.61 ms,
        // N = 4000 2.13 instead of 2.38; from N = 6000 on two per CU win (4.55 against 5.2).  Swept at T = 8 ... 47
        // (profiles/r04_exec_workgroups_by_size.txt): 1 per CU up to 32 row tiles, 2 from 45 on, linear in between.
        const double per_cu = T <= 32 ? 1.0 : (T >= 45 ? 2.0 : 1.0 + (T - 32) / 13.0);
        const int exec_wgs = std::max(2, g_chol_exec_wgs > 0 ? std::min(g_chol_exec_wgs, 2 * cus_free) : (int)(per_cu * cus_free + 0.5));
        q.nurgent = std::max(1, std::min(exec_wgs / 8, g_chol_exec_urgent > 0 ? g_chol_exec_urgent : (T >= 56 ? 16 : 32)));   // queue 0 is served by these only: never zero
        q.second_from = g_chol_exec_second && exec_wgs > cus_free ? cus_free : 0;
        // (idling the 32 second workgroups that are dispatched as the urgent workgroups' CU mates was measured: the lost throughput costs more,
        // N = 8000 8.85 against 8.60 ms, N = 10^4 alone 9.4 against 9.1)
        // (with one workgroup per CU nobody slows a neighbour down and the reserve buys nothing: 1.43 ms at N = 3000 with 0, 30 or 59 of them)
        // (36 more workgroups than fit beside the chain -- they start on its 18 CUs when it has ended, at N = 10^4 with 6 ms still to go --
        // were measured: 15.4-15.6 ms either way.  What ran last then was the inverse's row chain, not a lack of workgroups.)
        q.nfast = std::max(0, std::min(exec_wgs / 4, g_chol_exec_fast >= 0 ? g_chol_exec_fast : (T <= 48 && per_cu > 1.0 ? (int)(112 * (per_cu - 1.0)) : 0)));
        int pairs_ = g_chol_exec_pairs >= 0 ? g_chol_exec_pairs : (T >= 72 ? 2 : 1);
        if (pairs_ >= 2 && !(pairs_ == 2 && g->ex_bulk4_ok)) pairs_ = 1;   // (more than two tiles per claim: never checked, never faster)
        q.stride[0] = 1; q.stride[1] = 1; q.stride[2] = pairs_ ? 2 : 1; q.stride[EX_QBULK] = pairs_ >= 2 ? 2 * std::min(pairs_, 4) : (pairs_ ? 2 : 1);
        q.stride[EX_QROWS] = q.stride[EX_QWAVE] = g_chol_exec_inv_pairs ? 2 : 1;
        q.fill_inv = g_chol_exec_fill_inv;
        q.patience_ticks = (unsigned)g_chol_exec_patience_us * 100u;
        HIPCHK(hipStreamWaitEvent(g->col_stream, g->ev_panels, 0));
        // (Holding this launch back until the chain's workgroups are resident -- hipStreamWaitValue32 on a word they count themselves into:
        // the kernel then starts ~10 us behind them -- was built against the time-outs of refits that share the device with other host
        // threads' work: it cost 0.1 ms at N = 10^4 and the time-outs stayed.  What causes those is streams of several handles sharing a
        // hardware queue, where a kernel waits for the END of the one before it; the per-device lock in refit_once removed most of them.)
        const size_t exec_lds = per_cu <= 1.0 && g_chol_exec_excl ? std::max<size_t>(glds3_lds_bytes<4>(), 84 * 1024) : glds3_lds_bytes<4>();
        hipLaunchKernelGGL(k_chol_exec, dim3(exec_wgs), dim3(GEMM_THREADS_8), exec_lds, g->col_stream, q);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(g->ev_inv, g->col_stream));
        HIPCHK(hipStreamWaitEvent(g->stream, g->ev_inv, 0));
    }
    if (!g_chol_copy_early) {
        hipLaunchKernelGGL(k_copy_offdiag_tiles, dim3(T * (T - 1) / 2), dim3(256), 0, g->stream, g->dS, g->dL, ld, T);
        HIPCHK(hipGetLastError());
    }
    g->w_seeded = true;
    g->w_done = g->ex_qbeg[EX_QROWS + 1] > g->ex_qbeg[EX_QROWS];
    return 0;
}

// multiprocessors of the current device (the dataflow forms need their persistent workgroups resident at the same time)
static int device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    int& c = cus[dev & 63];
    if (c == 0 && hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) c = 0;
    return c;
}
static int refit_once(bohip_gp* g, double jitter);
// Full rebuild.  A factorisation that fails (BOHIP_E_NOTPD) is reported as such unless jitter escalation was asked for
// (bohip_gp_set_jitter / BOHIP_JITTER): then the diagonal gets jitter_rel x mean(diag) more, x10 per further try -- the shape
// of GaussianProcesses.jl's make_posdef! (UPSTREAM-UNVERIFIED, hence off by default; oracle twin: oracle.py fit_with_jitter).
static int refit(bohip_gp* g) {
    g->jitter_steps_last = 0;
    g->jitter_last = 0.0;
    int rc = refit_once(g, 0.0);
    if (rc != BOHIP_E_NOTPD || g->jitter_tries <= 0 || !(g->jitter_rel > 0.0)) return rc;
    const double mean_diag = std::exp(2.0 * g->logsig) + std::exp(2.0 * g->lognoise);   // stationary kernels: every diagonal entry
    double jit = g->jitter_rel * mean_diag;
    for (int t = 1; t <= g->jitter_tries; ++t, jit *= 10.0) {
        rc = refit_once(g, jit);
        if (rc != BOHIP_E_NOTPD) {
            if (rc == 0) { g->jitter_steps_last = t; g->jitter_last = jit; }
            return rc;
        }
    }
    return rc;
}
static int refit_once(bohip_gp* g, double jitter) {
    CHK(one_time_kernel_setup());
    const int64_t N = g->n;
    if (N == 0) {
        g->stale = false;
        g->n_factored = 0;
        return 0;
    }
    const int64_t Npad = round_up(N + 1, TILE), ld = g->ld;
    const int T = (int)(Npad / TILE);
    const KernelHyper hp = make_hyper(g);
    const double noise = std::exp(2.0 * g->lognoise) + std::numeric_limits<double>::epsilon() + jitter;
    if (g->mirror_dirty) {   // a staged append failed after the mirrors advanced: the host copies are the truth
        HIPCHK(hipMemcpyAsync(g->dX, g->hX.data(), (size_t)N * g->d * 8, hipMemcpyHostToDevice, g->stream));
        HIPCHK(hipMemcpyAsync(g->dy, g->hy.data(), (size_t)N * 8, hipMemcpyHostToDevice, g->stream));
        HIPCHK(hipStreamSynchronize(g->stream));
        g->mirror_dirty = false;
    }
    HIPCHK(hipMemsetAsync(g->dinfo, 0, sizeof(int), g->stream));
    t_begin(g, "build_cov");
    {
        const int rpb = 32;
        dim3 grid((Npad + 255) / 256, (Npad + rpb - 1) / rpb);
#define BC(DTV) hipLaunchKernelGGL(k_build_cov<DTV>, grid, dim3(256), 0, g->stream, g->dX, N, Npad, hp, noise, g->dL, ld, rpb)
        if (g->d <= 2) BC(2); else if (g->d <= 4) BC(4); else if (g->d <= 8) BC(8); else if (g->d <= 16) BC(16);
        else if (g->d <= 32) BC(32); else BC(64);
#undef BC
    }
    HIPCHK(hipGetLastError());
    t_end(g);
    // Which form.  The dataflow forms make progress only while their persistent workgroups are resident TOGETHER, one per CU
    // (the chain's fill the LDS): the chain's 8-9 (+ solve followers), form 1's T - 3 row followers and 2 (T - 3) column
    // updaters.  On a device (or partition: CPX mode exposes 32 CUs) that cannot hold them the launch chain is used.
    const int cus = device_cus();
    const bool df_size = (g_chol_df == 1 && T >= 2 && T <= g_chol_df_tmax) || (g_chol_df == 2 && T >= 2 && T <= CHOL_DF_TCAP);
    bool paused = false;
    if (df_size)   // the pause after a time-out counts refits that WOULD have used a dataflow form, nothing else
        for (int sk = g_chol_df_skip.load(std::memory_order_relaxed); sk > 0;)
            if (g_chol_df_skip.compare_exchange_weak(sk, sk - 1, std::memory_order_relaxed)) { paused = true; break; }
    const bool want_df = !paused && df_size;
    // (from TWO row tiles on with the inverse queues: at T = 2, 3 the executor has no factorisation task at all -- every tile is inside the
    // chain kernel's window -- but it grows W = L^-1 behind the chain: N = 200 0.15 instead of 0.21 ms, N = 300 0.20 instead of 0.28)
    const int exec_min = g_chol_exec_min >= 0 ? g_chol_exec_min : (g_chol_inv_g > 0 ? 2 : 24);   // (alone: 32 until round 6 -- since the chain's blocks run on the matrix pipe the first form's followers are what lags from 24 row tiles on: N=3000 1.15 against 1.18 ms, N=3500 1.37 / 1.51; N=2500 0.935 / 0.908)
    const bool exec_ok = g_chol_exec && T >= std::max(g_chol_inv_g > 0 ? 2 : 4, exec_min) && cus >= 9 + g_chol_nsf + 6 + 8;
    const bool form2_ok = T >= g_chol_df2_min && cus >= 9 + 32;
    const bool form1_ok = T >= 3 && cus >= 8 + 3 * std::max(0, T - 3) + 8;
    // (stage name: with the executor's inverse queues the factorisation and W = L^-1 are ONE stage)
    // One dataflow refit per device at a time (see below): the host lock of this process and the file lock between processes, taken before
    // the stage begins.  Round 6: when ANOTHER PROCESS keeps the file lock for the whole wait this refit takes the launch-chained form instead
    // of going ahead unlocked (two processes' persistent kernels on one chip are what the 200 ms time-outs of profiles/r06_soak.txt are made of).
    std::unique_lock<std::mutex> df_lock;
    DfFileLock df_file;
    bool df_go = want_df && (exec_ok || form2_ok || form1_ok);
    if (df_go) {
        df_lock = std::unique_lock<std::mutex>(g_df_mutex[g->device & 63]);
        df_file.acquire(g->device, T);
        if (df_file.contended) {
            df_go = false;
            df_lock.unlock();
            g->chol_lock_skips++;
        }
    }
    t_begin(g, df_go && exec_ok && g_chol_inv_g > 0 ? "cholesky+inverse" : "cholesky");
    if (df_go) {
        // One dataflow refit per device at a time: its persistent workgroups must be resident together, and two refits from two host threads
        // (several models on one GPU, the logical shards of bohip_mgp_*) take each other's CUs -- every other one then sat out its 200 ms
        // time-out and fell back (tools/w_stress.py: 6 time-outs in 64 refits on 8 threads).  The refit is synchronous anyway (the abort word is
        // read back below), so a host lock from the first launch to that read costs nothing a contended chip would not have cost.
        g->w_done = false;
        if (exec_ok) { g->chol_form_last = 4; CHK(cholesky_exec(g, T)); }
        else if (form2_ok && g_chol_df2_ll) { g->chol_form_last = 3; CHK(cholesky_dataflow3(g, T)); }
        else if (form2_ok) { g->chol_form_last = 2; CHK(cholesky_dataflow2(g, T)); }
        else { g->chol_form_last = 1; CHK(cholesky_dataflow(g, T)); }
        t_end(g);
        if (!g->w_done) {
            t_begin(g, "tri_inverse");
            if (!g->w_seeded) {
                hipLaunchKernelGGL(k_inv128, dim3(T), dim3(PF_THREADS), POTF2_LDS_BYTES, g->stream, g->dL, ld, g->dW, g->dWT, ld);
                HIPCHK(hipGetLastError());
            }
            for (int h = 1; h < T; h *= 2) CHK(inverse_level(g, g->stream, 0, T, h));
            t_end(g);
        }
        t_begin(g, "alpha");
        CHK(compute_alpha(g));
        t_end(g);
        // the abort word and the pivot word come back through the pinned block behind the kernels: ONE host round trip per refit (three before:
        // this synchronisation, a blocking copy of the abort word, check_info's copy + synchronisation -- ~25 us of a 1.5 ms refit)
        unsigned aborted = 0;
        int* const pwords = g->hpin ? reinterpret_cast<int*>(g->hpin + 3 * SMALL_R) : nullptr;
        if (pwords) {
            HIPCHK(hipMemcpyAsync(pwords, g->dchol_flags + chol_abort_word(T), sizeof(unsigned), hipMemcpyDeviceToHost, g->stream));
            HIPCHK(hipMemcpyAsync(pwords + 1, g->dinfo, sizeof(int), hipMemcpyDeviceToHost, g->stream));
        }
        HIPCHK(hipStreamSynchronize(g->stream));
        if (pwords) aborted = (unsigned)pwords[0];
        else HIPCHK(hipMemcpy(&aborted, g->dchol_flags + chol_abort_word(T), sizeof(unsigned), hipMemcpyDeviceToHost));
        df_file.release();
        df_lock.unlock();
        if (aborted) {
            // a flag never arrived (e.g. two of the streams share a hardware queue on this system, or a spinning launch kept a
            // persistent workgroup off the chip): every wait has returned, nothing hangs; the factor -- and any pivot failure it
            // reports, hence this check BEFORE check_info -- is garbage.  Fall back to the launch-chained form for the rest of
            // the process (BOHIP_CHOL_DF_STRICT=1: report it instead, for tests and tools).
            g->chol_fallbacks++;
            g->chol_abort_T = T;
            {
                // the chain kernel's waits leave the word address of the flag that never arrived (bit 31 set); the executor's tasks and the
                // flagged launches of the older forms leave 1
                char which[96] = "counters of an executor task or a flagged launch";
                if (aborted & 0x80000000u) {
                    const unsigned base_w = (unsigned)(reinterpret_cast<uintptr_t>(g->dchol_flags) >> 2) | 0x80000000u;
                    snprintf(which, sizeof which, "flag word %u of %zu", (aborted - base_w) & 0x7fffffffu, chol_flag_words(T));
                }
                fprintf(stderr, "libbohip: dataflow factorisation (form %d, %d row tiles) timed out on a dependency (%s); using the launch-chained form for a while\n",
                        g->chol_form_last, T, which);
            }
            if (g_chol_df_dump) {   // diagnosis: which flags of the dataflow form never arrived
                std::vector<unsigned> hf(chol_flag_words(T));
                hipMemcpy(hf.data(), g->dchol_flags, hf.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
                const CholFlags fl = chol_flags_layout(g, T);
                auto at = [&](const unsigned* p) { return hf[(size_t)(p - g->dchol_flags)]; };
                fprintf(stderr, "  abort word 0x%x, words behind it:", aborted);
                for (int w_ = 1; w_ < 8; ++w_) fprintf(stderr, " %u", hf[chol_abort_word(T) + w_]);
                fprintf(stderr, "\n");
                for (int k = 0; k < T; ++k) {
                    fprintf(stderr, "  block %2d: panel", k);
                    for (int p = 0; p < CH_PANELS; ++p) fprintf(stderr, " %u", at(fl.panel + k * CH_PANELS + p));
                    fprintf(stderr, " | crit %u rest %u col %u farall %u colall %u | rows with xp[7] unset:", at(fl.crit + k), at(fl.rest + k), at(fl.col + k), at(fl.farall + k), at(fl.colall + k));
                    for (int i = k + 1; i < T; ++i) {
                        const unsigned* x = fl.xp + ((size_t)k * T + i) * CH_PANELS;
                        if (at(x + CH_PANELS - 1) == 0u) { int p0 = 0; while (p0 < CH_PANELS && at(x + p0) != 0u) ++p0; fprintf(stderr, " %d(p%d)", i, p0); }
                    }
                    fprintf(stderr, " | colr<8:");
                    for (int i = k + 2; i < T; ++i) if (at(fl.colr + (size_t)k * T + i) < 8u) fprintf(stderr, " %d(%u)", i, at(fl.colr + (size_t)k * T + i));
                    fprintf(stderr, "\n");
                }
            }
            if (g_chol_df_strict) return fail(BOHIP_E_HIP, "dataflow factorisation timed out on a dependency (BOHIP_CHOL_DF_STRICT)");
            {
                const int bo = std::min(1024, 2 * g_chol_df_backoff.load(std::memory_order_relaxed) + 8);
                g_chol_df_backoff.store(bo, std::memory_order_relaxed);
                g_chol_df_skip.store(bo + 1, std::memory_order_relaxed);   // (+1: the repeat of this very refit)
            }
            g->stale = true;
            return refit_once(g, jitter);
        }
        if (pwords) {
            if (pwords[1] != 0) {
                g->pivot = pwords[1];
                g->stale = true;
                return fail(BOHIP_E_NOTPD, "kernel matrix not positive definite at pivot " + std::to_string(pwords[1]));
            }
            g->pivot = 0;
        } else {
            CHK(check_info(g));
        }
        g->stale = false;
        g->n_factored = N;
        g->refits++;
        // a dataflow refit went through: the next time-out starts from half the pause (one transient episode late in a long
        // process must not cost 1024 refits on the slow form)
        for (int bo = g_chol_df_backoff.load(std::memory_order_relaxed); bo > 0;)
            if (g_chol_df_backoff.compare_exchange_weak(bo, bo / 2, std::memory_order_relaxed)) break;
        return 0;
    }
    g->chol_form_last = 0;
    // Right-looking on 128-column panels, with the trailing update applied in two tiers: inside an outer block
    // of OB panels only the block's own remaining columns are updated after every panel (K = 128); everything to
    // the right of the outer block is updated ONCE per outer block with K = 128 * OB.  The C tiles of the bulk
    // are therefore read-modified-written T/OB times instead of T times and the bulk contraction is OB x deeper.
    const int OB = 4;
    bool side_pending = false;
    // The bulk update of the columns right of the NEXT block runs on the side stream.  Run as one launch it shares the chip
    // with the next block's whole diagonal chain and slows the chain's small GEMMs 3-6x (N = 10^4: 90-120 us instead of
    // 15-25 us each).  So it is cut into pieces of equal area, and piece j is released by an event recorded just before the
    // j-th diagonal-block kernel of the next block starts: the pieces fill the ~70 us during which that single-workgroup
    // kernel leaves the chip idle and are (mostly) gone when the chain's GEMMs arrive.
    struct Pending { int ob = 0, oe = 0, c0 = 0, rest = 0, next = 0, pieces = 0; int cut[9] = {0}; } pend;
    auto bulk_params = [&](int pob, int poe, int r0t, int c0t, int mt, int nct) {
        GemmNTParams b{};  // A[i, j] -= L[i, pob:poe] L[j, pob:poe]'  on rows >= r0t, columns [c0t, c0t + nct)
        b.A = g->dS + (int64_t)r0t * TILE * ld + (int64_t)pob * TILE; b.lda = ld;
        b.B = g->dS + (int64_t)c0t * TILE * ld + (int64_t)pob * TILE; b.ldb = ld;
        b.C = g->dL + (int64_t)r0t * TILE * ld + (int64_t)c0t * TILE; b.ldc = ld;
        b.mt = mt; b.nt64 = 2 * nct; b.kc = (poe - pob) * (TILE / KC); b.alpha = -1.0; b.beta = 1.0;
        b.diag_skip = 1; b.row0 = (int64_t)r0t * TILE; b.col0 = (int64_t)c0t * TILE;
        return b;
    };
    auto release_piece = [&](bool gated) -> int {   // next piece of the pending bulk update -> side stream
        if (pend.next >= pend.pieces) return 0;
        if (gated) {
            HIPCHK(hipEventRecord(g->ev_gate, g->stream));
            HIPCHK(hipStreamWaitEvent(g->side_stream, g->ev_gate, 0));
        }
        const int a = pend.cut[pend.next], b = pend.cut[pend.next + 1];
        ++pend.next;
        if (b > a)   // columns [c0 + a, c0 + b) of the pending region, rows from the first of them down
            CHK(launch_gemm_nt(g, bulk_params(pend.ob, pend.oe, pend.c0 + a, pend.c0 + a, pend.rest - a, b - a), 1, g->side_stream));
        if (pend.next == pend.pieces) HIPCHK(hipEventRecord(g->ev_bulk, g->side_stream));
        return 0;
    };
    for (int ob = 0; ob < T; ob += OB) {
        const int oe = std::min(T, ob + OB);
        for (int kb = ob; kb < oe; ++kb) {
            double* Lkk = g->dL + (int64_t)kb * TILE * (ld + 1);
            double* Wkk = g->dW + (int64_t)kb * TILE * (ld + 1);
            CHK(release_piece(true));
            hipLaunchKernelGGL(k_potf2_inv, dim3(1), dim3(PF_THREADS), POTF2_LDS_BYTES, g->stream, Lkk, ld, Wkk,
                               g->dWT + (int64_t)kb * TILE * (ld + 1), ld, g->dinfo, kb * TILE);
            HIPCHK(hipGetLastError());
            const int rem = T - kb - 1;
            if (rem == 0) break;
            const int64_t poff = (int64_t)(kb + 1) * TILE * ld + (int64_t)kb * TILE;
            double* panel = g->dS + poff;  // solved panel L[kb+1:, kb] lives in the scratch matrix until the final copy
            GemmNTParams p{};  // panel solve L[i,kb] = A[i,kb] * inv(L_kk)'   (out of place: dL -> dS)
            p.A = g->dL + poff; p.lda = ld; p.B = Wkk; p.ldb = ld; p.C = panel; p.ldc = ld;
            p.mt = rem; p.nt64 = 2; p.kc = TILE / KC; p.alpha = 1.0; p.beta = 0.0;
            CHK(launch_gemm_nt(g, p));
            const int inner_cols = oe - kb - 1;  // remaining panels of this outer block
            if (inner_cols > 0) {
                GemmNTParams u{};  // A[i, j] -= L[i,kb] L[j,kb]'  for j in (kb, oe), i >= j
                u.A = panel; u.lda = ld; u.B = panel; u.ldb = ld;
                u.C = g->dL + (int64_t)(kb + 1) * TILE * (ld + 1); u.ldc = ld;
                u.mt = rem; u.nt64 = 2 * inner_cols; u.kc = TILE / KC; u.alpha = -1.0; u.beta = 1.0;
                u.diag_skip = 1; u.row0 = (int64_t)(kb + 1) * TILE; u.col0 = (int64_t)(kb + 1) * TILE;
                CHK(launch_gemm_nt(g, u));
            }
        }
        while (pend.next < pend.pieces) CHK(release_piece(false));   // a short last block: whatever is left goes now
        if (g_inv_overlap) {
            // The block's panels are final: invert its diagonal region and join it to the leading inverse on the inverse
            // stream, beside the next blocks' diagonal chain (which leaves most CUs idle).  Reads dS panels of columns
            // < oe (final), W/W' of tiles < oe; writes W/W' rows/columns of THIS block and the upper-right part of dS.
            HIPCHK(hipEventRecord(g->ev_blk, g->stream));
            HIPCHK(hipStreamWaitEvent(g->inv_stream, g->ev_blk, 0));
            for (int h = 1; h < oe - ob; h *= 2) CHK(inverse_level(g, g->inv_stream, ob, oe - ob, h));
            CHK(inverse_join(g, g->inv_stream, ob, oe - ob));
        }
        const int rem = T - oe;
        if (rem > 0) {
            // Bulk update with the OB solved panels: the columns of the NEXT outer block on the critical stream, everything
            // to their right in gated pieces on the side stream.  Hazards: (1) the side stream needs the panels -> ev_panels;
            // (2) the next block's columns were last written by the previous block's pieces -> ev_bulk (after the last one).
            const int nxt = std::min(OB, rem), rest = rem - nxt;
            if (rest > 0) {
                HIPCHK(hipEventRecord(g->ev_panels, g->stream));
                HIPCHK(hipStreamWaitEvent(g->side_stream, g->ev_panels, 0));
            }
            if (side_pending) HIPCHK(hipStreamWaitEvent(g->stream, g->ev_bulk, 0));
            CHK(launch_gemm_nt(g, bulk_params(ob, oe, oe, oe, rem, nxt)));
            if (rest > 0) {
                pend = Pending{};
                pend.ob = ob; pend.oe = oe; pend.c0 = oe + nxt; pend.rest = rest; pend.next = 0;
                pend.pieces = g_bulk_pieces > 0 ? std::min({g_bulk_pieces, nxt, rest}) : 1;
                // equal-area cuts of the lower-triangular region: the area left of column c is c rest - c (c - 1) / 2
                const double total = 0.5 * rest * (rest + 1.0);
                for (int q = 0, c = 0; q <= pend.pieces; ++q) {
                    const double want = total * q / pend.pieces;
                    while (c < rest && c * (double)rest - 0.5 * c * (c - 1.0) < want - 1e-9) ++c;
                    pend.cut[q] = q == pend.pieces ? rest : c;
                }
                if (pend.pieces == 1) CHK(release_piece(false));   // ungated single launch (small problems, BOHIP_BULK_PIECES=0)
                side_pending = true;
            } else {
                side_pending = false;
            }
        }
    }
    if (side_pending) HIPCHK(hipStreamWaitEvent(g->stream, g->ev_bulk, 0));
    if (T > 1) {
        hipLaunchKernelGGL(k_copy_offdiag_tiles, dim3(T * (T - 1) / 2), dim3(256), 0, g->stream, g->dS, g->dL, ld, T);
        HIPCHK(hipGetLastError());
    }
    t_end(g);
    t_begin(g, "tri_inverse");
    if (g_inv_overlap) {   // only the tail of the last block's join is still running
        HIPCHK(hipEventRecord(g->ev_inv, g->inv_stream));
        HIPCHK(hipStreamWaitEvent(g->stream, g->ev_inv, 0));
    } else {               // recursive doubling over the whole matrix after the factorisation
        for (int h = 1; h < T; h *= 2) CHK(inverse_level(g, g->stream, 0, T, h));
    }
    t_end(g);
    t_begin(g, "alpha");
    CHK(compute_alpha(g));
    t_end(g);
    CHK(check_info(g));
    g->stale = false;
    g->n_factored = N;
    g->refits++;
    return 0;
}

// out[r][j] = (W rows[r])[j] for r < P <= 32: one launch, workgroups of 8 rows of W x 8 right-hand sides
static int launch_rows_trimv(bohip_gp* g, const double* W, int64_t N0, const double* rows, int P, double* out, int upper) {
    if (N0 <= 0 || P <= 0) return 0;
    // one workgroup per CU walks its share of the 8-row blocks of W (k_trimv_stream, kernels_linalg.hip); 16 right-hand sides per pass
    const int64_t nblk = (N0 + 7) / 8;
    const unsigned wgs = (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::max(device_cus(), 1), nblk));
    hipLaunchKernelGGL(k_trimv_stream<TRIMV_D>, dim3(wgs, (unsigned)((P + 15) / 16)), dim3(TRIMV_THREADS), trimv_lds_bytes(TRIMV_D), g->stream,
                       W, g->ld, N0, rows, g->ld, P, out, g->ld, upper, (const unsigned*)g->asc_go);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- A2': incremental extension by p new observations (already in dX/dy and the host mirror) ----------
// defer_info != nullptr: the pivot word is copied there (pinned) behind the kernels and looked at by the caller after ITS synchronisation
// (one host round trip less per append)
static int append_incremental(bohip_gp* g, int64_t N0, int64_t p, int* defer_info = nullptr) {
    CHK(one_time_kernel_setup());
    const int64_t N1 = N0 + p, Npad1 = round_up(N1 + 1, TILE), ld = g->ld;
    const KernelHyper hp = make_hyper(g);
    // (a refit that needed jitter J put J on every diagonal entry: the appended rows carry it too, so that the factor
    // stays the factor of ONE matrix -- cK + J I -- that the oracle's fit_with_jitter reproduces)
    const double noise = std::exp(2.0 * g->lognoise) + std::numeric_limits<double>::epsilon() + g->jitter_last;
    HIPCHK(hipMemsetAsync(g->dinfo, 0, sizeof(int), g->stream));
    t_begin(g, "append_cov_rows");
    hipLaunchKernelGGL(k_cov_rows, dim3((Npad1 + 255) / 256, Npad1 - N0), dim3(256), 0, g->stream, g->dX, N0, N1, Npad1,
                       hp, noise, g->dL, g->dW, g->dWT, ld);
    HIPCHK(hipGetLastError());
    t_end(g);
    t_begin(g, "append_L21");
    CHK(launch_rows_trimv(g, g->dW, N0, g->dL + N0 * ld, (int)p, g->dApp, 0));
    HIPCHK(hipMemcpy2DAsync(g->dL + N0 * ld, ld * 8, g->dApp, ld * 8, N0 * 8, p, hipMemcpyDeviceToDevice, g->stream));
    t_end(g);
    t_begin(g, "append_schur_chol");
    const double* schur_in = nullptr;
    if (p >= 3) {      // the Schur complement's inner products, one workgroup each (same bits as k_schur_chol's own loop, which p <= 2 keeps: one launch less)
        if (!g->dschur) HIPCHK(hipMalloc(&g->dschur, (size_t)APPEND_PMAX * APPEND_PMAX * 8));
        hipLaunchKernelGGL(k_schur_dots, dim3((unsigned)(p * (p + 1) / 2)), dim3(256), 0, g->stream, g->dL, ld, N0, (int)p, g->dschur);
        schur_in = g->dschur;
    }
    hipLaunchKernelGGL(k_schur_chol, dim3(1), dim3(256), 0, g->stream, g->dL, g->dW, g->dWT, ld, N0, (int)p, g->dinfo, schur_in);
    HIPCHK(hipGetLastError());
    t_end(g);
    t_begin(g, "append_W21");
    double* Tm = g->dApp + (int64_t)APP_UT_ROW0 * ld;   // T = L21 W11, row-wise on W' (k >= c)
    CHK(launch_rows_trimv(g, g->dWT, N0, g->dL + N0 * ld, (int)p, Tm, 1));
    hipLaunchKernelGGL(k_apply_w22, dim3((N0 + 255) / 256), dim3(256), 0, g->stream, g->dW, g->dWT, ld, N0, (int)p, Tm, ld);
    HIPCHK(hipGetLastError());
    t_end(g);
    t_begin(g, "alpha");
    if (g_append_alpha_inc && g->alpha_inc_run < 256) {      // (every 256 appends the full product: the increments' rounding does not accumulate without bound)
        g->alpha_inc_run++;
        // alpha_new = [alpha_old + W21' u2; W22' u2] (kernels_linalg.hip k_alpha_append_*); BOHIP_APPEND_ALPHA_INC=0: the full product
        hipLaunchKernelGGL(k_alpha_append_u, dim3((unsigned)p), dim3(1024), 0, g->stream, g->dW, ld, N0, (int)p, g->dy, g->beta, g->dr, g->dt);
        hipLaunchKernelGGL(k_alpha_append_apply, dim3((unsigned)((N1 + 255) / 256)), dim3(256), 0, g->stream, g->dW, ld, N0, (int)p, g->dt, g->dalpha);
        HIPCHK(hipGetLastError());
    } else {
        CHK(compute_alpha(g));
    }
    t_end(g);
    if (defer_info) HIPCHK(hipMemcpyAsync(defer_info, g->dinfo, sizeof(int), hipMemcpyDeviceToHost, g->stream));
    else CHK(check_info(g));
    g->n_factored = N1;
    g->appends++;
    return 0;
}

static int ensure_fresh(bohip_gp* g) {
    if (g->dL == nullptr || g->cap < g->n) CHK(alloc_model(g, std::max<int64_t>(g->n, 1)));   // a failed growth left no buffers
    if (g->stale || g->n_factored != g->n) return refit(g);
    return 0;
}

// ---- scoring --------------------------------------------------------------------------------------
static int64_t chunk_cap(const bohip_gp* g) {
    // A K*' chunk of about the size of the 256 MB Infinity Cache, in multiples of 512 candidates (8 XCDs x one 64-wide candidate
    // tile: k_trigemm_sq deals candidate tiles to XCDs round-robin).  Rounds 1-3 kept the chunk at 128 MB "so that it stays in the
    // cache next to W"; the sweep of round 4 (tools/chunk_sweep.py, profiles/r04_chunk_sweep.txt) says larger is better up to
    // ~200-330 MB -- fewer launches, each with more jobs per slot and a smaller share of ragged end -- and worse beyond:
    // N = 3000, R = 32768: 4096 per chunk 5.15 ms, 8192: 4.96, 16384: 5.09;  N = 10^4, R = 4096: 1024: 7.28, 2048: 6.49, 4096: 6.43;
    // N = 6000, R = 16384: 4096: 9.32, 8192: 9.42, 16384: 10.1;  N = 1000, R = 65536: 8192: 1.71, 32768: 1.57, 65536: 1.62.
    int64_t rows = (int64_t)(256.0 * 1024 * 1024 / (8.0 * g->ld));
    rows = std::max<int64_t>(1024, std::min<int64_t>(65536, rows / 512 * 512));
    return rows;
}
// Chunks of ONE batch are equal-sized multiples of 512 candidates: R = 32768 at N = 3000 is 8 x 4096, not 6 x 5376 + 512
// (a ragged last chunk is a launch with a few candidate tiles per XCD -- all tail; measured 5.39 against 5.08 ms).  Among the
// chunk counts that fit the cap the one that leaves the least padding wins, fewer chunks on a tie.
static int64_t chunk_rows(const bohip_gp* g, int64_t R) {
    if (g_chunk_rows_forced > 0) return std::min<int64_t>(round_up(std::max<int64_t>(R, 1), TILE), round_up(g_chunk_rows_forced, TILE));
    const int64_t cap = chunk_cap(g), R512 = round_up(std::max<int64_t>(R, 1), 512);
    if (R512 <= cap) return round_up(std::max<int64_t>(R, 1), TILE);   // one chunk (its buffer keeps the 128-row granularity)
    const int64_t nmin = (R512 + cap - 1) / cap;
    int64_t best_rows = cap, best_pad = INT64_MAX;
    for (int64_t n = nmin; n <= nmin + 4; ++n) {
        const int64_t rows = round_up((R512 + n - 1) / n, 512), pad = n * rows - R512;
        if (rows <= cap && pad < best_pad) { best_pad = pad; best_rows = rows; }
    }
    return best_rows;
}
static int ensure_score_scratch(bohip_gp* g, int64_t R) {
    const int64_t Rpad = round_up(std::max<int64_t>(R, 1), TILE);
    const int64_t T = g->ld / TILE;
    const int64_t rc = chunk_rows(g, R);
    g->chunk_now = rc;
    const int64_t SLACK = TILE;  // head-room so tile-granular writes past the last candidate stay inside the buffers
    if (g->kst_rows < rc || g->dKsT == nullptr) {
        if (g->dKsT) hipFree(g->dKsT);
        g->dKsT = nullptr;
        HIPCHK(hipMalloc(&g->dKsT, (size_t)(rc + SLACK) * g->ld * 8));
        g->kst_rows = rc;
    }
    if (g->q_cap < 2 * T * (Rpad + SLACK)) {   // two partial sums per row tile (k_trigemm_sq's row pieces)
        if (g->dq) hipFree(g->dq);
        g->dq = nullptr;
        HIPCHK(hipMalloc(&g->dq, (size_t)(2 * T * (Rpad + SLACK)) * 8));
        g->q_cap = 2 * T * (Rpad + SLACK);
    }
    if (g->r_cap < Rpad) {
        for (double** p : {&g->dmu_raw, &g->dmu, &g->dvar, &g->dscore})
            if (*p) { hipFree(*p); *p = nullptr; }
        HIPCHK(hipMalloc(&g->dmu_raw, (Rpad + SLACK) * 8));
        HIPCHK(hipMalloc(&g->dmu, Rpad * 8));
        HIPCHK(hipMalloc(&g->dvar, Rpad * 8));
        HIPCHK(hipMalloc(&g->dscore, Rpad * 8));
        g->r_cap = Rpad;
    }
    const int64_t tiles = Rpad / CTILE + 2;
    if (g->fz_cap < tiles) {
        if (g->dfz_cnt) hipFree(g->dfz_cnt);
        if (g->dfz_best) hipFree(g->dfz_best);
        g->dfz_cnt = nullptr; g->dfz_best = nullptr; g->fz_cap = 0;
        HIPCHK(hipMalloc(&g->dfz_cnt, (size_t)(tiles + 1 + 16) * sizeof(unsigned)));   // + the job cursors of k_trigemm_sq_pull
        HIPCHK(hipMalloc(&g->dfz_best, (size_t)tiles * sizeof(Best)));
        HIPCHK(hipMemsetAsync(g->dfz_cnt, 0, (size_t)(tiles + 1 + 16) * sizeof(unsigned), g->stream));
        g->fz_cap = tiles;
    }
    const int64_t nb = (R + 255) / 256 + 1;
    if (g->bb_cap < nb) {
        if (g->dblock_best) hipFree(g->dblock_best);
        g->dblock_best = nullptr;
        HIPCHK(hipMalloc(&g->dblock_best, nb * sizeof(Best)));
        g->bb_cap = nb;
    }
    return 0;
}
static int ensure_grad_scratch(bohip_gp* g) {
    if (g->vt_rows >= g->kst_rows && g->dVT) return 0;
    for (double** p : {&g->dVT, &g->dUT})
        if (*p) { hipFree(*p); *p = nullptr; }
    const size_t bytes = (size_t)g->kst_rows * g->ld * 8;
    HIPCHK(hipMalloc(&g->dVT, bytes));
    HIPCHK(hipMalloc(&g->dUT, bytes));
    HIPCHK(hipMemsetAsync(g->dVT, 0, bytes, g->stream));  // padding columns (>= N) must stay zero
    g->vt_rows = g->kst_rows;
    return 0;
}
static int ensure_xs(bohip_gp* g, int64_t R) {
    if (g->xs_cap < R * g->d) {
        if (g->dXs) hipFree(g->dXs);
        g->dXs = nullptr;
        HIPCHK(hipMalloc(&g->dXs, std::max<size_t>(8, (size_t)R * g->d * 8)));
        g->xs_cap = R * g->d;
    }
    return 0;
}

template <int DT>
static void launch_kstar(bohip_gp* g, const double* dXs, int64_t r0, int64_t r1, int64_t Npad, const KernelHyper& hp) {
    // 16 candidates per block: N/256 x R/16 blocks keep >= 8 waves per SIMD in flight (latency-bound loop); a handful of candidates:
    // two per block (ten in a row on 12 workgroups were a serial chain of ten `exp`s per thread)
    const int rb = r1 - r0 <= 32 ? 2 : 16;
    dim3 grid((Npad + 255) / 256, (r1 - r0 + rb - 1) / rb);
    hipLaunchKernelGGL(k_kstar<DT>, grid, dim3(256), 0, g->stream, g->dX, g->n, Npad, dXs, r0, r1, hp, g->dKsT, g->ld, rb);
}

// The row pieces of k_trigemm_sq (kernels_score.hip), heaviest first: which row tiles go as two 64-row halves is a function
// of the number of row tiles and of where the alpha row sits -- nothing else, so a candidate's score does not depend on the batch it
// is scored in.  Default: NONE (whole tiles; only a last tile whose live rows fit its upper half runs as a half piece, as it always
// did).  BOHIP_TRIGEMM_HALVE="lo,hi" halves the tiles lo <= rt < hi -- the round-4 experiment: tools/sim_trigemm_tail.py predicted
// -3.7 % for the shortest third of the tiles (the launch ends evenly: idle tail 3.2 -> 0.4 % in the model), the chip gave 0.0 %
// (profiles/r04_trigemm_halve_sweep.txt: "0,8" 609 us against 609; "0,12" 621; all halves 665): MI355X clocks to its power
// budget, CUs that idle at the end of the launch hand their share of it to the ones still working, and a half job spends more
// LDS reads per flop.
static void trigemm_pieces(int T, int64_t alpha_row, std::vector<int>& out) {
    int lo = 0, hi = 0;
    if (g_halve_lo >= 0) { lo = std::min(g_halve_lo, T); hi = std::min(std::max(g_halve_hi, lo), T); }
    struct P { double cost; int code; };
    std::vector<P> ps;
    for (int rt = 0; rt < T; ++rt) {
        const int64_t live = alpha_row + 1 - (int64_t)rt * TILE;   // live rows of this tile (the last one: up to the alpha row)
        if (live <= TILE / 2) { ps.push_back({(rt + 0.5) / 2.0, rt | PIECE_UPPER_SOLO << 16}); continue; }
        if (rt >= lo && rt < hi) {
            ps.push_back({(rt + 0.5) / 2.0, rt | PIECE_UPPER << 16});
            ps.push_back({(rt + 1.0) / 2.0, rt | PIECE_LOWER << 16});
        } else {
            ps.push_back({rt + 1.0, rt | PIECE_WHOLE << 16});
        }
    }
    std::stable_sort(ps.begin(), ps.end(), [](const P& a, const P& b) { return a.cost > b.cost; });
    out.clear();
    for (const P& q : ps) out.push_back(q.code);
}
static int ensure_pieces(bohip_gp* g, int T, int64_t alpha_row) {
    if (g->pieces_T == T && g->pieces_alpha == alpha_row && g->dpieces) return 0;
    std::vector<int> ps;
    trigemm_pieces(T, alpha_row, ps);
    if (T > 65535) return fail(BOHIP_E_UNSUPPORTED, "too many row tiles");
    if ((int64_t)ps.size() > g->pieces_cap) {
        if (g->dpieces) { HIPCHK(hipStreamSynchronize(g->stream)); HIPCHK(hipFree(g->dpieces)); g->dpieces = nullptr; }
        g->pieces_cap = 2 * (int64_t)ps.size() + 64;
        HIPCHK(hipMalloc(&g->dpieces, (size_t)g->pieces_cap * sizeof(int)));
    }
    // (stream-ordered: a launch of an earlier call may still be reading the old table)
    g->hpieces = ps;
    HIPCHK(hipMemcpyAsync(g->dpieces, g->hpieces.data(), ps.size() * sizeof(int), hipMemcpyHostToDevice, g->stream));
    g->n_pieces = (int)ps.size();
    g->pieces_T = T; g->pieces_alpha = alpha_row;
    return 0;
}

// V = W K* with the fused epilogue (k_trigemm_sq): one launch per K*' chunk.
static int launch_trigemm(bohip_gp* g, int T, int64_t ncand, int64_t N, int64_t Rpad, int64_t r0, double* VT,
                          FuseParams fz = FuseParams{}) {
    const int CT = (int)((ncand + CTILE - 1) / CTILE), n_local = (CT + 7) / 8;
    CHK(ensure_pieces(g, T, N));
    const int NP = g->n_pieces;
    fz.T = T;
    if (g->timing) {   // core clock under the dominant kernel (BOHIP_INFO_KERNEL_CLOCK_MHZ)
        if (!g->dclk) {
            HIPCHK(hipMalloc(&g->dclk, 2 * sizeof(unsigned long long)));
            HIPCHK(hipMemsetAsync(g->dclk, 0, 2 * sizeof(unsigned long long), g->stream));
        }
        fz.clk = g->dclk;
    }
    if (g_ks8)
        hipLaunchKernelGGL(k_trigemm_sq<2>, dim3(8 * n_local * NP), dim3(GEMM_THREADS_8), glds3_lds_bytes<4>(), g->stream, g->dW,
                           g->ld, g->dKsT, g->ld, g->dpieces, NP, CT, N, g->dq, Rpad, g->dmu_raw, r0, VT, g->ld, fz);
    else
        hipLaunchKernelGGL(k_trigemm_sq<1>, dim3(8 * n_local * NP), dim3(GEMM_THREADS), glds3_lds_bytes<4>(), g->stream, g->dW,
                           g->ld, g->dKsT, g->ld, g->dpieces, NP, CT, N, g->dq, Rpad, g->dmu_raw, r0, VT, g->ld, fz);
    HIPCHK(hipGetLastError());
    return 0;
}

// posterior pass over all R candidates: fills dq (partials) and dmu_raw.  VT optional (chunk-local).
template <int DT>
static void launch_kstar(bohip_gp* g, const double* dXs, int64_t r0, int64_t r1, int64_t Npad, const KernelHyper& hp);
static int launch_kstar_any(bohip_gp* g, const double* dXs, int64_t r0, int64_t r1, int64_t Npad, const KernelHyper& hp) {
    if (g->d <= 2) launch_kstar<2>(g, dXs, r0, r1, Npad, hp);
    else if (g->d <= 4) launch_kstar<4>(g, dXs, r0, r1, Npad, hp);
    else if (g->d <= 8) launch_kstar<8>(g, dXs, r0, r1, Npad, hp);
    else if (g->d <= 16) launch_kstar<16>(g, dXs, r0, r1, Npad, hp);
    else if (g->d <= 32) launch_kstar<32>(g, dXs, r0, r1, Npad, hp);
    else launch_kstar<64>(g, dXs, r0, r1, Npad, hp);
    HIPCHK(hipGetLastError());
    return 0;
}

// Small-batch posterior (R <= SMALL_R): V' rows into dApp[0..R), q and mu_raw; optionally U' = V' W into dApp[APP_UT_ROW0..).
static int ensure_small_counters(bohip_gp* g) {
    if (g->dgparts) return 0;
    HIPCHK(hipMalloc(&g->dgparts, (size_t)SMALL_MAX * 16 * (2 * DMAX + 2) * 8));
    HIPCHK(hipMalloc(&g->dgcount, (SMALL_MAX + 1) * sizeof(unsigned)));   // [0, SMALL_MAX): k_grad_finish, [SMALL_MAX]: k_small_finish
    HIPCHK(hipMemsetAsync(g->dgcount, 0, (SMALL_MAX + 1) * sizeof(unsigned), g->stream));
    return 0;
}
static bool split_applicable(const bohip_gp* g);
static int64_t small_limit(const bohip_gp* g) {
    if (g_small_r >= 0) return g_small_r;
    // measured break-even with the MFMA path: whole-K jobs (N=500: > 256, N=3000: ~190, N=10000: ~110); where the split-K
    // form applies it wins earlier (N=1500: ~150, N=3000: ~95, N=10000: ~55)
    // (round 4, k_trimv_stream: the row-wise products take 16 right-hand sides per pass over W, so the break-even moves in steps of 16:
    // N = 1000: ~240, N = 3000: 80, N = 10^4: 32 -- tools/small_limit_sweep.py, profiles/r04_small_limit_sweep.txt)
    // (round 5, kernels_small.hip: 256 at N = 1000, 96 at N = 3000, 32 at N = 10^4 -- profiles/r05_small_limit_sweep.txt)
    if (split_applicable(g)) return std::min<int64_t>(SMALL_MAX, std::max<int64_t>(16, (20 + 260000 / std::max<int64_t>(g->n, 1)) / 16 * 16));
    return std::min<int64_t>(SMALL_MAX, 90 + 300000 / std::max<int64_t>(g->n, 1));
}
// the batch size the path decision is based on (see bohip_gp_set_batch_hint)
static int64_t path_R(const bohip_gp* g, int64_t R) { return std::max(R, g->batch_hint); }
// ---- round 5: the small-batch pass as two MFMA kernels (kernels_small.hip) ------------------------------------------------------------
#ifdef BOHIP_SMALL_TRACE
static unsigned long long* g_small_trace = nullptr;
extern "C" int bohip_debug_small_trace_read(unsigned long long* out) {   // measurement build only (tools/small_pass_trace.py)
    if (!g_small_trace) return -1;
    return hipMemcpy(out, g_small_trace, 8192 * 16 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif
template <int DT, int G>
static void launch_small_pass(bohip_gp* g, const SmallCommon& sc, const SmallCommon& scu, const SmallV& sv, const SmallU* su, const KernelHyper& hp,
                              int npass) {
    hipLaunchKernelGGL((k_small_v<DT, G>), dim3((unsigned)sc.ntiles, (unsigned)npass), dim3(SP_THREADS), 0, g->stream, sc, sv, hp);
    if (su) hipLaunchKernelGGL((k_small_u<DT, G>), dim3((unsigned)sc.ntiles, (unsigned)npass), dim3(SP_THREADS), 0, g->stream, scu, *su, hp);
}
template <int DT>
static void launch_small_pass_g(bohip_gp* g, int G, const SmallCommon& sc, const SmallCommon& scu, const SmallV& sv, const SmallU* su,
                                const KernelHyper& hp, int npass) {
    if constexpr (DT <= 16) {     // (the wide-dimension instantiations are compiled for four MFMA groups only: build time)
        if (G == 1) return launch_small_pass<DT, 1>(g, sc, scu, sv, su, hp, npass);
        if (G == 2) return launch_small_pass<DT, 2>(g, sc, scu, sv, su, hp, npass);
        if (G == 3) return launch_small_pass<DT, 3>(g, sc, scu, sv, su, hp, npass);
    }
    launch_small_pass<DT, 4>(g, sc, scu, sv, su, hp, npass);
}
// all R <= SMALL_MAX candidates: values (mu, sigma^2, score, arg-max record) and, with d_grad, the gradient of the score
static int small_pass_mfma(bohip_gp* g, const double* dXs, int64_t R, const AcqParams& ap, double* d_mu, double* d_var, double* d_score,
                           Best* d_best, int64_t best_off, double* d_grad) {
    const int N = (int)g->n, T = (N + TILE - 1) / TILE, P = (int)R, npass = (P + 15) / 16, DTm = g->d <= 2 ? 2 : g->d <= 4 ? 4 : g->d <= 8 ? 8 : g->d <= 16 ? 16 : g->d <= 32 ? 32 : 64;
    // chunks per tile: every tile's partial sums cost one 16-KiB write and one read by the block's finisher, a column block's finisher adds up
    // to ceil(T / m) of them; BOHIP_SMALL_M overrides (tools/small_batch_bench.py sweeps it)
    // (one workgroup of these kernels per CU: all tiles of a pass are resident at once when there are no more tiles than CUs)
    int m = g_small_m;
    if (m <= 0) { const int cus = std::max(device_cus(), 1); for (m = 1; m < T && small_ntiles(T, m) > cus; ++m) {} }
    const int ntiles = small_ntiles(T, m);
    const size_t n_part = (size_t)npass * ntiles * 2048, n_v16 = (size_t)npass * T * 128 * 16, n_rec = (size_t)npass * T * 16,
                 n_g = (size_t)npass * ntiles * 16 * DTm, n_gm = (size_t)npass * T * 16 * DTm;
    const size_t need = (n_part + 2 * n_v16 + 2 * n_rec + (size_t)npass * 32 + SMALL_MAX + n_g + n_gm) * 8;
    if (need > g->sm_bytes) {
        if (g->dsm) { HIPCHK(hipStreamSynchronize(g->stream)); HIPCHK(hipFree(g->dsm)); g->dsm = nullptr; g->sm_bytes = 0; }
        HIPCHK(hipMalloc(&g->dsm, need + need / 4));
        g->sm_bytes = need + need / 4;
    }
    if (!g->dsm_cnt || T > g->sm_cnt_T) {
        if (g->dsm_cnt) { HIPCHK(hipStreamSynchronize(g->stream)); HIPCHK(hipFree(g->dsm_cnt)); g->dsm_cnt = nullptr; }
        const int Tc = T + 16;
        // [V counters | U counters]
        const size_t words = 2 * ((size_t)(SMALL_MAX / 16) * (Tc + 1) + 1);
        HIPCHK(hipMalloc(&g->dsm_cnt, words * sizeof(unsigned)));
        HIPCHK(hipMemsetAsync(g->dsm_cnt, 0, words * sizeof(unsigned), g->stream));
        g->sm_cnt_T = Tc;
    }
    const KernelHyper hp = make_hyper(g);
    SmallCommon sc{};
    sc.A = g->dWT; sc.ld = g->ld; sc.N = N; sc.T = T; sc.m = m; sc.P = P; sc.part = g->dsm; sc.ntiles = ntiles; sc.cnt = g->dsm_cnt;
    sc.X = g->dX; sc.Xs = dXs; sc.alpha = g->dalpha; sc.go = (const unsigned*)g->asc_go;
#ifdef BOHIP_SMALL_TRACE
    if (!g_small_trace) { HIPCHK(hipMalloc(&g_small_trace, 8192 * 16 * 8)); }
    HIPCHK(hipMemsetAsync(g_small_trace, 0, 8192 * 16 * 8, g->stream));
    sc.trace = g_small_trace;
#endif
    const size_t cnt_block = (size_t)(SMALL_MAX / 16) * (g->sm_cnt_T + 1) + 1;
    SmallV sv{};
    sv.v16 = g->dsm + n_part; sv.qpart = sv.v16 + n_v16; sv.mupart = sv.qpart + n_rec;
    sv.fstash = sv.mupart + n_rec + (size_t)npass * 32;
    sv.sigma2 = std::exp(2.0 * g->logsig); sv.beta = g->beta; sv.ap = ap;
    sv.mu_out = d_mu; sv.var_out = d_var; sv.score_out = d_score; sv.best_out = d_best; sv.idx_off = (long long)best_off;
    sv.finish = d_grad ? 0 : 1;
    SmallU su{};
    sv.ks16 = sv.fstash + SMALL_MAX;
    su.v16 = sv.v16; su.sv = sv; su.gpart = sv.ks16 + n_v16; su.gmpart = su.gpart + n_g; su.grad = d_grad;
    if (g->asc_fold && d_grad && npass == 1 && g->d <= 16 && 16 * ((g->d + 1) / 2) <= SP_THREADS - 128) {
        su.fold = *g->asc_fold;
        g->asc_fold_done = true;
    }
    const SmallU* sup = d_grad ? &su : nullptr;
    SmallCommon scu = sc;       // the U pass: W itself (contraction k >= c), its own partial planes and counters
    scu.A = g->dW; scu.cnt = g->dsm_cnt + cnt_block;
    const int G = (std::min(P, 16) + 3) / 4;
    t_begin(g, d_grad ? "small_V+U" : "small_V");
    switch (DTm) {
        case 2: launch_small_pass_g<2>(g, G, sc, scu, sv, sup, hp, npass); break;
        case 4: launch_small_pass_g<4>(g, G, sc, scu, sv, sup, hp, npass); break;
        case 8: launch_small_pass_g<8>(g, G, sc, scu, sv, sup, hp, npass); break;
        case 16: launch_small_pass_g<16>(g, G, sc, scu, sv, sup, hp, npass); break;
        case 32: launch_small_pass_g<32>(g, G, sc, scu, sv, sup, hp, npass); break;
        default: launch_small_pass_g<64>(g, G, sc, scu, sv, sup, hp, npass); break;
    }
    HIPCHK(hipGetLastError());
    t_end(g);
    g->q_tiles = 0;
    return 0;
}

// candidates [r0, r1), at most SMALL_MAX of them; the output pointers are indexed by the GLOBAL candidate number
static int small_posterior(bohip_gp* g, const double* dXs, int64_t r0, int64_t r1, bool want_u, const AcqParams& ap,
                           double* d_mu, double* d_var, double* d_score, Best* d_best, int64_t best_off = 0) {
    if (g_small_mfma && r0 == 0 && !want_u) return small_pass_mfma(g, dXs, r1, ap, d_mu, d_var, d_score, d_best, best_off, nullptr);
    CHK(ensure_small_counters(g));
    const int64_t N = g->n, Npad = round_up(N + 1, TILE), ld = g->ld;
    const int P = (int)(r1 - r0);
    const KernelHyper hp = make_hyper(g);
    t_begin(g, "kstar");
    CHK(launch_kstar_any(g, dXs, r0, r1, Npad, hp));
    t_end(g);
    t_begin(g, "small_V");
    // rows 0..N of W (row N carries alpha'): V'[r][j] = sum_{k<=j} W[j][k] K*'[r][k]
    CHK(launch_rows_trimv(g, g->dW, N + 1, g->dKsT, P, g->dApp, 0));
    hipLaunchKernelGGL(k_small_finish, dim3((unsigned)P), dim3(256), 0, g->stream, g->dApp, ld, N, P, g->dq + r0, g->dmu_raw + r0,
                       g->dgcount + SMALL_MAX, std::exp(2.0 * g->logsig), g->beta, ap, d_mu ? d_mu + r0 : nullptr,
                       d_var ? d_var + r0 : nullptr, d_score ? d_score + r0 : nullptr, d_best, (long long)best_off);
    HIPCHK(hipGetLastError());
    t_end(g);
    g->q_tiles = 0;
    if (want_u) {
        t_begin(g, "small_U");
        // U'[r][c] = sum_{k>=c} W'[c][k] V'[r][k]
        CHK(launch_rows_trimv(g, g->dWT, N, g->dApp, P, g->dApp + (int64_t)APP_UT_ROW0 * ld, 1));
        t_end(g);
    }
    return 0;
}

static int posterior_pass(bohip_gp* g, const double* dXs, int64_t R, const FuseParams& fz = FuseParams{}) {
    CHK(one_time_kernel_setup());
    const int64_t N = g->n, Npad = round_up(N + 1, TILE), Rpad = round_up(R, TILE) + TILE;  // +TILE: head-room for tile-granular writes
    const int T = (int)(Npad / TILE);
    g->q_tiles = T;
    const KernelHyper hp = make_hyper(g);
    const int64_t rc = g->chunk_now;
    g->score_launches = 0;
    for (int64_t r0 = 0; r0 < R; r0 += rc) {
        const int64_t r1 = std::min(R, r0 + rc);
        ++g->score_launches;
        t_begin(g, "kstar");
        CHK(launch_kstar_any(g, dXs, r0, r1, Npad, hp));
        t_end(g);
        t_begin(g, "trigemm_sq");
        CHK(launch_trigemm(g, T, r1 - r0, N, Rpad, r0, nullptr, fz));
        t_end(g);
    }
    return 0;
}

// ---- split-K posterior for batches too small to fill the chip with whole-K jobs (see k_split_combine_v) -----------
struct SplitPlan { int kz = 0, nsl = 0; };
static bool split_applicable(const bohip_gp* g) {
    return g_split && round_up(g->n + 1, TILE) / TILE >= 8;   // short K extents gain nothing
}
static SplitPlan split_plan(const bohip_gp* g, int64_t R) {
    SplitPlan sp;
    if (!split_applicable(g)) return sp;
    const int64_t Npad = round_up(g->n + 1, TILE);
    const int T = (int)(Npad / TILE);
    const int64_t CT64 = (path_R(g, R) + CTILE - 1) / CTILE;
    if (CT64 * T >= 512) return sp;                     // enough whole-K jobs for every workgroup slot
    const int units = std::max(2, (T + 7) / 8);         // 128-wide K units per slice
    int kz = units * (TILE / KC);
    int nsl = (T * (TILE / KC) + kz - 1) / kz;
    const int64_t Rc = round_up(R, TILE);
    while (nsl > 1 && (double)nsl * Rc * g->ld * 8.0 > 256.0 * 1024 * 1024) {   // keep the partial planes below 256 MB
        kz *= 2;
        nsl = (T * (TILE / KC) + kz - 1) / kz;
    }
    if (nsl < 2) return sp;
    sp.kz = kz; sp.nsl = nsl;
    return sp;
}
static int split_posterior(bohip_gp* g, const double* dXs, int64_t R, const SplitPlan& sp, bool want_u) {
    CHK(one_time_kernel_setup());
    const int64_t N = g->n, Npad = round_up(N + 1, TILE), ld = g->ld, Rc = round_up(R, TILE);
    const int T = (int)(Npad / TILE);
    const KernelHyper hp = make_hyper(g);
    const int64_t need = (int64_t)sp.nsl * Rc * ld;
    if (g->split_cap < need) {
        if (g->dsplit) HIPCHK(hipFree(g->dsplit));
        g->dsplit = nullptr; g->split_cap = 0;
        HIPCHK(hipMalloc(&g->dsplit, (size_t)need * 8));
        g->split_cap = need;
    }
    if (want_u) CHK(ensure_grad_scratch(g));
    t_begin(g, "kstar");
    CHK(launch_kstar_any(g, dXs, 0, R, Npad, hp));
    t_end(g);
    t_begin(g, "split_V");
    GemmNTParams p{};   // partial V'[r][i] = sum over the slice's k of K*'[r][k] W[i][k]  (A = W row tiles, B = K*')
    p.A = g->dW; p.lda = ld; p.B = g->dKsT; p.ldb = ld; p.C = nullptr; p.CT = g->dsplit; p.ldct = ld;
    p.zA = (int64_t)sp.kz * KC; p.zB = (int64_t)sp.kz * KC; p.zCT = Rc * ld;
    p.mt = T; p.nt64 = (int)((R + CTILE - 1) / CTILE); p.kc = sp.kz; p.alpha = 1.0; p.beta = 0.0;
    p.khi_from_m = 1; p.kz = sp.kz; p.ktot = T * (TILE / KC);
    CHK(launch_gemm_nt(g, p, sp.nsl));
    hipLaunchKernelGGL(k_split_combine_v, dim3((unsigned)R), dim3(256), 0, g->stream, g->dsplit, Rc * ld, ld, sp.kz, N, g->dq,
                       g->dmu_raw, want_u ? g->dVT : nullptr, ld);
    HIPCHK(hipGetLastError());
    t_end(g);
    g->q_tiles = 0;
    if (want_u) {
        t_begin(g, "split_U");
        GemmNTParams u{};   // partial U'[r][j] = sum over the slice's i of V'[r][i] W'[j][i]   (B = W' upper-triangular: i >= j)
        u.A = g->dVT; u.lda = ld; u.B = g->dWT; u.ldb = ld; u.C = g->dsplit; u.ldc = ld;
        u.zA = (int64_t)sp.kz * KC; u.zB = (int64_t)sp.kz * KC; u.zC = Rc * ld;
        u.mt = (int)(Rc / TILE); u.nt64 = 2 * T; u.kc = sp.kz; u.alpha = 1.0; u.beta = 0.0;
        u.klo_from_n = 1; u.kz = sp.kz; u.ktot = T * (TILE / KC);
        CHK(launch_gemm_nt(g, u, sp.nsl));
        hipLaunchKernelGGL(k_split_combine_u, dim3((unsigned)R), dim3(256), 0, g->stream, g->dsplit, Rc * ld, ld, sp.kz, sp.nsl, N,
                           g->dUT, ld);
        HIPCHK(hipGetLastError());
        t_end(g);
    }
    return 0;
}

// best_off: added to the winner's index (a shard of a larger candidate set reports GLOBAL columns)
static int score_core(bohip_gp* g, int acq_id, const double* acq_params, const double* dXs, int64_t R, double* d_mu,
                      double* d_var, double* d_score, Best* d_best, int64_t best_off = 0) {
    if (g->n == 0) return fail(BOHIP_E_STATE, "model has no observations");
    CHK(ensure_fresh(g));
    CHK(ensure_score_scratch(g, R));
    AcqParams ap{acq_id, 0.0, 0.0};
    if (acq_params) {
        if (acq_id != BOHIP_ACQ_MAXMEAN) ap.p0 = acq_params[0];
        if (acq_id == BOHIP_ACQ_MI) ap.p1 = acq_params[1];
    } else if (acq_id != BOHIP_ACQ_MAXMEAN) {
        return fail(BOHIP_E_ARG, "acq_params required for this acquisition");
    }
    if (path_R(g, R) <= small_limit(g) && R <= SMALL_MAX) {   // row-wise posterior, scoring and arg-max fused into its finish kernel
        CHK(one_time_kernel_setup());
        return small_posterior(g, dXs, 0, R, false, ap, d_mu, d_var, d_score, d_best, best_off);
    }
    const SplitPlan sp = split_plan(g, R);
    if (sp.nsl > 0) {
        CHK(split_posterior(g, dXs, R, sp, false));
    } else if (g_fuse_finish) {
        // whole-K jobs: scoring and arg-max ride in k_trigemm_sq's epilogue (the workgroup that completes a candidate tile
        // finishes it; the one that completes the last tile writes the record): no k_score / k_argmax_final launches
        FuseParams fz{};
        fz.tile_cnt = g->dfz_cnt; fz.total_cnt = g->dfz_cnt + g->fz_cap; fz.tile_best = g->dfz_best;
        fz.tiles_total = (int)((R + CTILE - 1) / CTILE); fz.R_total = R;
        fz.sigma2 = std::exp(2.0 * g->logsig); fz.beta = g->beta; fz.ap = ap;
        fz.mu_out = d_mu; fz.var_out = d_var; fz.score_out = d_score; fz.best_out = d_best; fz.best_off = best_off;
        // The fused finish counts arrivals in counters its last waves leave at zero.  If an earlier call failed between two of
        // its launches they are not: every later call would then miss its "last arrival" and write no result.  So a call that
        // did not enqueue all of its launches marks the counters dirty, and the next one clears them first.
        if (g->fz_dirty) HIPCHK(hipMemsetAsync(g->dfz_cnt, 0, (size_t)(g->fz_cap + 1 + 16) * sizeof(unsigned), g->stream));
        g->fz_dirty = true;
        const int rc = posterior_pass(g, dXs, R, fz);
        if (rc == 0) g->fz_dirty = false;
        return rc;
    } else {
        CHK(posterior_pass(g, dXs, R));
    }
    const int64_t Npad = round_up(g->n + 1, TILE), Rpad = round_up(R, TILE) + TILE;
    const int T = (int)(Npad / TILE);
    const int nb = (int)((R + 255) / 256);
    t_begin(g, "score");
    hipLaunchKernelGGL(k_score, dim3(nb), dim3(256), 0, g->stream, g->dq, Rpad, g->q_tiles, g->dmu_raw, R,
                       std::exp(2.0 * g->logsig), g->beta, ap, d_mu, d_var, d_score, d_best ? g->dblock_best : nullptr);
    if (d_best) hipLaunchKernelGGL(k_argmax_final, dim3(1), dim3(256), 0, g->stream, g->dblock_best, nb, d_best, (long long)best_off);
    HIPCHK(hipGetLastError());
    t_end(g);
    return 0;
}

template <int DT>
static void launch_grad(bohip_gp* g, const double* dXs, int64_t r0, int64_t r1, const KernelHyper& hp, const AcqParams& ap,
                        double* d_grad, const double* UT, int S, const GradQ& gq) {
    if constexpr (DT <= 16) {   // large batches: 4 candidates per workgroup share the observation stream
        if (r1 - r0 > SMALL_MAX) {
            hipLaunchKernelGGL(k_grad_finish_tiled<DT>, dim3((unsigned)((r1 - r0 + GC - 1) / GC)), dim3(256), 0, g->stream, g->dX, g->n,
                               dXs, r0, r1, hp, g->dalpha, UT, g->ld, g->dmu, g->dvar, ap, d_grad);
            return;
        }
    }
    hipLaunchKernelGGL(k_grad_finish<DT>, dim3((unsigned)(r1 - r0), (unsigned)S), dim3(256), 0, g->stream, g->dX, g->n, dXs, r0, r1,
                       hp, g->dalpha, UT, g->ld, g->dmu, g->dvar, ap, d_grad, g->dgparts, g->dgcount, gq);
}
static int launch_grad_any(bohip_gp* g, const double* dXs, int64_t r0, int64_t r1, const KernelHyper& hp, const AcqParams& ap,
                           double* d_grad, const double* UT, const GradQ& gq = GradQ{}) {
    // small batches: split the observations over S workgroups per candidate (see k_grad_finish)
    int S = 1;
    // (one 256-observation stride per workgroup -- (n + 255) / 256 splits -- was measured in round 4: 2 us of a 58 us pass, and the changed
    // summation order moved the ascent's end points enough to graze the SciPy check's KKT bound at N = 600: not kept)
    if (r1 - r0 <= SMALL_MAX) S = (int)std::min<int64_t>(16, std::max<int64_t>(1, g->n / 768));
    if (S > 1) CHK(ensure_small_counters(g));
    if (g->d <= 2) launch_grad<2>(g, dXs, r0, r1, hp, ap, d_grad, UT, S, gq);
    else if (g->d <= 4) launch_grad<4>(g, dXs, r0, r1, hp, ap, d_grad, UT, S, gq);
    else if (g->d <= 8) launch_grad<8>(g, dXs, r0, r1, hp, ap, d_grad, UT, S, gq);
    else if (g->d <= 16) launch_grad<16>(g, dXs, r0, r1, hp, ap, d_grad, UT, S, gq);
    else if (g->d <= 32) launch_grad<32>(g, dXs, r0, r1, hp, ap, d_grad, UT, S, gq);
    else launch_grad<64>(g, dXs, r0, r1, hp, ap, d_grad, UT, S, gq);
    HIPCHK(hipGetLastError());
    return 0;
}

// A8: scores + gradients for R candidates (device pointers).  Per chunk: K*' -> V' (trigemm, V stored)
// -> U' = V' W (k_gemm) -> mu, sigma^2, score -> gradient.
static int score_grad_core(bohip_gp* g, int acq_id, const double* acq_params, const double* dXs, int64_t R, double* d_score,
                           double* d_grad) {
    if (g->n == 0) return fail(BOHIP_E_STATE, "model has no observations");
    CHK(ensure_fresh(g));
    CHK(ensure_score_scratch(g, R));
    CHK(one_time_kernel_setup());
    AcqParams ap{acq_id, 0.0, 0.0};
    if (acq_params) {
        if (acq_id != BOHIP_ACQ_MAXMEAN) ap.p0 = acq_params[0];
        if (acq_id == BOHIP_ACQ_MI) ap.p1 = acq_params[1];
    } else if (acq_id != BOHIP_ACQ_MAXMEAN) {
        return fail(BOHIP_E_ARG, "acq_params required for this acquisition");
    }
    const int64_t N = g->n, Npad = round_up(N + 1, TILE), Rpad = round_up(R, TILE) + TILE;  // +TILE: head-room for tile-granular writes
    const int T = (int)(Npad / TILE);
    const KernelHyper hp = make_hyper(g);
    if (path_R(g, R) <= small_limit(g) && R <= SMALL_MAX) {  // the reference's default: a handful of L-BFGS restarts per call
        // round 5: two MFMA kernels (kernels_small.hip): K*' + V' + posterior finish, U' + gradient
        if (g_small_mfma) return small_pass_mfma(g, dXs, R, ap, g->dmu, g->dvar, d_score, nullptr, 0, d_grad);
        // K*' -> V' = K*' W' rows (row-wise) -> U' = V' W rows -> ONE finishing kernel: q, mu, sigma^2, value and gradient
        CHK(ensure_small_counters(g));
        t_begin(g, "kstar");
        CHK(launch_kstar_any(g, dXs, 0, R, Npad, hp));
        t_end(g);
        t_begin(g, "small_V");
        CHK(launch_rows_trimv(g, g->dW, N + 1, g->dKsT, (int)R, g->dApp, 0));
        t_end(g);
        g->q_tiles = 0;
        t_begin(g, "small_U");
        CHK(launch_rows_trimv(g, g->dWT, N, g->dApp, (int)R, g->dApp + (int64_t)APP_UT_ROW0 * g->ld, 1));
        t_end(g);
        t_begin(g, "grad");
        GradQ gq{g->dApp, g->ld, std::exp(2.0 * g->logsig), g->beta, g->dmu, g->dvar, d_score, g->asc_go};
        CHK(launch_grad_any(g, dXs, 0, R, hp, ap, d_grad, g->dApp + (int64_t)APP_UT_ROW0 * g->ld, gq));
        t_end(g);
        return 0;
    }
    const SplitPlan sp = split_plan(g, R);
    if (sp.nsl > 0) {
        CHK(split_posterior(g, dXs, R, sp, true));
        t_begin(g, "score+grad");
        hipLaunchKernelGGL(k_score, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, g->stream, g->dq, Rpad, 0, g->dmu_raw, R,
                           std::exp(2.0 * g->logsig), g->beta, ap, g->dmu, g->dvar, d_score, (Best*)nullptr);
        CHK(launch_grad_any(g, dXs, 0, R, hp, ap, d_grad, g->dUT));
        t_end(g);
        return 0;
    }
    CHK(ensure_grad_scratch(g));
    const int64_t rc = g->chunk_now;
    for (int64_t r0 = 0; r0 < R; r0 += rc) {
        const int64_t r1 = std::min(R, r0 + rc);
        t_begin(g, "kstar");
        CHK(launch_kstar_any(g, dXs, r0, r1, Npad, hp));
        t_end(g);
        const int CT = (int)((r1 - r0 + TILE - 1) / TILE);
        t_begin(g, "trigemm_sq+V");
        CHK(launch_trigemm(g, T, r1 - r0, N, Rpad, r0, g->dVT));
        t_end(g);
        t_begin(g, "gemm_U");
        GemmNTParams p{};  // U'[r][j] = sum_i V'[r][i] W'[j][i]   (B = W' upper-triangular: i >= j)
        p.A = g->dVT; p.lda = g->ld; p.B = g->dWT; p.ldb = g->ld; p.C = g->dUT; p.ldc = g->ld;
        p.mt = CT; p.nt64 = 2 * T; p.kc = T * (TILE / KC); p.alpha = 1.0; p.beta = 0.0; p.klo_from_n = 1;
        CHK(launch_gemm_nt(g, p));
        t_end(g);
        t_begin(g, "score+grad");
        const int nb = (int)((r1 - r0 + 255) / 256);
        hipLaunchKernelGGL(k_score, dim3(nb), dim3(256), 0, g->stream, g->dq + r0, Rpad, T, g->dmu_raw + r0, r1 - r0,
                           std::exp(2.0 * g->logsig), g->beta, ap, g->dmu + r0, g->dvar + r0, d_score + r0, (Best*)nullptr);
        CHK(launch_grad_any(g, dXs, r0, r1, hp, ap, d_grad, g->dUT));
        t_end(g);
    }
    return 0;
}

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

const char* bohip_last_error(void) { return g_err.c_str(); }
const char* bohip_version(void) { return "bohip 0.1 (gfx950, fp64 mfma 4x4x4)"; }
int bohip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bohip_gp_create(int64_t d, int64_t capacity, int kernel_id, int device, bohip_gp** out) {
    if (!out) return fail(BOHIP_E_ARG, "out is null");
    *out = nullptr;
    if (d < 1 || d > DMAX) return fail(BOHIP_E_ARG, "d must be in [1, 64]");
    if (capacity < 1) capacity = 1;
    if (kernel_id < 0 || kernel_id > 2) return fail(BOHIP_E_ARG, "unknown kernel_id");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(BOHIP_E_NODEVICE, "no HIP device visible; libbohip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(BOHIP_E_ARG, "device ordinal out of range");
    HIPCHK(hipSetDevice(device));
    bohip_gp* g = new bohip_gp();
    g->device = device;
    g->d = (int)d;
    g->kern = kernel_id;
    for (int k = 0; k < DMAX; ++k) g->loglen[k] = 0.0;
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    hipError_t e = hipStreamCreateWithPriority(&g->own_stream, hipStreamNonBlocking, prio_hi);  // critical path of the factorisation
    // (reserving CUs for the critical stream with a CU mask on the side stream was measured and does not help)
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&g->side_stream, hipStreamNonBlocking, prio_lo);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_panels, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_bulk, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&g->inv_stream, hipStreamNonBlocking, prio_lo);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&g->col_stream, hipStreamNonBlocking, prio_hi);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_blk, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_inv, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_gate, hipEventDisableTiming);
    if (e != hipSuccess) { delete g; return fail(BOHIP_E_HIP, hipGetErrorString(e)); }
    g->stream = g->own_stream;
    if (hipHostMalloc((void**)&g->hpin, (size_t)(3 * SMALL_R + 2 + 2 * SMALL_R * DMAX) * 8, hipHostMallocDefault) != hipSuccess) g->hpin = nullptr;
    if (hipMalloc(&g->dinfo, sizeof(int)) != hipSuccess || hipMalloc(&g->dmll, 8 * (DMAX + 4)) != hipSuccess ||
        hipMalloc(&g->dbest, 4096 * sizeof(Best)) != hipSuccess) {
        delete g;
        return fail(BOHIP_E_HIP, "hipMalloc failed");
    }
    int rc = alloc_model(g, capacity);
    if (rc != 0) { bohip_gp_destroy(g); return rc; }
    if (const char* e = getenv("BOHIP_JITTER")) {   // "rel[,tries]": jitter escalation for every handle of the process (default: off)
        double rel = 0.0;
        int tries = 10;
        if (sscanf(e, "%lf,%d", &rel, &tries) >= 1 && rel > 0.0) { g->jitter_rel = rel; g->jitter_tries = std::min(32, std::max(1, tries)); }
    }
    *out = g;
    return 0;
}

int bohip_gp_comm_destroy(bohip_gp* g);
void bohip_gp_destroy(bohip_gp* g) {
    if (!g) return;
    hipSetDevice(g->device);
    if (g->own_stream) hipStreamSynchronize(g->own_stream);
    if (g->comm) bohip_gp_comm_destroy(g);
    free_model(g);
    for (double** p : {&g->dKsT, &g->dq, &g->dmu_raw, &g->dXs, &g->dmu, &g->dvar, &g->dscore, &g->dmll, &g->dVT, &g->dUT})
        if (*p) hipFree(*p);
    if (g->dex_tasks) hipFree(g->dex_tasks);
    if (g->dblock_best) hipFree(g->dblock_best);
    if (g->dbest) hipFree(g->dbest);
    if (g->dfz_cnt) hipFree(g->dfz_cnt);
    if (g->dfz_best) hipFree(g->dfz_best);
    if (g->dpieces) hipFree(g->dpieces);
    if (g->dclk) hipFree(g->dclk);
    if (g->dgrad) hipFree(g->dgrad);
    if (g->dgparts) hipFree(g->dgparts);
    if (g->dsplit) hipFree(g->dsplit);
    if (g->dgcount) hipFree(g->dgcount);
    if (g->dsm) hipFree(g->dsm);
    if (g->dsm_cnt) hipFree(g->dsm_cnt);
    if (g->hpin) hipHostFree(g->hpin);
    if (g->dschur) hipFree(g->dschur);
    if (g->asc_block) hipFree(g->asc_block);
    if (g->asc_ints) hipFree(g->asc_ints);
    if (g->asc_hints) hipHostFree(g->asc_hints);
    if (g->asc_hio) hipHostFree(g->asc_hio);
    if (g->asc_dio) hipFree(g->asc_dio);
    if (g->asc_bounds) hipFree(g->asc_bounds);
    if (g->asc_best) hipFree(g->asc_best);
    if (g->dthompson) hipFree(g->dthompson);
    if (g->ddmll_parts) hipFree(g->ddmll_parts);
    if (g->dVV) hipFree(g->dVV);
    if (g->dcov) hipFree(g->dcov);
    if (g->dinfo) hipFree(g->dinfo);
    for (auto& e : g->tpool) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    if (g->side_stream) { hipStreamSynchronize(g->side_stream); hipStreamDestroy(g->side_stream); }
    if (g->ev_panels) hipEventDestroy(g->ev_panels);
    if (g->ev_bulk) hipEventDestroy(g->ev_bulk);
    if (g->inv_stream) { hipStreamSynchronize(g->inv_stream); hipStreamDestroy(g->inv_stream); }
    if (g->col_stream) { hipStreamSynchronize(g->col_stream); hipStreamDestroy(g->col_stream); }
    for (hipEvent_t ev : g->ev_tier) hipEventDestroy(ev);
    if (g->ev_blk) hipEventDestroy(g->ev_blk);
    if (g->ev_inv) hipEventDestroy(g->ev_inv);
    if (g->ev_gate) hipEventDestroy(g->ev_gate);
    if (g->own_stream) hipStreamDestroy(g->own_stream);
    delete g;
}

int bohip_gp_set_hyper(bohip_gp* g, const double* loglen, double logsig, double lognoise, double mean_const) {
    if (!g || !loglen) return fail(BOHIP_E_ARG, "null argument");
    const int nl = g->kern == KERN_SEISO ? 1 : g->d;
    for (int k = 0; k < nl; ++k) g->loglen[k] = loglen[k];
    if (g->kern == KERN_SEISO)
        for (int k = 1; k < g->d; ++k) g->loglen[k] = loglen[0];
    g->logsig = logsig;
    g->lognoise = lognoise;
    g->beta = mean_const;
    g->stale = true;
    return 0;
}

int bohip_gp_append(bohip_gp* g, const double* X, const double* y, int64_t p) {
    if (!g || p < 0 || (p > 0 && (!X || !y))) return fail(BOHIP_E_ARG, "bad arguments");
    HIPCHK(hipSetDevice(g->device));
    t_reset(g);
    int rc = 0;
    bool done = false;
    if (p > 0) {
        const int64_t n_old = g->n, n_new = g->n + p;
        if (n_new > g->cap) {
            // growth: alloc_model re-uploads everything from the host mirrors, so they are extended first -- and rolled
            // back if the allocation fails (the handle then holds its old observations and no device buffers; the next
            // call that needs them allocates again, see ensure_fresh)
            g->hX.insert(g->hX.end(), X, X + p * g->d);
            g->hy.insert(g->hy.end(), y, y + p);
            g->n = n_new;
            const int arc = alloc_model(g, std::max<int64_t>(2 * g->cap, n_new));
            if (arc != 0) {
                g->hX.resize((size_t)n_old * g->d);
                g->hy.resize((size_t)n_old);
                g->n = n_old;
                free_model(g);
                g->cap = 0;
                g->stale = true;
                return arc;
            }
        } else {
            // A few rows (what a BO iteration appends) go through the pinned block: the copies are then plain DMA commands in front of the
            // update's kernels, the pivot word comes back the same way, and the call synchronises ONCE (three times before: behind the
            // pageable copies, for the pivot word, at the end -- ~60 us of a 0.19 ms append).  Otherwise: the mirrors and n advance only
            // after the device holds the new rows, a failed copy leaves the handle unchanged.
            const bool staged = g->hpin && p <= SMALL_R && (size_t)p * g->d <= (size_t)SMALL_R * DMAX;
            int* pinfo = nullptr;
            if (staged) {
                double* sx = g->hpin + 3 * SMALL_R + 2;   // (the gradient area and the score area of the result block: idle during an append)
                double* sy = g->hpin;
                std::memcpy(sx, X, (size_t)p * g->d * 8);
                std::memcpy(sy, y, (size_t)p * 8);
                HIPCHK(hipMemcpyAsync(g->dX + n_old * g->d, sx, (size_t)p * g->d * 8, hipMemcpyHostToDevice, g->stream));
                HIPCHK(hipMemcpyAsync(g->dy + n_old, sy, (size_t)p * 8, hipMemcpyHostToDevice, g->stream));
                pinfo = reinterpret_cast<int*>(g->hpin + 3 * SMALL_R);
                *pinfo = 0;
            } else {
                HIPCHK(hipMemcpyAsync(g->dX + n_old * g->d, X, (size_t)p * g->d * 8, hipMemcpyHostToDevice, g->stream));
                HIPCHK(hipMemcpyAsync(g->dy + n_old, y, (size_t)p * 8, hipMemcpyHostToDevice, g->stream));
                HIPCHK(hipStreamSynchronize(g->stream));  // caller's buffers are only valid during the call
            }
            g->hX.insert(g->hX.end(), X, X + p * g->d);
            g->hy.insert(g->hy.end(), y, y + p);
            g->n = n_new;
            if (!g->stale && g->n_factored == n_old && n_old > 0 && p <= APPEND_PMAX) {
                rc = append_incremental(g, n_old, p, pinfo);
                if (rc != 0) { g->stale = true; g->mirror_dirty = staged; }   // (staged: the rows' copies were only queued -- the device may not hold them)
                done = true;
                if (rc == 0 && pinfo) {
                    const hipError_t e = hipStreamSynchronize(g->stream);
                    if (e != hipSuccess) { g->stale = true; g->mirror_dirty = true; return fail(BOHIP_E_HIP, hipGetErrorString(e)); }
                    t_collect(g);
                    if (*pinfo != 0) {
                        g->pivot = *pinfo;
                        g->stale = true;
                        return fail(BOHIP_E_NOTPD, "kernel matrix not positive definite at pivot " + std::to_string(*pinfo));
                    }
                    g->pivot = 0;
                    return 0;
                }
            }
        }
    }
    if (!done) rc = ensure_fresh(g);
    HIPCHK(hipStreamSynchronize(g->stream));
    t_collect(g);
    return rc;
}

int bohip_gp_refit(bohip_gp* g) {
    if (!g) return fail(BOHIP_E_ARG, "null handle");
    HIPCHK(hipSetDevice(g->device));
    t_reset(g);
    int rc = refit(g);
    HIPCHK(hipStreamSynchronize(g->stream));
    t_collect(g);
    return rc;
}

int bohip_gp_dims(const bohip_gp* g, int64_t* d, int64_t* n) {
    if (!g) return fail(BOHIP_E_ARG, "null handle");
    if (d) *d = g->d;
    if (n) *n = g->n;
    return 0;
}
int bohip_gp_maxy(const bohip_gp* g, double* maxy) {
    if (!g || !maxy) return fail(BOHIP_E_ARG, "null argument");
    double m = -INFINITY;
    for (double v : g->hy) if (v > m) m = v;
    *maxy = m;
    return 0;
}
int bohip_gp_get_xy(const bohip_gp* g, double* X, double* y) {
    if (!g) return fail(BOHIP_E_ARG, "null handle");
    if (X && g->n) std::memcpy(X, g->hX.data(), (size_t)g->n * g->d * 8);
    if (y && g->n) std::memcpy(y, g->hy.data(), (size_t)g->n * 8);
    return 0;
}

int bohip_gp_mll(bohip_gp* g, double* mll) {
    if (!g || !mll) return fail(BOHIP_E_ARG, "null argument");
    HIPCHK(hipSetDevice(g->device));
    if (g->n == 0) { *mll = 0.0; return 0; }
    CHK(ensure_fresh(g));
    hipLaunchKernelGGL(k_mll, dim3(1), dim3(256), 0, g->stream, g->dL, g->ld, g->n, g->dr, g->dalpha, g->dmll);
    HIPCHK(hipGetLastError());
    double* const pm = g->hpin ? g->hpin : mll;     // (through the pinned block: a plain DMA command)
    HIPCHK(hipMemcpyAsync(pm, g->dmll, 8, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    *mll = *pm;
    return 0;
}

int bohip_gp_mll_grad(bohip_gp* g, double* mll, double* d_lognoise, double* d_mean, double* d_kern) {
    if (!g || !mll || !d_lognoise || !d_mean || !d_kern) return fail(BOHIP_E_ARG, "null argument");
    HIPCHK(hipSetDevice(g->device));
    const int iso = g->kern == KERN_SEISO, nl = iso ? 1 : g->d;
    if (g->n == 0) {
        *mll = 0.0; *d_lognoise = 0.0; *d_mean = 0.0;
        for (int k = 0; k <= nl; ++k) d_kern[k] = 0.0;
        return 0;
    }
    CHK(ensure_fresh(g));
    const int64_t N = g->n, Npad = round_up(N + 1, TILE), ld = g->ld;
    const int T = (int)(Npad / TILE), NP = g->d + 3;
    hipLaunchKernelGGL(k_mll, dim3(1), dim3(256), 0, g->stream, g->dL, ld, N, g->dr, g->dalpha, g->dmll);
    t_begin(g, "kinv");
    {
        GemmNTParams p{};  // cK^-1 = W'W, lower 128-tiles only:  (W'W)_ij = sum_{k >= max(i,j)} W'[i][k] W'[j][k]
        p.A = g->dWT; p.lda = ld; p.B = g->dWT; p.ldb = ld; p.C = g->dS; p.ldc = ld;
        p.mt = T; p.nt64 = 2 * T; p.kc = T * (TILE / KC); p.alpha = 1.0; p.beta = 0.0;
        p.klo_from_m = 1; p.klo_from_n = 1; p.diag_skip = 1; p.row0 = 0; p.col0 = 0;
        CHK(launch_gemm_nt(g, p));
    }
    t_end(g);
    t_begin(g, "dmll_reduce");
    const int rpb = 32;
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)((N + rpb - 1) / rpb));
    const int64_t nblocks = (int64_t)grid.x * grid.y;
    if (g->dmll_cap < nblocks * NP) {
        if (g->ddmll_parts) HIPCHK(hipFree(g->ddmll_parts));
        g->ddmll_parts = nullptr; g->dmll_cap = 0;
        HIPCHK(hipMalloc(&g->ddmll_parts, (size_t)nblocks * NP * 8));
        g->dmll_cap = nblocks * NP;
    }
    const KernelHyper hp = make_hyper(g);
    const double noise_var = std::exp(2.0 * g->lognoise);
#define DM(DTV) hipLaunchKernelGGL(k_dmll_parts<DTV>, grid, dim3(256), 0, g->stream, g->dX, N, hp, noise_var, g->dS, ld, \
                                   g->dalpha, rpb, g->ddmll_parts)
    if (g->d <= 2) DM(2); else if (g->d <= 4) DM(4); else if (g->d <= 8) DM(8); else if (g->d <= 16) DM(16);
    else if (g->d <= 32) DM(32); else DM(64);
#undef DM
    const int nout = nl + 3;
    hipLaunchKernelGGL(k_dmll_final, dim3(nout), dim3(256), 0, g->stream, g->ddmll_parts, nblocks, NP, g->d, iso, g->dmll + 1);
    HIPCHK(hipGetLastError());
    t_end(g);
    double hbuf[DMAX + 4];
    double* const h = g->hpin ? g->hpin + 3 * SMALL_R + 2 : hbuf;     // (the pinned block's gradient area: a plain DMA command)
    HIPCHK(hipMemcpyAsync(h, g->dmll, 8 * (size_t)(nout + 1), hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    *mll = h[0]; *d_lognoise = h[1]; *d_mean = h[2];
    for (int k = 0; k <= nl; ++k) d_kern[k] = h[3 + k];
    return 0;
}

int bohip_gp_predict_dev(bohip_gp* g, const double* dXs, int64_t R, double* d_mu, double* d_var) {
    if (!g || !dXs || R < 0) return fail(BOHIP_E_ARG, "bad arguments");
    if (R == 0) return 0;
    HIPCHK(hipSetDevice(g->device));
    return score_core(g, BOHIP_ACQ_MAXMEAN, nullptr, dXs, R, d_mu, d_var, nullptr, nullptr);
}

int bohip_gp_score_dev(bohip_gp* g, int acq_id, const double* acq_params, const double* dXs, int64_t R,
                       double* d_score, bohip_best* d_best) {
    if (!g || !dXs || R < 0) return fail(BOHIP_E_ARG, "bad arguments");
    if (acq_id < 0 || acq_id > BOHIP_ACQ_MAXMEAN) return fail(BOHIP_E_ARG, "unknown acq_id");
    HIPCHK(hipSetDevice(g->device));
    t_reset(g);
    if (R == 0) {
        if (d_best) {
            Best b{-INFINITY, -1};
            HIPCHK(hipMemcpyAsync(d_best, &b, sizeof(b), hipMemcpyHostToDevice, g->stream));
            HIPCHK(hipStreamSynchronize(g->stream));
        }
        return 0;
    }
    return score_core(g, acq_id, acq_params, dXs, R, nullptr, nullptr, d_score, reinterpret_cast<Best*>(d_best));
}

// Candidates of a small host call (R <= SMALL_R): the kernels read them straight from the pinned, device-visible block -- no copy
// command in front of the scoring pass (measured, same box, alternating: bohip_gp_predict 46.6 -> 45.2 us, bohip_gp_score_grad 63.5 -> 59.8 us at
// N = 3000; polling hipStreamQuery before the blocking synchronisation: no difference beyond the 43-51 us run-to-run spread).
// Every caller synchronises the stream before it returns, so the block is free again at the next call.
static const double* small_candidates_in_place(bohip_gp* g, const double* Xs, int64_t R) {
    if (!g->hpin || R > SMALL_R || g->d > DMAX || !g_small_zero_copy) return nullptr;
    double* sx = g->hpin + 3 * SMALL_R + 2 + SMALL_R * DMAX;
    std::memcpy(sx, Xs, (size_t)R * g->d * 8);
    return sx;
}

int bohip_gp_predict(bohip_gp* g, const double* Xs, int64_t R, double* mu, double* var) {
    if (!g || R < 0 || (R > 0 && (!Xs || !mu || !var))) return fail(BOHIP_E_ARG, "bad arguments");
    if (R == 0) return 0;
    HIPCHK(hipSetDevice(g->device));
    t_reset(g);
    CHK(ensure_xs(g, R));
    CHK(ensure_score_scratch(g, R));
    const double* xin = small_candidates_in_place(g, Xs, R);
    if (!xin) HIPCHK(hipMemcpyAsync(g->dXs, Xs, (size_t)R * g->d * 8, hipMemcpyHostToDevice, g->stream));
    if (R <= SMALL_R && g->hpin) {   // results land in pinned host memory, no copy commands
        double *pmu = g->hpin + SMALL_R, *pvar = g->hpin + 2 * SMALL_R;
        CHK(score_core(g, BOHIP_ACQ_MAXMEAN, nullptr, xin ? xin : g->dXs, R, pmu, pvar, nullptr, nullptr));
        HIPCHK(hipStreamSynchronize(g->stream));
        std::memcpy(mu, pmu, (size_t)R * 8);
        std::memcpy(var, pvar, (size_t)R * 8);
        t_collect(g);
        return 0;
    }
    CHK(score_core(g, BOHIP_ACQ_MAXMEAN, nullptr, g->dXs, R, g->dmu, g->dvar, nullptr, nullptr));
    HIPCHK(hipMemcpyAsync(mu, g->dmu, (size_t)R * 8, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipMemcpyAsync(var, g->dvar, (size_t)R * 8, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    t_collect(g);
    return 0;
}

int bohip_gp_predict_cov(bohip_gp* g, const double* Xs, int64_t R, double* mu, double* cov) {
    if (!g || R < 0 || (R > 0 && (!Xs || !mu || !cov))) return fail(BOHIP_E_ARG, "bad arguments");
    if (R == 0) return 0;
    if (g->n == 0) return fail(BOHIP_E_STATE, "model has no observations");
    HIPCHK(hipSetDevice(g->device));
    t_reset(g);
    CHK(ensure_fresh(g));
    CHK(ensure_xs(g, R));
    CHK(ensure_score_scratch(g, R));
    if (R > g->chunk_now)
        return fail(BOHIP_E_UNSUPPORTED, "predict_cov: R exceeds one candidate chunk (" + std::to_string(chunk_cap(g)) + ")");
    CHK(ensure_grad_scratch(g));
    CHK(one_time_kernel_setup());
    const int64_t N = g->n, Npad = round_up(N + 1, TILE), Rpad = round_up(R, TILE) + TILE, Rp = round_up(R, TILE);
    const int T = (int)(Npad / TILE), CT = (int)(Rp / TILE);
    if (g->cov_cap < Rp) {
        for (double** p : {&g->dVV, &g->dcov})
            if (*p) { HIPCHK(hipFree(*p)); *p = nullptr; }
        g->cov_cap = 0;
        HIPCHK(hipMalloc(&g->dVV, (size_t)Rp * Rp * 8));
        HIPCHK(hipMalloc(&g->dcov, (size_t)Rp * Rp * 8));
        g->cov_cap = Rp;
    }
    const KernelHyper hp = make_hyper(g);
    HIPCHK(hipMemcpyAsync(g->dXs, Xs, (size_t)R * g->d * 8, hipMemcpyHostToDevice, g->stream));
    g->q_tiles = T;
    t_begin(g, "kstar");
    CHK(launch_kstar_any(g, g->dXs, 0, R, Npad, hp));
    t_end(g);
    t_begin(g, "trigemm_sq+V");
    CHK(launch_trigemm(g, T, R, N, Rpad, 0, g->dVT));
    t_end(g);
    t_begin(g, "gemm_VV");
    GemmNTParams p{};  // (V'V)[r][s] = sum_i V'[r][i] V'[s][i], lower 128-tiles
    p.A = g->dVT; p.lda = g->ld; p.B = g->dVT; p.ldb = g->ld; p.C = g->dVV; p.ldc = g->cov_cap;
    p.mt = CT; p.nt64 = 2 * CT; p.kc = T * (TILE / KC); p.alpha = 1.0; p.beta = 0.0; p.diag_skip = 1;
    CHK(launch_gemm_nt(g, p));
    t_end(g);
    t_begin(g, "post_cov");
    AcqParams ap{BOHIP_ACQ_MAXMEAN, 0.0, 0.0};
    hipLaunchKernelGGL(k_score, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, g->stream, g->dq, Rpad, T, g->dmu_raw, R,
                       std::exp(2.0 * g->logsig), g->beta, ap, g->dmu, g->dvar, (double*)nullptr, (Best*)nullptr);
    dim3 grid((unsigned)((R + 255) / 256), (unsigned)((R + 15) / 16));
#define PC(DTV) hipLaunchKernelGGL(k_post_cov<DTV>, grid, dim3(256), 0, g->stream, g->dXs, R, hp, g->dVV, g->cov_cap, g->dcov, R)
    if (g->d <= 2) PC(2); else if (g->d <= 4) PC(4); else if (g->d <= 8) PC(8); else if (g->d <= 16) PC(16);
    else if (g->d <= 32) PC(32); else PC(64);
#undef PC
    HIPCHK(hipGetLastError());
    t_end(g);
    HIPCHK(hipMemcpyAsync(mu, g->dmu, (size_t)R * 8, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipMemcpyAsync(cov, g->dcov, (size_t)R * R * 8, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    t_collect(g);
    return 0;
}

int bohip_gp_score(bohip_gp* g, int acq_id, const double* acq_params, const double* Xs, int64_t R, double* score,
                   bohip_best* best) {
    if (!g || R < 0 || (R > 0 && !Xs)) return fail(BOHIP_E_ARG, "bad arguments");
    if (acq_id < 0 || acq_id > BOHIP_ACQ_MAXMEAN) return fail(BOHIP_E_ARG, "unknown acq_id");
    if (R == 0) {
        if (best) { best->val = -INFINITY; best->idx = -1; }
        return 0;
    }
    HIPCHK(hipSetDevice(g->device));
    t_reset(g);
    CHK(ensure_xs(g, R));
    CHK(ensure_score_scratch(g, R));
    const double* xin = small_candidates_in_place(g, Xs, R);
    if (!xin) HIPCHK(hipMemcpyAsync(g->dXs, Xs, (size_t)R * g->d * 8, hipMemcpyHostToDevice, g->stream));
    if (R <= SMALL_R && g->hpin) {
        double* pscore = g->hpin;
        Best* pbest = reinterpret_cast<Best*>(g->hpin + 3 * SMALL_R);
        CHK(score_core(g, acq_id, acq_params, xin ? xin : g->dXs, R, nullptr, nullptr, score ? pscore : nullptr, best ? pbest : nullptr));
        HIPCHK(hipStreamSynchronize(g->stream));
        if (score) std::memcpy(score, pscore, (size_t)R * 8);
        if (best) std::memcpy(best, pbest, sizeof(Best));
        t_collect(g);
        return 0;
    }
    CHK(score_core(g, acq_id, acq_params, g->dXs, R, nullptr, nullptr, score ? g->dscore : nullptr,
                   best ? g->dbest : nullptr));
    if (score) HIPCHK(hipMemcpyAsync(score, g->dscore, (size_t)R * 8, hipMemcpyDeviceToHost, g->stream));
    if (best) HIPCHK(hipMemcpyAsync(best, g->dbest, sizeof(Best), hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    t_collect(g);
    return 0;
}

int bohip_gp_score_grad(bohip_gp* g, int acq_id, const double* acq_params, const double* Xs, int64_t R, double* score,
                        double* grad) {
    if (!g || R < 0 || (R > 0 && (!Xs || !score || !grad))) return fail(BOHIP_E_ARG, "bad arguments");
    if (acq_id < 0 || acq_id > BOHIP_ACQ_MAXMEAN) return fail(BOHIP_E_ARG, "unknown acq_id");
    if (R == 0) return 0;
    HIPCHK(hipSetDevice(g->device));
    t_reset(g);
    CHK(ensure_xs(g, R));
    if (g->grad_cap < R * g->d) {
        if (g->dgrad) hipFree(g->dgrad);
        g->dgrad = nullptr;
        HIPCHK(hipMalloc(&g->dgrad, (size_t)R * g->d * 8));
        g->grad_cap = R * g->d;
    }
    double* dgrad = g->dgrad;
    const double* xin = small_candidates_in_place(g, Xs, R);
    if (!xin) HIPCHK(hipMemcpyAsync(g->dXs, Xs, (size_t)R * g->d * 8, hipMemcpyHostToDevice, g->stream));
    int rc = ensure_score_scratch(g, R);
    if (rc == 0 && R <= SMALL_R && g->hpin) {
        double *pscore = g->hpin, *pgrad = g->hpin + 3 * SMALL_R + 2;
        rc = score_grad_core(g, acq_id, acq_params, xin ? xin : g->dXs, R, pscore, pgrad);
        if (rc == 0) {
            const hipError_t e = hipStreamSynchronize(g->stream);
            if (e != hipSuccess) rc = fail(BOHIP_E_HIP, hipGetErrorString(e));
        }
        if (rc == 0) {
            std::memcpy(score, pscore, (size_t)R * 8);
            std::memcpy(grad, pgrad, (size_t)R * g->d * 8);
        }
        t_collect(g);
        return rc;
    }
    if (rc == 0) rc = score_grad_core(g, acq_id, acq_params, g->dXs, R, g->dscore, dgrad);
    if (rc == 0) {
        hipError_t e = hipMemcpyAsync(score, g->dscore, (size_t)R * 8, hipMemcpyDeviceToHost, g->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(grad, dgrad, (size_t)R * g->d * 8, hipMemcpyDeviceToHost, g->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
        if (e != hipSuccess) rc = fail(BOHIP_E_HIP, hipGetErrorString(e));
    }
    t_collect(g);
    return rc;
}

static int ensure_ascent(bohip_gp* g, int64_t R) {
    if (g->asc_cap >= R && g->asc_block) return 0;
    if (g->asc_block) HIPCHK(hipFree(g->asc_block));
    if (g->asc_ints) HIPCHK(hipFree(g->asc_ints));
    if (g->asc_hints) HIPCHK(hipHostFree(g->asc_hints));
    g->asc_block = nullptr; g->asc_ints = nullptr; g->asc_hints = nullptr; g->asc_cap = 0;
    const int64_t cap = std::max<int64_t>(R, 32), d = g->d, rd = cap * d;
    HIPCHK(hipMalloc(&g->asc_block, (size_t)((9 + 2 * ASC_M) * rd + 5 * cap) * 8));
    HIPCHK(hipMalloc(&g->asc_ints, (size_t)(4 * cap + 2 * ASC_RING + 2) * sizeof(int)));
    HIPCHK(hipMemset(g->asc_ints, 0, (size_t)(4 * cap + 2 * ASC_RING + 2) * sizeof(int)));
    HIPCHK(hipHostMalloc((void**)&g->asc_hints, (size_t)(2 * cap + ASC_RING) * sizeof(int), hipHostMallocDefault));
    std::memset(g->asc_hints, 0, (size_t)(2 * cap + ASC_RING) * sizeof(int));
    if (!g->asc_bounds) HIPCHK(hipMalloc(&g->asc_bounds, (size_t)3 * DMAX * 8));
    if (!g->asc_best) HIPCHK(hipMalloc(&g->asc_best, sizeof(Best)));
    double* p = g->asc_block;
    AscentState& a = g->asc;
    a.X = p; p += rd; a.G = p; p += rd; a.Xt = p; p += rd; a.Gt = p; p += rd; a.Xn = p; p += rd; a.Gn = p; p += rd;
    a.D = p; p += rd; a.Gp = p; p += rd; a.best_X = p; p += rd;
    a.S = p; p += ASC_M * rd; a.Y = p; p += ASC_M * rd;
    a.f = p; p += cap; a.ft = p; p += cap; a.fn = p; p += cap; a.step = p; p += cap; a.best_f = p; p += cap;
    a.active = g->asc_ints; a.accepted = g->asc_ints + cap;
    a.h_accepted = g->asc_hints; a.h_active = g->asc_hints + cap;
    a.it = g->asc_ints + 2 * cap; a.bt = g->asc_ints + 3 * cap;
    a.nact = reinterpret_cast<unsigned*>(g->asc_ints + 4 * cap); a.ticket = a.nact + 2 * ASC_RING;
    a.h_cnt = g->asc_hints + 2 * cap;
    g->asc_cap = cap;
    return 0;
}

// acquire_max(acquisition, model, lb, ub, restarts) for the gradient-based methods (reference src/acquisition.jl:48-68 with
// nlopt_setup :23-35): lock-step projected L-BFGS ascent from R start columns, state resident in HBM.
int bohip_gp_acquire_max(bohip_gp* g, int acq_id, const double* acq_params, const double* lb, const double* ub,
                         const double* starts, int64_t R, int64_t maxeval, double ftol_rel, double xtol_abs, double* x_out,
                         double* f_out, bohip_best* best, double* best_x, int64_t* evals_out) {
    if (!g || !lb || !ub || R < 0 || (R > 0 && !starts)) return fail(BOHIP_E_ARG, "bad arguments");
    if (acq_id < 0 || acq_id > BOHIP_ACQ_MAXMEAN) return fail(BOHIP_E_ARG, "unknown acq_id");
    if (acq_id != BOHIP_ACQ_MAXMEAN && !acq_params) return fail(BOHIP_E_ARG, "acq_params required for this acquisition");
    if (evals_out) *evals_out = 0;
    if (R == 0) {
        if (best) { best->val = -INFINITY; best->idx = -1; }
        return 0;
    }
    if (g->n == 0) return fail(BOHIP_E_STATE, "model has no observations");
    HIPCHK(hipSetDevice(g->device));
    t_reset(g);
    const int d = g->d;
    for (int k = 0; k < d; ++k)
        if (!(lb[k] <= ub[k])) return fail(BOHIP_E_ARG, "lowerbounds must not exceed upperbounds");
    CHK(ensure_fresh(g));
    CHK(ensure_xs(g, R));
    CHK(ensure_score_scratch(g, R));
    CHK(ensure_ascent(g, R));
    AscentState& st = g->asc;
    st.ftol_abs = g->asc_ftol_abs; st.xtol_rel = g->asc_xtol_rel; st.stopval = g->asc_stopval;
    double* dlb = g->asc_bounds;
    double* dub = dlb + DMAX;
    double* dbx = dub + DMAX;
    const bool use_wg = g_asc_wg_nmax > 0 && g->n <= std::min<int64_t>(g_asc_wg_nmax, AWG_NMAX - 1) && d <= 16 && !g_asc_lockstep;
    // one pinned block in ([lb | ub | starts]), one pinned block out (k_asc_final's packed result): seven pageable copies were ~0.1 ms
    // of a call that is 1.3 ms at N = 3000 since the search needs 17 passes instead of 228
    const size_t n_in = (size_t)(2 + R) * d, n_out = (size_t)2 + d + R + (size_t)R * d + R, n_io = std::max(n_in, n_out);
    if (n_io > g->asc_io_cap) {
        if (g->asc_hio) HIPCHK(hipHostFree(g->asc_hio));
        if (g->asc_dio) HIPCHK(hipFree(g->asc_dio));
        g->asc_hio = nullptr; g->asc_dio = nullptr; g->asc_io_cap = 0;
        HIPCHK(hipHostMalloc((void**)&g->asc_hio, n_io * 2 * 8, hipHostMallocDefault));
        HIPCHK(hipMalloc(&g->asc_dio, n_io * 2 * 8));
        g->asc_io_cap = n_io * 2;
    }
    std::memcpy(g->asc_hio, lb, (size_t)d * 8);
    std::memcpy(g->asc_hio + d, ub, (size_t)d * 8);
    std::memcpy(g->asc_hio + 2 * d, starts, (size_t)R * d * 8);
    HIPCHK(hipMemcpyAsync(g->asc_dio, g->asc_hio, n_in * 8, hipMemcpyHostToDevice, g->stream));
    if (!use_wg) {   // (the batched kernels read the bounds and the starts where the block landed)
        dlb = g->asc_dio;
        dub = g->asc_dio + d;
    }
    double span = INFINITY;
    for (int k = 0; k < d; ++k) span = std::min(span, ub[k] - lb[k] + 1e-300);
    const unsigned nR = (unsigned)R;
    auto any_of = [&](const int* v, int want) {
        for (int64_t r = 0; r < R; ++r)
            if ((v[r] != 0) == (want != 0)) return true;
        return false;
    };
    if (use_wg) {
        // small models: one workgroup per start point runs its whole ascent, ONE launch (kernels_ascent.hip k_ascent_wg)
        AscWgParams pw{};
        pw.W = g->dW; pw.WT = g->dWT; pw.X = g->dX; pw.alpha = g->dalpha; pw.ld = g->ld; pw.N = g->n;
        pw.hp = make_hyper(g);
        pw.ap = AcqParams{acq_id, 0.0, 0.0};
        if (acq_params) {
            if (acq_id != BOHIP_ACQ_MAXMEAN) pw.ap.p0 = acq_params[0];
            if (acq_id == BOHIP_ACQ_MI) pw.ap.p1 = acq_params[1];
        }
        pw.beta = g->beta; pw.st = st; pw.starts = g->asc_dio + 2 * d; pw.lb = g->asc_dio; pw.ub = g->asc_dio + d; pw.R = (int)R;
        pw.maxeval = (int)std::min<int64_t>(maxeval, 1 << 30); pw.ftol_rel = ftol_rel; pw.xtol_abs = xtol_abs; pw.first_step_scale = g_asc_first_step * span;
        pw.max_ticks = g->asc_maxtime > 0.0 ? (unsigned long long)(g->asc_maxtime * 1e8) : 0ull;
        pw.passes = st.accepted;
        t_begin(g, "ascent_wg");
        if (d <= 2) hipLaunchKernelGGL(k_ascent_wg<2>, dim3(nR), dim3(AWG_THREADS), 0, g->stream, pw);
        else if (d <= 4) hipLaunchKernelGGL(k_ascent_wg<4>, dim3(nR), dim3(AWG_THREADS), 0, g->stream, pw);
        else if (d <= 8) hipLaunchKernelGGL(k_ascent_wg<8>, dim3(nR), dim3(AWG_THREADS), 0, g->stream, pw);
        else hipLaunchKernelGGL(k_ascent_wg<16>, dim3(nR), dim3(AWG_THREADS), 0, g->stream, pw);
        HIPCHK(hipGetLastError());
        t_end(g);
        // the packed result goes straight into the pinned host block (second half: the first holds the inputs): ~100 doubles written by one
        // workgroup, no copy command behind the kernel (~10 us of a 0.2-0.4 ms call)
        double* hout = g->asc_hio + n_io;
        hipLaunchKernelGGL(k_asc_final, dim3(1), dim3(256), 0, g->stream, st, d, (int)R, g->asc_best, dbx, hout, st.accepted);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(g->stream));
        if (best) { best->val = hout[0]; best->idx = (long long)hout[1]; }
        if (best_x) for (int k = 0; k < d; ++k) best_x[k] = hout[1] >= 0.0 ? hout[2 + k] : lb[k];   // :56  maxx = lowerbounds when nothing beat -Inf
        if (f_out) std::memcpy(f_out, hout + 2 + d, (size_t)R * 8);
        if (x_out) std::memcpy(x_out, hout + 2 + d + R, (size_t)R * d * 8);
        if (evals_out) {
            double mx = 0.0;
            for (int64_t r = 0; r < R; ++r) mx = std::max(mx, hout[2 + d + R + (size_t)R * d + r]);
            *evals_out = (int64_t)mx;
        }
        t_collect(g);
        return 0;
    }
    const bool free_run = !g_asc_lockstep;
    if (free_run) {
        // (the counters of a pass clear themselves; this is for a call that was cut short -- queued before the start kernel, where the device
        // waits for the host's copy anyway)
        HIPCHK(hipMemsetAsync(st.nact, 0, (size_t)2 * ASC_RING * sizeof(unsigned), g->stream));
        for (int i = 0; i < ASC_RING; ++i) st.h_cnt[i] = 0;
    }
    hipLaunchKernelGGL(k_asc_start, dim3(nR), dim3(64), 0, g->stream, st, d, g->asc_dio + 2 * d, dlb, dub);
    bool first_folded = false;   // the start points' adoption and first direction ran in the gradient kernel's last workgroup (SmallFold, on = 2)
    {
        SmallFold fold{};
        fold.on = 2; fold.R = (int)R; fold.ring_slot = ASC_RING - 1; fold.st = st; fold.lb = dlb; fold.ub = dub; fold.first_step_scale = g_asc_first_step * span;
        g->asc_fold = (free_run && g_asc_fold) ? &fold : nullptr;
        g->asc_fold_done = false;
        const int rc_first = score_grad_core(g, acq_id, acq_params, st.Xt, R, st.ft, st.Gt);
        g->asc_fold = nullptr;
        CHK(rc_first);
        first_folded = g->asc_fold_done;
    }
    bool any_active = true;
    if (first_folded) {
    } else if (free_run) {
        // no synchronisation here: how many start points are active at all is counted by the adopt kernel into the ring slot the host reads
        // once the first pass is queued (below)
        hipLaunchKernelGGL(k_asc_adopt_count, dim3(nR), dim3(64), 0, g->stream, st, d, (int)R, ASC_RING - 1);
        HIPCHK(hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_asc_adopt, dim3(nR), dim3(64), 0, g->stream, st, d);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(g->stream));
        any_active = any_of(st.h_active, 1);
    }
    int64_t evals = 1;
    int nh = 0, it = 0;
    int64_t fr_e = 0, fr_converged_at = -1;   // free-running form: what its loop leaves for after the call's one synchronisation
    bool fr_none_active = false, fr_pending = false;
    const auto t_start = std::chrono::steady_clock::now();
    auto out_of_time = [&]() {   // NLopt's maxtime (reference src/acquisition.jl:24-27 forwards it): checked once per iteration
        return g->asc_maxtime > 0.0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() >= g->asc_maxtime;
    };
    if (!g_asc_lockstep) {
        // FREE-RUNNING form (k_asc_step): pass after pass is enqueued without waiting; the number of start points still active
        // after pass e is read LAG passes later from pinned memory (by then it has long been written: no stall), so at most
        // LAG passes run beyond convergence.  Same per-start-point trajectories as the lock-step form below.
        const int LAG = 1;   // (2 until round 4: with 17 passes per call instead of 228 a wasted pass is 5 % of it; one queued pass keeps the device fed)
        if (any_active) {
            if (!first_folded) hipLaunchKernelGGL(k_asc_direction, dim3(nR), dim3(64), 0, g->stream, st, d, (int)R, 0, 0, dlb, dub, g_asc_first_step * span);
            int64_t e = 0, converged_at = -1;
            bool none_active = false;
            auto read_count = [&](int64_t pass) -> int {   // active start points after `pass` (waits for it if need be)
                volatile int* w = st.h_cnt + (pass % ASC_RING);
                const auto t0 = std::chrono::steady_clock::now();
                int v;
                while ((v = *w) == 0) {
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0) return -1;
                }
                *w = 0;
                return v - 1;
            };
            for (; evals < maxeval && !out_of_time(); ++e) {
                g->asc_go = st.ticket;   // (written by the adopt kernel and by every pass's step: active start points)
                SmallFold fold{};
                fold.on = 1; fold.R = (int)R; fold.ring_slot = (int)(e % ASC_RING); fold.st = st; fold.lb = dlb; fold.ub = dub; fold.ftol_rel = ftol_rel; fold.xtol_abs = xtol_abs;
                g->asc_fold = g_asc_fold ? &fold : nullptr;
                g->asc_fold_done = false;
                const int rc_pass = score_grad_core(g, acq_id, acq_params, st.Xt, R, st.ft, st.Gt);
                g->asc_go = nullptr;
                g->asc_fold = nullptr;
                CHK(rc_pass);
                ++evals;
                if (!g->asc_fold_done)   // (more than 16 start points, d > 16, or the batch took another path: the step as a launch of its own)
                    hipLaunchKernelGGL(k_asc_step, dim3(nR), dim3(64), 0, g->stream, st, d, (int)R, dlb, dub, g_asc_first_step * span, ftol_rel, xtol_abs,
                                       (int)(e % ASC_RING));
                HIPCHK(hipGetLastError());
                if (e == 0) {   // the adopt kernel's count (the device is busy with the first pass meanwhile)
                    const int n0 = read_count(ASC_RING - 1);
                    if (n0 < 0) return fail(BOHIP_E_HIP, "device ascent: the first evaluation did not report back within 30 s");
                    if (n0 == 0) { none_active = true; ++e; break; }   // no start point with a finite value: the queued pass changes nothing
                }
                if (e >= LAG) {
                    const int n = read_count(e - LAG);
                    if (n < 0) return fail(BOHIP_E_HIP, "device ascent: an evaluation pass did not report back within 30 s");
                    if (n == 0) { converged_at = e - LAG; ++e; break; }
                }
            }
            // (no synchronisation here: the final arg-max and the copy home are queued behind the last pass at once; the counts of the last
            // LAG passes are looked at after the one synchronisation of the call, below)
            fr_e = e; fr_converged_at = converged_at; fr_none_active = none_active; fr_pending = true;
        }
    } else {
    while (evals < maxeval && any_active && !out_of_time()) {
        hipLaunchKernelGGL(k_asc_direction, dim3(nR), dim3(64), 0, g->stream, st, d, (int)R, nh, (it + ASC_M - 1) % ASC_M, dlb, dub,
                           g_asc_first_step * span);
        for (int bt = 0; bt < ASC_MAX_BT; ++bt) {   // backtracking Armijo, all start points per device pass
            CHK(score_grad_core(g, acq_id, acq_params, st.Xt, R, st.ft, st.Gt));
            ++evals;
            hipLaunchKernelGGL(k_asc_linesearch, dim3(nR), dim3(64), 0, g->stream, st, d, dlb, dub);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(g->stream));
            if (bt == 0 && it > 0) any_active = any_of(st.h_active, 1);   // written by the previous k_asc_update
            if (!any_of(st.h_accepted, 0) || evals >= maxeval || !any_active) break;
        }
        if (!any_active) { --evals; break; }   // the speculative evaluation of an already converged set is not counted
        hipLaunchKernelGGL(k_asc_update, dim3(nR), dim3(64), 0, g->stream, st, d, (int)R, it % ASC_M, ftol_rel, xtol_abs);
        nh = std::min(nh + 1, ASC_M);
        ++it;
    }
    }
    {
        double* hout = g->asc_hio + n_io;   // (pinned: written by the kernel itself, see the one-launch form above)
        hipLaunchKernelGGL(k_asc_final, dim3(1), dim3(256), 0, g->stream, st, d, (int)R, g->asc_best, dbx, hout, (const int*)nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(g->stream));
        if (fr_pending) {
            const int LAG = 1;
            if (fr_converged_at < 0)   // stopped by maxeval / maxtime, or converged within the last LAG passes
                for (int64_t p2 = std::max<int64_t>(0, fr_e - LAG); p2 < fr_e && fr_converged_at < 0; ++p2)
                    if (st.h_cnt[p2 % ASC_RING] == 1) fr_converged_at = p2;
            if (fr_converged_at >= 0) evals = 2 + fr_converged_at;   // passes that were needed: the first one + passes 0 .. converged_at
            if (fr_none_active) evals = 1;
        }
        if (best) { best->val = hout[0]; best->idx = (long long)hout[1]; }
        if (best_x) for (int k = 0; k < d; ++k) best_x[k] = hout[1] >= 0.0 ? hout[2 + k] : lb[k];   // :56  maxx = lowerbounds when nothing beat -Inf
        if (f_out) std::memcpy(f_out, hout + 2 + d, (size_t)R * 8);
        if (x_out) std::memcpy(x_out, hout + 2 + d + R, (size_t)R * d * 8);
    }
    if (evals_out) *evals_out = evals;
    t_collect(g);
    return 0;
}

double bohip_thompson_normal(uint64_t seed, int64_t s, int64_t j) { return thompson_normal(seed, s, j); }

// ---- :GN_DIRECT_L (reference src/acquisition.jl:7-9, :20-38): the dividing-rectangles search, bookkeeping in direct_l.h ----
struct bohip_direct { DirectL s; };

int bohip_direct_create(int64_t d, const double* lb, const double* ub, int64_t maxeval, double stopval, double maxtime,
                        bohip_direct** out) {
    if (d < 1 || !lb || !ub || !out) return fail(BOHIP_E_ARG, "bad arguments");
    for (int64_t i = 0; i < d; ++i)
        if (!(lb[i] <= ub[i])) return fail(BOHIP_E_ARG, "direct: lower bound above upper bound");
    try {
        *out = new bohip_direct{DirectL(d, lb, ub, maxeval, stopval, maxtime)};
    } catch (const std::exception& e) { return fail(BOHIP_E_ARG, e.what()); }
    return 0;
}
void bohip_direct_destroy(bohip_direct* s) { delete s; }
int bohip_direct_ask(bohip_direct* s, double* X, int64_t cap, int64_t* n) {
    if (!s || !n || cap < 0 || (cap > 0 && !X)) return fail(BOHIP_E_ARG, "bad arguments");
    if (cap == 0) {   // size query: plans the iteration, hands out nothing yet
        try { *n = s->s.plan_next(); } catch (const std::exception& e) { return fail(BOHIP_E_ARG, e.what()); }
        return 0;
    }
    int64_t m = 0;
    try { m = s->s.ask(X, cap); } catch (const std::exception& e) { return fail(BOHIP_E_ARG, e.what()); }
    if (m < 0) return fail(BOHIP_E_ARG, "direct: the buffer is smaller than this iteration's batch (maxeval columns always suffice)");
    *n = m;
    return 0;
}
int bohip_direct_tell(bohip_direct* s, const double* f, int64_t n) {
    if (!s || !f || n < 1) return fail(BOHIP_E_ARG, "bad arguments");
    try {
        if (!s->s.tell(f, n)) return fail(BOHIP_E_STATE, "direct: tell without a matching ask");
    } catch (const std::exception& e) { return fail(BOHIP_E_ARG, e.what()); }   // (nothing throws across the ABI: allocation failures end here)
    return 0;
}
int bohip_direct_best(const bohip_direct* s, double* best_f, double* best_x, int64_t* evaluations, int64_t* iterations) {
    if (!s) return fail(BOHIP_E_ARG, "bad arguments");
    const double f = s->s.best(best_x);
    if (best_f) *best_f = f;
    if (evaluations) *evaluations = s->s.evals;
    if (iterations) *iterations = s->s.iterations;
    return 0;
}

// The whole search in one call: ask -> ONE scoring call of the iteration's points -> tell, until DIRECT-L stops.
// acq_id = BOHIP_ACQ_THOMPSON_DRAW: the objective is x -> myrand(model, x) (src/acquisitionfunctions.jl:107-108), one posterior draw
// per evaluated point, mu + sigma z with z = bohip_thompson_normal(seed, 0, e) for the e-th evaluation of the search.
int bohip_gp_direct_max(bohip_gp* g, int acq_id, const double* acq_params, const double* lb, const double* ub, int64_t maxeval,
                        double stopval, double maxtime, uint64_t seed, double* best_f, double* best_x, int64_t* evaluations,
                        int64_t* device_calls) {
#pragma clang fp contract(off)
    if (!g || !lb || !ub) return fail(BOHIP_E_ARG, "bad arguments");
    if (acq_id < 0 || acq_id > BOHIP_ACQ_THOMPSON_DRAW) return fail(BOHIP_E_ARG, "unknown acq_id");
    if (g->n == 0) return fail(BOHIP_E_STATE, "model has no observations");
    const int64_t d = g->d;
    for (int64_t i = 0; i < d; ++i)
        if (!(lb[i] <= ub[i])) return fail(BOHIP_E_ARG, "direct: lower bound above upper bound");
    try {
    DirectL s(d, lb, ub, maxeval, stopval, maxtime);
    std::vector<double> X, f, var;
    int64_t calls = 0, e = 0;
    for (;;) {
        const int64_t m = s.plan_next();       // (an iteration's batch is at most 2 d points per potentially optimal rectangle)
        if (m == 0) break;
        if ((int64_t)f.size() < m) { X.resize((size_t)m * d); f.resize(m); var.resize(m); }
        s.ask(X.data(), m);
        if (acq_id == BOHIP_ACQ_THOMPSON_DRAW) {
            CHK(bohip_gp_predict(g, X.data(), m, f.data(), var.data()));
            for (int64_t c = 0; c < m; ++c, ++e) {
                const double sd = std::sqrt(std::max(var[c], 0.0));
                const double t = sd * thompson_normal(seed, 0, e);
                f[c] = f[c] + t;
            }
        } else {
            CHK(bohip_gp_score(g, acq_id, acq_params, X.data(), m, f.data(), nullptr));
        }
        ++calls;
        s.tell(f.data(), m);
    }
    const double bf = s.best(best_x);
    if (best_f) *best_f = bf;
    if (evaluations) *evaluations = s.evals;
    if (device_calls) *device_calls = calls;
    } catch (const std::exception& e) { return fail(BOHIP_E_ARG, e.what()); }   // (nothing throws across the ABI)
    return 0;
}

int bohip_gp_thompson(bohip_gp* g, const double* Xs, int64_t R, int64_t S, uint64_t seed, int64_t j0, bohip_best* best) {
    if (!g || R <= 0 || S <= 0 || !Xs || !best) return fail(BOHIP_E_ARG, "bad arguments");
    HIPCHK(hipSetDevice(g->device));
    t_reset(g);
    CHK(ensure_xs(g, R));
    CHK(ensure_score_scratch(g, R));
    HIPCHK(hipMemcpyAsync(g->dXs, Xs, (size_t)R * g->d * 8, hipMemcpyHostToDevice, g->stream));
    CHK(score_core(g, BOHIP_ACQ_MAXMEAN, nullptr, g->dXs, R, g->dmu, g->dvar, nullptr, nullptr));
    if (g->thompson_cap < S) {
        if (g->dthompson) hipFree(g->dthompson);
        g->dthompson = nullptr;
        HIPCHK(hipMalloc(&g->dthompson, S * sizeof(Best)));
        g->thompson_cap = S;
    }
    Best* dout = g->dthompson;
    t_begin(g, "thompson");
    hipLaunchKernelGGL(k_thompson, dim3(S), dim3(256), 0, g->stream, g->dmu, g->dvar, R, seed, j0, dout, 0ll);
    t_end(g);
    hipError_t e = hipMemcpyAsync(best, dout, S * sizeof(Best), hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e != hipSuccess) return fail(BOHIP_E_HIP, hipGetErrorString(e));
    t_collect(g);
    return 0;
}

int bohip_gp_set_stream(bohip_gp* g, void* stream) {
    if (!g) return fail(BOHIP_E_ARG, "null handle");
    HIPCHK(hipStreamSynchronize(g->stream));
    g->stream = stream ? (hipStream_t)stream : g->own_stream;
    return 0;
}
int bohip_gp_synchronize(bohip_gp* g) {
    if (!g) return fail(BOHIP_E_ARG, "null handle");
    HIPCHK(hipStreamSynchronize(g->stream));
    t_collect(g);
    return 0;
}

int bohip_gp_get_factor(bohip_gp* g, double* L) {
    if (!g || !L) return fail(BOHIP_E_ARG, "null argument");
    HIPCHK(hipSetDevice(g->device));
    CHK(ensure_fresh(g));
    if (g->n == 0) return 0;
    HIPCHK(hipMemcpy2DAsync(L, g->n * 8, g->dL, g->ld * 8, g->n * 8, g->n, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    return 0;
}
int bohip_gp_get_alpha(bohip_gp* g, double* alpha) {
    if (!g || !alpha) return fail(BOHIP_E_ARG, "null argument");
    HIPCHK(hipSetDevice(g->device));
    CHK(ensure_fresh(g));
    if (g->n == 0) return 0;
    HIPCHK(hipMemcpyAsync(alpha, g->dalpha, g->n * 8, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    return 0;
}
static int comm_nranks(const bohip_gp* g, int64_t* value);
static int comm_rccl_version(int64_t* value);
int bohip_gp_info(const bohip_gp* g, int what, int64_t* value) {
    if (!g || !value) return fail(BOHIP_E_ARG, "null argument");
    switch (what) {
        case BOHIP_INFO_PIVOT: *value = g->pivot; return 0;
        case BOHIP_INFO_CAPACITY: *value = g->cap; return 0;
        case BOHIP_INFO_REFITS: *value = g->refits; return 0;
        case BOHIP_INFO_APPENDS: *value = g->appends; return 0;
        case BOHIP_INFO_CHOL_FORM: *value = g->chol_form_last; return 0;
        case BOHIP_INFO_CHOL_FALLBACKS: *value = g->chol_fallbacks; return 0;
        case BOHIP_INFO_CHOL_ABORT_TILES: *value = g->chol_abort_T; return 0;
        case BOHIP_INFO_JITTER_STEPS: *value = g->jitter_steps_last; return 0;
        case BOHIP_INFO_SCORE_LAUNCHES: *value = g->score_launches; return 0;
        case BOHIP_INFO_SCORE_CHUNK: *value = g->chunk_now; return 0;
        case BOHIP_INFO_KERNEL_CLOCK_MHZ: {   // since the previous read; synchronises the handle's stream
            *value = 0;
            if (!g->dclk) return 0;
            unsigned long long c[2] = {0, 0};
            if (hipSetDevice(g->device) != hipSuccess || hipStreamSynchronize(g->stream) != hipSuccess ||
                hipMemcpy(c, g->dclk, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess || hipMemset(g->dclk, 0, sizeof(c)) != hipSuccess)
                return fail(BOHIP_E_HIP, "reading the clock sample failed");
            if (c[1] > 0) *value = (int64_t)((double)c[0] / (double)c[1] * 100.0 + 0.5);   // wall_clock64 ticks at 100 MHz
            return 0;
        }
        case BOHIP_INFO_COMM_NRANKS: return comm_nranks(g, value);   // read back from the communicator (multigpu.hip)
        case BOHIP_INFO_COMM_EXCHANGES: *value = g->comm_exchanges; return 0;
        case BOHIP_INFO_COMM_RCCL_VERSION: return comm_rccl_version(value);
        case BOHIP_INFO_CHOL_LOCK_SKIPS: *value = g->chol_lock_skips; return 0;
        default: return fail(BOHIP_E_ARG, "unknown info id");
    }
}
int bohip_gp_set_batch_hint(bohip_gp* g, int64_t total_candidates) {
    if (!g || total_candidates < 0) return fail(BOHIP_E_ARG, "bad arguments");
    g->batch_hint = total_candidates;
    return 0;
}
int bohip_gp_set_jitter(bohip_gp* g, double rel, int max_tries) {
    if (!g || !(rel >= 0.0) || max_tries < 0 || max_tries > 32) return fail(BOHIP_E_ARG, "bad arguments");
    g->jitter_rel = rel;
    g->jitter_tries = rel > 0.0 ? max_tries : 0;
    return 0;
}
int bohip_gp_set_maxtime(bohip_gp* g, double seconds) {
    if (!g || !(seconds >= 0.0)) return fail(BOHIP_E_ARG, "bad arguments");
    g->asc_maxtime = seconds;
    return 0;
}
int bohip_gp_set_ascent_stop(bohip_gp* g, double ftol_abs, double xtol_rel, double stopval) {
    if (!g || !(ftol_abs >= 0.0) || !(xtol_rel >= 0.0) || stopval != stopval) return fail(BOHIP_E_ARG, "bad arguments");
    g->asc_ftol_abs = ftol_abs;
    g->asc_xtol_rel = xtol_rel;
    g->asc_stopval = stopval;
    return 0;
}
int bohip_gp_enable_timing(bohip_gp* g, int on) {
    if (!g) return fail(BOHIP_E_ARG, "null handle");
    g->timing = on != 0;
    g->timing_dominant_only = on == 2 || on == 3;
    g->timing_accumulate = on == 3;
    g->tused = 0;
    g->tlabel.clear();
    return 0;
}
int bohip_gp_get_timing(bohip_gp* g, const char** names, double* ms, int cap) {
    if (!g) return 0;
    if (g->timing_accumulate) {   // everything recorded since enable_timing(3) / the previous read
        hipSetDevice(g->device);
        t_collect(g, true);
    }
    const int n = (int)std::min<size_t>(g->tnames.size(), cap > 0 ? cap : 0);
    for (int i = 0; i < n; ++i) {
        if (names) names[i] = g->tnames[i].c_str();
        if (ms) ms[i] = g->tms[i];
    }
    return (int)g->tnames.size();
}

}  // extern "C"
#include "multigpu.hip"
extern "C" {
// tools only: the resident W = L^-1 (which = 0) or W' (which = 1), n x n row-major
int bohip_debug_read_w(bohip_gp* g, int which, double* out) {
    if (!g || !out) return BOHIP_E_ARG;
    hipSetDevice(g->device);
    hipStreamSynchronize(g->stream);
    const double* src = which ? g->dWT : g->dW;
    return hipMemcpy2D(out, (size_t)g->n * 8, src, (size_t)g->ld * 8, (size_t)g->n * 8, (size_t)g->n, hipMemcpyDeviceToHost) == hipSuccess ? 0 : BOHIP_E_HIP;
}
// test hook (tests/test_exec_tasks.py): the executor's task records for T row tiles with the three matrices at the fake
// addresses base_L/S/W (bytes) and flag word 0 at index 0 -- the CPU test replays them against a model of the chain.
// out: n x 16 uint64 words (the 128-byte records); returns the number of records, qbeg[0..EX_NQ] (EX_NQ + 1 = 7 ints) the queue boundaries,
// layout[0..9] the word offsets of panel, solved, crit, rest, col, farall, fol, colall, colr, xp inside the flag area;
// layout[10] the number of solve-follower workgroups of the chain kernel (CH_NSF).
int64_t bohip_debug_exec_tasks(int T, int64_t ld, uint64_t base_L, uint64_t base_S, uint64_t base_W, uint64_t base_WT, int inv_g, uint64_t* out,
                               int64_t cap, int* qbeg, int64_t* layout) {
    std::vector<bohip::ExTask> all;
    int qb[bohip::EX_NQ + 1];
    unsigned* fb = reinterpret_cast<unsigned*>(uintptr_t(1) << 40);
    // the follower count the records are built for = the one reported in layout[10]: the env knob is read HERE (the library's
    // one-time setup may not have run yet), and the process-wide setting is left alone
    const int nsf = g_chol_nsf;
    int grp_min = g_chol_inv_grp_min;
    if (const char* e = getenv("BOHIP_CHOL_INV_GRP_MIN")) grp_min = std::max(0, atoi(e));
    exec_task_list(reinterpret_cast<double*>(base_L), reinterpret_cast<double*>(base_S), reinterpret_cast<double*>(base_W),
                   reinterpret_cast<double*>(base_WT), fb, ld, T, nsf, inv_g, all, qb, grp_min);
    for (int i = 0; i <= bohip::EX_NQ; ++i) qbeg[i] = qb[i];
    const bohip::CholFlags fl = chol_flags_layout_at(fb, nullptr, T);
    const unsigned* ptrs[10] = {fl.panel, fl.solved, fl.crit, fl.rest, fl.col, fl.farall, fl.fol, fl.colall, fl.colr, fl.xp};
    for (int i = 0; i < 10; ++i) layout[i] = ptrs[i] - fb;
    layout[10] = nsf;
    layout[11] = (int64_t)chol_inv_word(T);
    layout[12] = (int64_t)chol_xp3_word(T);
    layout[13] = exec_bulk_stride_ok(all, qb, 4) ? 2 : 1;   // bulk tiles per claim the library may use at this T
    if ((int64_t)all.size() <= cap && out) std::memcpy(out, all.data(), all.size() * sizeof(bohip::ExTask));
    return (int64_t)all.size();
}
// tools only (tools/trace_trigemm.py): the row pieces k_trigemm_sq runs for T row tiles with the alpha row at alpha_row, in issue order
// (piece = rt | mode << 16, see kernels_score.hip); returns their number
int bohip_debug_trigemm_pieces(int T, int64_t alpha_row, int* out, int cap) {
    one_time_kernel_setup();
    std::vector<int> ps;
    trigemm_pieces(T, alpha_row, ps);
    for (int i = 0; i < (int)ps.size() && i < cap; ++i) out[i] = ps[i];
    return (int)ps.size();
}
// tools and bench.py: switch the executor's inverse queues at run time (returns the previous chunk size; 0 = off: the factorisation
// alone can then be timed against its own flop count)
int bohip_debug_set_chol_inv_g(int g_new) {
    return g_chol_inv_g.exchange(std::min(64, std::max(0, g_new)));   // (atomic; a refit in flight on another thread may see either value)
}
// tools only (tools/exec_throughput.py): how fast does the executor kernel get through its task list when NOTHING has to be waited for?
// The records of the queues in `qmask` run with their counters removed (every task runnable at once, no chain kernel beside them), on
// whatever the matrices hold: the time is the floor under any schedule of the same records.  hot = 1: every operand address of the bulk
// and wave records points at the first panel of the scratch matrix (L2-resident) -- the same instruction stream without its fabric-side
// traffic.  The model's factor and inverse are garbage afterwards (the handle is marked stale: the next use refits).
int bohip_debug_exec_throughput(bohip_gp* g, unsigned qmask, int hot, int wgs, double* ms_out, double* gflop_out) {
    if (!g || !ms_out || g->n == 0) return BOHIP_E_ARG;
    hipSetDevice(g->device);
    CHK(one_time_kernel_setup());
    const int T = (int)(round_up(g->n + 1, TILE) / TILE);
    if (T <= 3) return BOHIP_E_UNSUPPORTED;
    std::vector<ExTask> all, run;
    int qb[EX_NQ + 1], qb2[EX_NQ + 1];
    exec_task_list(g->dL, g->dS, g->dW, g->dWT, g->dchol_flags, g->ld, T, g_chol_nsf, g_chol_inv_g, all, qb);
    double fl = 0.0;
    qb2[0] = 0;
    for (int qi = 0; qi < EX_NQ; ++qi) {
        if ((qmask >> qi) & 1u)
            for (int i = qb[qi]; i < qb[qi + 1]; ++i) {
                ExTask t = all[i];
                for (int d = 0; d < EX_NDEP; ++d) { t.dep_idx[d] = EX_NONE; t.dep_want[d] = 0; }
                t.sig_idx[0] = t.sig_idx[1] = EX_NONE;
                t.kc_split = 0; t.dep2_idx[0] = t.dep2_idx[1] = EX_NONE;
                if (hot && (qi == EX_QBULK || qi == EX_QWAVE)) { t.A = g->dS; t.B = g->dS + (int64_t)TILE * g->ld; }
                fl += 2.0 * TILE * CTILE * KC * t.kc;
                run.push_back(t);
            }
        qb2[qi + 1] = (int)run.size();
    }
    if (run.empty()) return BOHIP_E_ARG;
    ExTask* dt = nullptr;
    HIPCHK(hipMalloc(&dt, run.size() * sizeof(ExTask)));
    HIPCHK(hipMemcpy(dt, run.data(), run.size() * sizeof(ExTask), hipMemcpyHostToDevice));
    HIPCHK(hipMemsetAsync(g->dchol_flags, 0, chol_flag_words(T) * sizeof(unsigned), g->stream));
    ExQueues q{};
    q.tasks = dt;
    for (int i = 0; i <= EX_NQ; ++i) q.qbeg[i] = qb2[i];
    q.flags = g->dchol_flags;
    q.abort = g->dchol_flags + chol_abort_word(T);
    q.heads = q.abort + 1;
    q.ld = g->ld;
    q.spin_ticks = g_chol_spin_ticks;
    q.fill = g_chol_exec_fill;
    const int exec_wgs = wgs > 0 ? wgs : std::max(2, std::min(g_chol_exec_wgs > 0 ? g_chol_exec_wgs : 1 << 20, 2 * std::max(1, device_cus() - (9 + g_chol_nsf + 6))));
    q.nurgent = std::max(1, std::min(exec_wgs / 8, g_chol_exec_urgent > 0 ? g_chol_exec_urgent : (T >= 56 ? 16 : 32)));
    const int pairs_ = g_chol_exec_pairs == 0 ? 0 : 1;
        q.stride[0] = 1; q.stride[1] = 1; q.stride[2] = pairs_ ? 2 : 1; q.stride[EX_QBULK] = pairs_ >= 2 ? 2 * std::min(pairs_, 4) : (pairs_ ? 2 : 1);
    q.stride[EX_QROWS] = q.stride[EX_QWAVE] = g_chol_exec_inv_pairs ? 2 : 1;
    q.fill_inv = g_chol_exec_fill_inv;
    q.patience_ticks = (unsigned)g_chol_exec_patience_us * 100u;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, g->stream));
    hipLaunchKernelGGL(k_chol_exec, dim3(exec_wgs), dim3(GEMM_THREADS_8), glds3_lds_bytes<4>(), g->stream, q);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, g->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(dt);
    *ms_out = ms;
    if (gflop_out) *gflop_out = fl * 1e-9;
    g->stale = true;
    g->w_done = false;
    return 0;
}
#if BOHIP_CHOL_TRACE
int bohip_debug_chol_trace_read(unsigned long long* out, int64_t n_words) {   // tools only
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(bohip::g_chol_trace), (size_t)n_words * 8) == hipSuccess ? 0 : -3;
}
int bohip_debug_exec_trace_read(unsigned long long* out, int64_t n_words) {   // tools only
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(bohip::g_ex_trace), (size_t)n_words * 8) == hipSuccess ? 0 : -3;
}
#endif
#if BOHIP_TRACE
int bohip_debug_trace_read(unsigned long long* out, int64_t n_words) {   // tools only
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(bohip::g_trace), (size_t)n_words * 8) == hipSuccess ? 0 : -3;
}
int bohip_debug_phase_read(unsigned long long* out, int64_t n_words) {   // tools only
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(bohip::g_phase), (size_t)n_words * 8) == hipSuccess ? 0 : -3;
}
#endif
}  // extern "C"
