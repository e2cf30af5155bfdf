// kernels_ascent.hip -- SURVEY.md section 8(f) N1: the local search of acquire_max (reference src/acquisition.jl:48-68,
// NLopt :LD_LBFGS with box bounds, :23-35) as a LOCK-STEP projected L-BFGS ascent of all R start points on the device.
// One wave per start point, lane k = coordinate k (d <= 64), inner products by wave butterflies; the expensive part of an
// iteration -- value and gradient of the acquisition at R trial points -- is ONE score_grad pass of the model.
// The state (X, f, G, curvature pairs, best point seen) never leaves HBM; per evaluation the host reads one
// int per start point from pinned memory to steer the backtracking.
//   k_asc_start      X <- clip(starts), first trial = X
//   k_asc_adopt      after the first evaluation: f, G, active, best
//   k_asc_direction  two-loop recursion over the (<= 8) curvature pairs, bound blocking, first trial point
//   k_asc_linesearch Armijo test of the trial; on failure halve the step and write the next trial
//   k_asc_update     curvature pair, convergence tests (ftol_rel, xtol_abs), best point
//   k_asc_final      arg-max over the start points, strict '>' => first maximum wins (:58-66)
#include "common.h"

namespace bohip {

constexpr int ASC_M = 8;   // curvature pairs kept
// Trial points per line search (the step is halved between them).  12 until round 6: on the flank of a narrow EI ridge -- the value falls to 0
// within 1e-3 |g| of the point, tests' N = 600 model at tau = median y -- a start needs ~14 halvings and was given up as "line search failed"
// short of a KKT point (1-3 of 576 start points, tools/ascent_kkt_margin.py).  30 halvings reach 1e-9 of the first step.
constexpr int ASC_MAX_BT = 30;

struct AscentState {
    double *X, *f, *G;          // [R][d], [R], [R][d]   current point
    double *Xt, *ft, *Gt;       // trial point and its evaluation (Xt is the candidate block of score_grad)
    double *Xn, *fn, *Gn;       // accepted point of this iteration
    double *D, *Gp, *step;      // direction, projected gradient, step length
    double *S, *Y;              // [ASC_M][R][d] curvature pairs (zero rows where s'y was not positive)
    double *best_f, *best_X;
    int *active, *accepted;
    int *h_accepted, *h_active; // pinned host mirrors read by the driver loop
    // free-running form (k_asc_step): per start point its own iteration count and backtracking count, and per evaluation pass a
    // ring slot: how many start points are still active after it (device counters + the word the host polls, value + 1)
    int *it, *bt;
    unsigned *nact, *ticket;    // nact: [ASC_RING] 64-bit counters (arrivals | active << 32), 8-byte aligned; ticket: ONE word behind them: start points
                                // still active after the last finished pass -- the kernels of a queued pass return at once when it is 0 (the pass the
                                // host queued before it could know that everything had converged: ~50 us of a call)
    int *h_cnt;                 // [ASC_RING] pinned
    // NLopt's remaining stop criteria (bohip_gp_set_ascent_stop; the reference forwards them, src/acquisition.jl:24-27, and its own
    // test sets ftol_abs = eps(), test/acquisition.jl:6,9): 0 / 0 / +Inf = off
    double ftol_abs, xtol_rel, stopval;
};
__device__ __forceinline__ double asc_wsum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// The same sum when only the lanes k < d (one coordinate each) hold anything but zero -- every inner product of the ascent's step.
// __shfl_xor is two ds_bpermute_b32 per level, a trip through the LDS crossbar each: k_asc_step was 384 of them in 192 dependent round
// trips (32 sums x 6 levels), ~7 us for a few hundred flops.  Here the levels that matter (lane distance < the power of two above d) are
// DPP moves -- quad_perm for 1 and 2, row_shl / row_shr with complementary bank masks for 4 and 8 (checked lane for lane on the chip) --
// in the butterfly's own order, largest distance first, and lane 0's total is handed to every lane with one readlane: the value is the
// butterfly's bit for bit (its upper levels only ever added zeros).  d > 16 (more than one DPP row): the butterfly itself.
template <int CTRL_A, int BANK_A, int CTRL_B, int BANK_B>
__device__ __forceinline__ double asc_dpp_pair(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int l2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL_A, 0xf, BANK_A, false), h2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL_A, 0xf, BANK_A, false);
    if constexpr (CTRL_B != 0) {
        l2 = __builtin_amdgcn_update_dpp(l2, lo, CTRL_B, 0xf, BANK_B, false);
        h2 = __builtin_amdgcn_update_dpp(h2, hi, CTRL_B, 0xf, BANK_B, false);
    }
    return __hiloint2double(h2, l2);
}
__device__ __forceinline__ double asc_csum(double v, int d) {
    if (d > 16) return asc_wsum(v);
    if (d > 8) v += asc_dpp_pair<0x108, 0x3, 0x118, 0xc>(v);   // lane ^ 8: row_shl:8 into lanes 0-7, row_shr:8 into lanes 8-15
    if (d > 4) v += asc_dpp_pair<0x104, 0x5, 0x114, 0xa>(v);   // lane ^ 4
    if (d > 2) v += asc_dpp_pair<(2 | (3 << 2) | (0 << 4) | (1 << 6)), 0xf, 0, 0>(v);   // lane ^ 2: quad_perm [2, 3, 0, 1]
    if (d > 1) v += asc_dpp_pair<(1 | (0 << 2) | (3 << 4) | (2 << 6)), 0xf, 0, 0>(v);   // lane ^ 1: quad_perm [1, 0, 3, 2]
    return readlane_f64(v, 0);
}
// Four start points per wave (round 6, the step folded into k_small_u's last workgroup): lane = 16 * (start point in the wave) + coordinate,
// d <= 16.  asc_csum's levels inside every row of 16 lanes, and no readlane: the rows of a wave are at different places of their searches.
__device__ __forceinline__ double asc_rsum(double v, int d) {
    if (d > 8) v += asc_dpp_pair<0x108, 0x3, 0x118, 0xc>(v);
    if (d > 4) v += asc_dpp_pair<0x104, 0x5, 0x114, 0xa>(v);
    if (d > 2) v += asc_dpp_pair<(2 | (3 << 2) | (0 << 4) | (1 << 6)), 0xf, 0, 0>(v);
    if (d > 1) v += asc_dpp_pair<(1 | (0 << 2) | (3 << 4) | (2 << 6)), 0xf, 0, 0>(v);
    // lane 0 of the row has run asc_csum's own sequence; its total goes to the row's other lanes (row_newbcast:0).  NOT the lanes' own totals:
    // the compiler may fold the product that feeds a sum into the first level's addition (an fma whose two partners then round differently)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <bool ROWS>
__device__ __forceinline__ double asc_sum_t(double v, int d) { if constexpr (ROWS) return asc_rsum(v, d); else return asc_csum(v, d); }
// A sum over all 64 lanes (k_ascent_wg: a row of W against k*, one partial sum per lane), the same way: four DPP levels inside every row
// of 16 lanes, then the four row totals from lanes 0, 16, 32, 48.  Wave-uniform result.  (The association differs from the butterfly's --
// rows first --: the one-workgroup form agrees with the batched kernels to rounding, as it always did, not bit for bit.)
__device__ __forceinline__ double asc_wsum_dpp(double v) {
    v += asc_dpp_pair<(1 | (0 << 2) | (3 << 4) | (2 << 6)), 0xf, 0, 0>(v);
    v += asc_dpp_pair<(2 | (3 << 2) | (0 << 4) | (1 << 6)), 0xf, 0, 0>(v);
    v += asc_dpp_pair<0x104, 0x5, 0x114, 0xa>(v);
    v += asc_dpp_pair<0x108, 0x3, 0x118, 0xc>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
// Does start point r go on after an iteration that moved it by s (this lane's coordinate, new value xn) and changed the value by
// df to fn?  NLopt's tests for a maximisation: ftol_rel, ftol_abs on the improvement, xtol_abs on the step (its norm, as before),
// xtol_rel per coordinate (stop when EVERY |dx_k| <= xtol_rel |x_k|), stopval.  Wave-uniform (the sums are butterflies).
template <bool ROWS = false>
__device__ __forceinline__ bool asc_goes_on(const AscentState& st, int d, bool on, double s, double xn, double df, double fn, double moved,
                                            double ftol_rel, double xtol_abs) {
    const double n_big = asc_sum_t<ROWS>((on && fabs(s) > st.xtol_rel * fabs(xn)) ? 1.0 : 0.0, d);   // coordinates that moved by more than xtol_rel |x|
    return df > ftol_rel * fmax(fabs(fn), 1e-300) && moved > xtol_abs && df > st.ftol_abs && (st.xtol_rel <= 0.0 || n_big > 0.0) &&
           !(fn >= st.stopval);
}
constexpr int ASC_RING = 8;

__device__ __forceinline__ double asc_clip(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

__global__ __launch_bounds__(64) void k_asc_start(AscentState st, int d, const double* __restrict__ starts,
                                                  const double* __restrict__ lb, const double* __restrict__ ub) {
    const int r = blockIdx.x, k = threadIdx.x;
    if (k >= d) return;
    const double x = asc_clip(starts[(int64_t)r * d + k], lb[k], ub[k]);
    st.X[(int64_t)r * d + k] = x;
    st.Xt[(int64_t)r * d + k] = x;
}

__device__ __forceinline__ void asc_adopt_one(const AscentState& st, int r, int k, int d) {
    const double f = st.ft[r];
    if (k < d) {
        st.G[(int64_t)r * d + k] = st.Gt[(int64_t)r * d + k];
        st.best_X[(int64_t)r * d + k] = st.X[(int64_t)r * d + k];
    }
    if (k == 0) {
        st.f[r] = f;
        st.best_f[r] = f;
        const int a = isfinite(f) ? 1 : 0;
        st.active[r] = a;
        st.h_active[r] = a;
    }
}
__global__ __launch_bounds__(64) void k_asc_adopt(AscentState st, int d) { asc_adopt_one(st, blockIdx.x, threadIdx.x, d); }
// The free-running form's adopt: the same, plus this start point's iteration / backtracking counters zeroed (two memset launches per call
// before) and the number of start points that are active at all counted into ring slot `slot` the way a pass counts (see asc_step_one): the
// host reads it while the first pass is already queued, where it used to synchronise the stream to look at h_active (~20 us per call).
__global__ __launch_bounds__(64) void k_asc_adopt_count(AscentState st, int d, int R, int slot) {
    const int r = blockIdx.x, k = threadIdx.x;
    asc_adopt_one(st, r, k, d);
    if (k == 0) {
        st.it[r] = 0;
        st.bt[r] = 0;
        const int active = isfinite(st.ft[r]) ? 1 : 0;
        unsigned long long* cnt = reinterpret_cast<unsigned long long*>(st.nact) + slot;
        const unsigned long long before = atomicAdd(cnt, 1ull + ((unsigned long long)active << 32));
        if ((unsigned)(before & 0xffffffffull) == (unsigned)R - 1u) {
            const unsigned n = (unsigned)(before >> 32) + (unsigned)active;
            __hip_atomic_store(cnt, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st.ticket, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st.h_cnt + slot, (int)n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// nh curvature pairs are valid; the newest sits in slot (newest), older ones in the slots before it (ring of ASC_M)
// (ROWS: four start points per wave as in asc_step_compute, for the FIRST direction only -- nh = 0 --, with the gradient and the start
// point's activity handed in by asc_first_rows, which has just adopted them)
template <bool ROWS = false>
__device__ __forceinline__ void asc_direction_one(const AscentState& st, int r, int k, int d, int R, int nh, int newest,
                                                  const double* __restrict__ lb, const double* __restrict__ ub, double first_step_scale,
                                                  double g_in = 0.0, int active_in = 0, double f_in = 0.0) {
    const bool on = k < d && r < R;
    const int64_t o = (int64_t)r * d + k;
    const double x = on ? st.X[o] : 0.0, g = ROWS ? (on ? g_in : 0.0) : (on ? st.G[o] : 0.0);
    const double lo = on ? lb[k] : 0.0, hi = on ? ub[k] : 0.0;
    // FREE SUBSPACE (round 4): a coordinate that sits on a bound with the gradient pushing outward takes no part in the two-loop
    // recursion -- not in the vector it starts from and not in the curvature pairs' inner products.  Rounds 1-3 ran the recursion in the
    // full space and zeroed the blocked components of the result: with many active bounds (UCB at the reference's beta_t ~ 10 peaks in
    // the corners of the box) that is no quasi-Newton direction of the reduced problem, and the search crawled -- 228-309 evaluation
    // passes for the headline model's ten starts where this form needs 22-35 and SciPy's L-BFGS-B 24, same or better maxima
    // (tools/ascent_vs_scipy.py).  With no bound active the arithmetic is the old one, operation for operation.
    const bool fr = on && !((x <= lo && g < 0.0) || (x >= hi && g > 0.0));
    double q = fr ? g : 0.0;
    double al[ASC_M], rho[ASC_M];
#pragma unroll
    for (int i = 0; i < ASC_M; ++i) {   // newest -> oldest
        if (i >= nh) break;
        const int slot = (newest - i + ASC_M) % ASC_M;
        const int64_t ho = ((int64_t)slot * R + r) * d + k;
        const double s = fr ? st.S[ho] : 0.0, y = fr ? st.Y[ho] : 0.0;
        const double sy = asc_sum_t<ROWS>(y * s, d);
        rho[i] = sy > 1e-14 ? 1.0 / sy : 0.0;     // (a pair without curvature in the free subspace is skipped)
        al[i] = rho[i] * asc_sum_t<ROWS>(s * q, d);
        q -= al[i] * y;
    }
    if (nh > 0) {
        const int64_t ho = ((int64_t)newest * R + r) * d + k;
        const double s = fr ? st.S[ho] : 0.0, y = fr ? st.Y[ho] : 0.0;
        const double sy = asc_sum_t<ROWS>(s * y, d), yy = fmax(asc_sum_t<ROWS>(y * y, d), 1e-300);
        q *= sy > 1e-14 ? sy / yy : 1.0;
    }
#pragma unroll
    for (int i = ASC_M - 1; i >= 0; --i) {   // oldest -> newest
        if (i >= nh) continue;
        const int slot = (newest - i + ASC_M) % ASC_M;
        const int64_t ho = ((int64_t)slot * R + r) * d + k;
        const double s = fr ? st.S[ho] : 0.0, y = fr ? st.Y[ho] : 0.0;
        const double b = rho[i] * asc_sum_t<ROWS>(y * q, d);
        q += (al[i] - b) * s;
    }
    // do not push active constraints outward; fall back to the projected gradient if that is no ascent direction
    double D = q;
    if ((x <= lo && D < 0.0) || (x >= hi && D > 0.0)) D = 0.0;
    const double gp = ((x <= lo && g < 0.0) || (x >= hi && g > 0.0)) ? 0.0 : g;
    double slope = asc_sum_t<ROWS>(on ? gp * D : 0.0, d);
    if (!(slope > 0.0)) {
        D = gp;
        slope = asc_sum_t<ROWS>(on ? gp * gp : 0.0, d);
    }
    const int active = ROWS ? active_in : st.active[r];
    double step = 1.0;
    // The FIRST step (no curvature pair yet).  Round 6: the unit step on the projected gradient itself, which is what L-BFGS-B does -- its first
    // iterate is the generalised Cauchy point of the model with B = I, P(x + g), tried with step 1 -- but never shorter than a tenth of the box
    // (first_step_scale / |D|, the cautious step rounds 3-5 always took: it slid into the nearest local maximum where SciPy's reaches the
    // corner the gradient points at -- over 8 x 10 starts on the headline model 44 of 80 ended at or above SciPy's value from the same start,
    // with the unit step 72 of 80 -- but on a flat tail, |g| << 1, it is the only step that gets anywhere).
    if (nh == 0) step = fmax(1.0, first_step_scale / fmax(sqrt(asc_sum_t<ROWS>(on ? D * D : 0.0, d)), 1e-12));
    if (!(active && slope > 0.0)) step = 0.0;
    if (on) {
        st.D[o] = D;
        st.Gp[o] = gp;
        st.Xn[o] = x;
        st.Gn[o] = g;
        st.Xt[o] = asc_clip(x + step * D, lo, hi);
    }
    if (k == 0 && r < R) {
        st.step[r] = step;
        st.fn[r] = ROWS ? f_in : st.f[r];
        const int acc = (!active || !(slope > 0.0)) ? 1 : 0;
        st.accepted[r] = acc;
        st.h_accepted[r] = acc;
    }
}
// The free-running form's first pass in k_small_u's last workgroup (ROWS layout): k_asc_adopt_count's adoption of the start points' values and
// gradients, then k_asc_direction's first direction and trial point.  Returns (lanes k == 0) whether start point r is active at all.
__device__ __forceinline__ int asc_first_rows(const AscentState& st, int r, int k, int d, int R, const double* __restrict__ lb,
                                              const double* __restrict__ ub, double first_step_scale, double ft, double g_trial) {
    const bool on = k < d && r < R;
    const int active = (r < R && isfinite(ft)) ? 1 : 0;
    if (on) {
        st.G[(int64_t)r * d + k] = g_trial;
        st.best_X[(int64_t)r * d + k] = st.X[(int64_t)r * d + k];
    }
    if (k == 0 && r < R) {
        st.f[r] = ft;
        st.best_f[r] = ft;
        st.active[r] = active;
        st.h_active[r] = active;
        st.it[r] = 0;
        st.bt[r] = 0;
    }
    asc_direction_one<true>(st, r, k, d, R, 0, 0, lb, ub, first_step_scale, g_trial, active, ft);
    return active;
}
__global__ __launch_bounds__(64) void k_asc_direction(AscentState st, int d, int R, int nh, int newest,
                                                      const double* __restrict__ lb, const double* __restrict__ ub,
                                                      double first_step_scale) {
    asc_direction_one(st, blockIdx.x, threadIdx.x, d, R, nh, newest, lb, ub, first_step_scale);
}

__global__ __launch_bounds__(64) void k_asc_linesearch(AscentState st, int d, const double* __restrict__ lb,
                                                       const double* __restrict__ ub) {
    const int r = blockIdx.x, k = threadIdx.x;
    if (st.accepted[r]) return;
    const bool on = k < d;
    const int64_t o = (int64_t)r * d + k;
    const double x = on ? st.X[o] : 0.0, xt = on ? st.Xt[o] : 0.0, gp = on ? st.Gp[o] : 0.0;
    const double dot = asc_csum(on ? gp * (xt - x) : 0.0, d);
    const double ft = st.ft[r], f = st.f[r];
    const bool ok = isfinite(ft) && ft >= f + 1e-4 * dot;
    if (ok) {
        if (on) {
            st.Xn[o] = xt;
            st.Gn[o] = st.Gt[o];
        }
        if (k == 0) {
            st.fn[r] = ft;
            st.accepted[r] = 1;
            st.h_accepted[r] = 1;
        }
    } else {
        const double step = st.step[r] * 0.5;
        if (on) st.Xt[o] = asc_clip(x + step * st.D[o], lb[k], ub[k]);
        if (k == 0) st.step[r] = step;
    }
}

__global__ __launch_bounds__(64) void k_asc_update(AscentState st, int d, int R, int slot, double ftol_rel, double xtol_abs) {
    const int r = blockIdx.x, k = threadIdx.x;
    const bool on = k < d;
    const int64_t o = (int64_t)r * d + k;
    const double x = on ? st.X[o] : 0.0, xn = on ? st.Xn[o] : 0.0, g = on ? st.G[o] : 0.0, gn = on ? st.Gn[o] : 0.0;
    const double s = xn - x, y = -(gn - g);
    const double f = st.f[r], fn = st.fn[r], df = fn - f, best_before = st.best_f[r];
    const double moved = sqrt(asc_csum(s * s, d));
    const bool good = asc_csum(s * y, d) > 1e-14;
    int active = st.active[r];
    active = (active && asc_goes_on(st, d, on, s, xn, df, fn, moved, ftol_rel, xtol_abs)) ? 1 : 0;
    if (on) {
        const int64_t ho = ((int64_t)slot * R + r) * d + k;
        st.S[ho] = good ? s : 0.0;
        st.Y[ho] = good ? y : 0.0;
        st.X[o] = xn;
        st.G[o] = gn;
        st.Xt[o] = xn;
        if (fn > best_before) st.best_X[o] = xn;
    }
    if (k == 0) {
        st.f[r] = fn;
        if (fn > best_before) st.best_f[r] = fn;
        st.active[r] = active;
        st.h_active[r] = active;
    }
}

// FREE-RUNNING form: one launch per evaluation pass does, for every start point on its own, whatever comes next in ITS iteration --
// the Armijo test of its trial; on failure the halved step (up to ASC_MAX_BT trials, as in the lock-step driver); on success (or after
// the 12th failure) the update of k_asc_update and at once the next direction of k_asc_direction with its first trial point.
// A start point's sequence of trial points, values and curvature pairs is exactly that of the lock-step form (its arithmetic
// never looks at another start point, and a score_grad result does not depend on what else is in the batch), but no start
// point waits for the slowest line search of the batch, and the HOST takes no decision between two passes: it enqueues pass
// after pass and reads, two passes behind, how many start points were still active (a pinned word written by the last
// workgroup of a pass).  Lock-step: five launches + a stream synchronisation per pass, 94-117 us at N = 3000 with 10 starts.
// (one wave: lane k = coordinate k of start point r; ring_slot < 0: no pass bookkeeping -- the one-workgroup-per-start kernel)
// slot order -> age order of the curvature pairs (see asc_step_one); NEWEST = slot of the pair made in this step (taken from registers)
template <int NEWEST>
__device__ __forceinline__ void asc_age_order_t(const double (&sv)[ASC_M], const double (&yv)[ASC_M], double s_new, double y_new,
                                                double (&ps)[ASC_M], double (&py)[ASC_M]) {
    ps[0] = s_new;
    py[0] = y_new;
#pragma unroll
    for (int i = 1; i < ASC_M; ++i) {
        ps[i] = sv[(NEWEST - i + ASC_M) % ASC_M];
        py[i] = yv[(NEWEST - i + ASC_M) % ASC_M];
    }
}
template <bool ROWS = false>
__device__ __forceinline__ void asc_age_order(int newest, const double (&sv)[ASC_M], const double (&yv)[ASC_M], double s_new, double y_new,
                                              double (&ps)[ASC_M], double (&py)[ASC_M]) {
    static_assert(ASC_M == 8, "eight arms");
    switch (ROWS ? newest : __builtin_amdgcn_readfirstlane(newest)) {     // (one start point per wave: uniform; four: up to four arms run)
        case 0: asc_age_order_t<0>(sv, yv, s_new, y_new, ps, py); break;
        case 1: asc_age_order_t<1>(sv, yv, s_new, y_new, ps, py); break;
        case 2: asc_age_order_t<2>(sv, yv, s_new, y_new, ps, py); break;
        case 3: asc_age_order_t<3>(sv, yv, s_new, y_new, ps, py); break;
        case 4: asc_age_order_t<4>(sv, yv, s_new, y_new, ps, py); break;
        case 5: asc_age_order_t<5>(sv, yv, s_new, y_new, ps, py); break;
        case 6: asc_age_order_t<6>(sv, yv, s_new, y_new, ps, py); break;
        default: asc_age_order_t<7>(sv, yv, s_new, y_new, ps, py); break;
    }
}
struct AscStepRegs {   // everything asc_step_compute may need of start point r's state, coordinate k: ONE round of loads
    int active, bt, it;
    double lo, hi, x, xt, gpo, g_old, d_old, f, step_old, best_before;
    double sv[ASC_M], yv[ASC_M];   // the curvature pairs (slot order)
};
template <bool ROWS = false>
__device__ __forceinline__ void asc_step_load(const AscentState& st, int r, int k, int d, int R, const double* __restrict__ lb,
                                              const double* __restrict__ ub, AscStepRegs& a) {
    const bool on = k < d && r < R;
    const int64_t o = (int64_t)r * d + k;
    const int rl = r < R ? r : 0;
    // (taken one by one as the branches reach them these were a chain of five or six dependent memory round trips: 8 us for a few hundred flops)
    a.active = r < R ? st.active[rl] : 0;
    a.lo = on ? lb[k] : 0.0; a.hi = on ? ub[k] : 0.0;
    a.x = on ? st.X[o] : 0.0; a.xt = on ? st.Xt[o] : 0.0; a.gpo = on ? st.Gp[o] : 0.0;
    a.g_old = on ? st.G[o] : 0.0; a.d_old = on ? st.D[o] : 0.0;
    a.f = st.f[rl]; a.step_old = st.step[rl]; a.best_before = st.best_f[rl];
    a.bt = st.bt[rl]; a.it = st.it[rl];
#pragma unroll
    for (int i = 0; i < ASC_M; ++i) {
        const int64_t ho = ((int64_t)i * R + r) * d + k;
        a.sv[i] = on ? st.S[ho] : 0.0;
        a.yv[i] = on ? st.Y[ho] : 0.0;
    }
}
// ROWS (four start points per wave, lane = 16 * row + k, d <= 16; rows with r >= R idle): every sum is the row's own (asc_rsum), and the
// number of the pass's active start points is the caller's to add up -- the return value, valid in the lanes k == 0, says whether r goes on.
// ft / g_trial: the trial point's value and gradient (k_small_u's last workgroup has just made them; k_asc_step loads them).
template <bool ROWS = false>
__device__ __forceinline__ int asc_step_compute(const AscentState& st, const AscStepRegs& a, int r, int k, int d, int R, double ftol_rel, double xtol_abs,
                                                int ring_slot, double ft, double g_trial) {
    const bool on = k < d && r < R;
    const int64_t o = (int64_t)r * d + k;
    int active = a.active;
    const double lo = a.lo, hi = a.hi, x = a.x, xt = a.xt, gpo = a.gpo, g_old = a.g_old, d_old = a.d_old, f = a.f, step_old = a.step_old,
                 best_before = a.best_before;
    const int bt = a.bt, it = a.it;
    const double (&sv)[ASC_M] = a.sv;
    const double (&yv)[ASC_M] = a.yv;
    if (active) {
        const double dot = asc_sum_t<ROWS>(on ? gpo * (xt - x) : 0.0, d);
        const bool ok = isfinite(ft) && ft >= f + 1e-4 * dot;
        if (!ok && bt < ASC_MAX_BT - 1) {
            const double step = step_old * 0.5;
            if (on) st.Xt[o] = asc_clip(x + step * d_old, lo, hi);
            if (k == 0) { st.step[r] = step; st.bt[r] = bt + 1; }
        } else {
            // ---- the iteration ends (k_asc_update with Xn = the accepted trial, or X itself after ASC_MAX_BT failures)
            const double g = g_old;
            const double xn = ok ? xt : x, gn = ok ? g_trial : g, fn = ok ? ft : f;
            const double s = xn - x, y = -(gn - g), df = fn - f;
            const double moved = sqrt(asc_sum_t<ROWS>(s * s, d));
            const bool good = asc_sum_t<ROWS>(s * y, d) > 1e-14;
            active = asc_goes_on<ROWS>(st, d, on, s, xn, df, fn, moved, ftol_rel, xtol_abs) ? 1 : 0;
            const int slot = it % ASC_M;
            const double s_new = good ? s : 0.0, y_new = good ? y : 0.0;
            if (on) {
                const int64_t ho = ((int64_t)slot * R + r) * d + k;
                st.S[ho] = s_new;
                st.Y[ho] = y_new;
                st.X[o] = xn;
                st.G[o] = gn;
                if (fn > best_before) st.best_X[o] = xn;
            }
            if (k == 0) {
                st.f[r] = fn;
                if (fn > best_before) st.best_f[r] = fn;
                st.it[r] = it + 1;
                st.bt[r] = 0;
            }
            double xt_new = xn;
            if (active) {
                // ---- the next direction (k_asc_direction at x = xn, g = gn; the newest pair is the one just written: taken
                // from registers, the older ones were loaded up front by slot -- every lane reads only elements it wrote itself)
                const int nh = it + 1 < ASC_M ? it + 1 : ASC_M, newest = slot;
                // The pairs in AGE order (i = 0: the one just made, i = 1: the newest loaded one, ...).  The registers hold them by SLOT; the
                // rotation is resolved by ONE wave-uniform switch on `newest` whose eight arms are register renames (28 moves) -- until
                // round 5 every use selected its pair out of all eight slots (16 v_cndmask per pair and use: the step grew by 0.6 us per pair).
                double ps_age[ASC_M], py_age[ASC_M];
                asc_age_order<ROWS>(newest, sv, yv, s_new, y_new, ps_age, py_age);
                auto pair_s = [&](int i) { return ps_age[i]; };
                auto pair_y = [&](int i) { return py_age[i]; };
                // (the free subspace of asc_direction_one: coordinates on a bound with the gradient pushing outward stay out)
                const bool fr = on && !((xn <= lo && gn < 0.0) || (xn >= hi && gn > 0.0));
                double q = fr ? gn : 0.0;
                double al[ASC_M], rho[ASC_M];
#pragma unroll
                for (int i = 0; i < ASC_M; ++i) {   // newest -> oldest
                    if (i >= nh) break;
                    const double ps = fr ? pair_s(i) : 0.0, py = fr ? pair_y(i) : 0.0;
                    const double sy = asc_sum_t<ROWS>(py * ps, d);
                    rho[i] = sy > 1e-14 ? 1.0 / sy : 0.0;
                    al[i] = rho[i] * asc_sum_t<ROWS>(ps * q, d);
                    q -= al[i] * py;
                }
                {
                    const double sf = fr ? s_new : 0.0, yf = fr ? y_new : 0.0;
                    const double sy = asc_sum_t<ROWS>(sf * yf, d), yy = fmax(asc_sum_t<ROWS>(yf * yf, d), 1e-300);
                    q *= sy > 1e-14 ? sy / yy : 1.0;
                }
#pragma unroll
                for (int i = ASC_M - 1; i >= 0; --i) {   // oldest -> newest
                    if (i >= nh) continue;
                    const double ps = fr ? pair_s(i) : 0.0, py = fr ? pair_y(i) : 0.0;
                    const double b = rho[i] * asc_sum_t<ROWS>(py * q, d);
                    q += (al[i] - b) * ps;
                }
                double D = q;
                if ((xn <= lo && D < 0.0) || (xn >= hi && D > 0.0)) D = 0.0;
                const double gp = ((xn <= lo && gn < 0.0) || (xn >= hi && gn > 0.0)) ? 0.0 : gn;
                double slope = asc_sum_t<ROWS>(on ? gp * D : 0.0, d);
                if (!(slope > 0.0)) {
                    D = gp;
                    slope = asc_sum_t<ROWS>(on ? gp * gp : 0.0, d);
                }
                if (slope > 0.0) {
                    if (on) {
                        st.D[o] = D;
                        st.Gp[o] = gp;
                    }
                    if (k == 0) st.step[r] = 1.0;
                    xt_new = asc_clip(xn + D, lo, hi);
                } else {
                    active = 0;   // no ascent direction left: the lock-step form spends one more pass to find s = 0
                }
            }
            if (on) st.Xt[o] = xt_new;
            if (k == 0) st.active[r] = active;
        }
    }
    if constexpr (ROWS) return active;
    if (k == 0 && ring_slot >= 0) {
        // ONE 64-bit counter per pass: arrivals in the low word, still-active start points in the high word -- no fence between
        // two counters (a __threadfence() is an L2 write-back here: microseconds per workgroup)
        unsigned long long* cnt = reinterpret_cast<unsigned long long*>(st.nact) + ring_slot;
        const unsigned long long before = atomicAdd(cnt, 1ull + ((unsigned long long)(active ? 1 : 0) << 32));
        if ((unsigned)(before & 0xffffffffull) == (unsigned)R - 1u) {   // the last workgroup of this pass publishes the count and clears the slot
            const unsigned n = (unsigned)(before >> 32) + (active ? 1u : 0u);
            __hip_atomic_store(cnt, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st.ticket, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st.h_cnt + ring_slot, (int)n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (the host reads nothing else on the strength of it: no release, which would be a cache write-back)
        }
    }
    return active;
}
__device__ __forceinline__ int asc_step_one(const AscentState& st, int r, int k, int d, int R, const double* __restrict__ lb,
                                            const double* __restrict__ ub, double ftol_rel, double xtol_abs, int ring_slot) {
    AscStepRegs a;
    asc_step_load<false>(st, r, k, d, R, lb, ub, a);
    const double g_trial = k < d ? st.Gt[(int64_t)r * d + k] : 0.0, ft = st.ft[r];
    return asc_step_compute<false>(st, a, r, k, d, R, ftol_rel, xtol_abs, ring_slot, ft, g_trial);
}
__global__ __launch_bounds__(64) void k_asc_step(AscentState st, int d, int R, const double* __restrict__ lb,
                                                 const double* __restrict__ ub, double first_step_scale, double ftol_rel,
                                                 double xtol_abs, int ring_slot) {
    asc_step_one(st, blockIdx.x, threadIdx.x, d, R, lb, ub, ftol_rel, xtol_abs, ring_slot);
}

// ------------------------------------------------------------------------------------------------------------------------
// SMALL MODELS: the whole ascent of one start point inside ONE workgroup, ONE launch for the whole acquire_max.
// Below N ~ 800 a pass of the free-running form is six kernels of 6-15 us that each move a megabyte or less: 45-49 us per pass
// at N = 500 whatever the arithmetic.  A start point's ascent never looks at another start point, so here workgroup r
// (512 threads) runs start r from the first evaluation to convergence: K* of its trial point, V = W k* (row N of W is
// alpha: mu comes with it), q = sum V^2, U = W'V, the analytic gradient, the L-BFGS step (asc_adopt_one / asc_direction_one /
// asc_step_one on wave 0) -- phases separated by workgroup barriers only, W and W' from L2 (2 x 1 MB at N = 500, shared by
// the R workgroups).  No flags, no atomics, no host decision.  Summation orders differ from the kernels of the batched paths
// (a row of W is one wave's strided sum + butterfly), so values agree with them to rounding, not bit for bit.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int AWG_THREADS = 512, AWG_WAVES = AWG_THREADS / 64, AWG_NMAX = 1024;   // (1024 threads: 128 VGPRs per thread, the inlined two-loop recursion spills)
struct AscWgParams {
    const double *W, *WT, *X, *alpha;   // resident model: W (row N = alpha'), W', observations [N][d], alpha
    int64_t ld, N;
    KernelHyper hp;
    AcqParams ap;
    double beta;
    AscentState st;
    const double *starts, *lb, *ub;
    int R, maxeval;
    double ftol_rel, xtol_abs, first_step_scale;
    unsigned long long max_ticks;       // maxtime in wall_clock64 ticks (0: none)
    int* passes;                        // [R] evaluation passes start r needed
};
template <int DT>
__global__ __launch_bounds__(AWG_THREADS) void k_ascent_wg(AscWgParams p) {
#pragma clang fp contract(off)
    __shared__ double ks[AWG_NMAX + 1], V[AWG_NMAX + 1], U[AWG_NMAX];
    __shared__ double red[AWG_WAVES][2 * DT];
    __shared__ double s_q, s_il2[DT];
    __shared__ int s_go;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = p.hp.d;
    if (tid < DT) s_il2[tid] = tid < d ? p.hp.il2[tid] : 0.0;   // (indexed from the kernel arguments the weights cost > 100 scalar-register spills)
    const int64_t N = p.N, ld = p.ld;
    // The start point's whole state lives in LDS: the bookkeeping functions are the batched drivers' (they take an AscentState of
    // pointers), handed a state whose pointers lead here, with r = 0 of R = 1 start points -- in HBM every one of their dozen
    // dependent loads was a ~1 us round trip, 20 us per pass whatever N.  Only the best point goes back to the global state.
    __shared__ double s_state[(9 + 2 * ASC_M) * DT + 8];
    __shared__ int s_ints[8];
    AscentState st;
    {
        double* q = s_state;
        st.X = q; q += DT; st.G = q; q += DT; st.Xt = q; q += DT; st.Gt = q; q += DT; st.Xn = q; q += DT; st.Gn = q; q += DT;
        st.D = q; q += DT; st.Gp = q; q += DT; st.best_X = q; q += DT;
        st.S = q; q += ASC_M * DT; st.Y = q; q += ASC_M * DT;
        st.f = q++; st.ft = q++; st.fn = q++; st.step = q++; st.best_f = q++;
        st.active = s_ints; st.accepted = s_ints + 1; st.it = s_ints + 2; st.bt = s_ints + 3; st.h_accepted = s_ints + 4; st.h_active = s_ints + 5;
        st.nact = nullptr; st.ticket = nullptr; st.h_cnt = nullptr;
        st.ftol_abs = p.st.ftol_abs; st.xtol_rel = p.st.xtol_rel; st.stopval = p.st.stopval;
    }
    const unsigned long long t_begin = wall_clock64();
    if (tid < 64) {   // k_asc_start
        if (tid < d) {
            const double x = asc_clip(p.starts[(int64_t)r * d + tid], p.lb[tid], p.ub[tid]);
            st.X[tid] = x;
            st.Xt[tid] = x;
        }
        if (tid == 0) { st.it[0] = 0; st.bt[0] = 0; }
    }
    __syncthreads();
    int pass = 0;
    for (;;) {
        // ---- one evaluation of the trial point: value and gradient of the acquisition
        double xs[DT];
#pragma unroll
        for (int k = 0; k < DT; ++k) xs[k] = k < d ? st.Xt[k] : 0.0;
        for (int64_t j = tid; j <= N; j += AWG_THREADS) {
            double rr = 0.0;
#pragma unroll
            for (int k = 0; k < DT; ++k) {
                const double t = ((k < d && j < N) ? p.X[j * d + k] : 0.0) - xs[k];
                rr += s_il2[k] * (t * t);
            }
            ks[j] = j < N ? cov_from_r_fast(p.hp.kern, p.hp.sigma2, rr) : 0.0;
        }
        __syncthreads();
        // V[i] = sum_{k <= i} W[i][k] k*[k]   (row N: alpha).  A wave takes FOUR of its rows at a time (rows i, i + 8, i + 16, i + 24):
        // one row at a time was a chain of dependent L2 round trips, ~0.8 us per row, 50 us per product at N = 500
        for (int64_t i0 = wave; i0 <= N; i0 += 4 * AWG_WAVES) {
            double s[4] = {0.0, 0.0, 0.0, 0.0};
            int64_t kend[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + (int64_t)u * AWG_WAVES;
                kend[u] = i > N ? -1 : (i < N ? i : N - 1);
            }
            int64_t kmax = kend[0];
#pragma unroll
            for (int u = 1; u < 4; ++u) kmax = kend[u] > kmax ? kend[u] : kmax;
#pragma unroll 2
            for (int64_t k = lane; k <= kmax; k += 64) {
                const double kv = ks[k];
                double wv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) wv[u] = k <= kend[u] ? p.W[(i0 + (int64_t)u * AWG_WAVES) * ld + k] : 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u) s[u] += wv[u] * kv;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double t = asc_wsum_dpp(s[u]);
                if (lane == 0 && kend[u] >= 0) V[i0 + (int64_t)u * AWG_WAVES] = t;
            }
        }
        __syncthreads();
        {   // q = sum_{i < N} V[i]^2 in a fixed order
            double s = 0.0;
            for (int64_t i = tid; i < N; i += AWG_THREADS) s += V[i] * V[i];
            s = asc_wsum_dpp(s);
            if (lane == 0) red[wave][0] = s;
        }
        for (int64_t j0 = wave; j0 < N; j0 += 4 * AWG_WAVES) {     // U[j] = sum_{i >= j} W'[j][i] V[i], four rows of W' at a time
            double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
            for (int64_t i = j0 + lane; i < N; i += 64) {   // (row j0 starts first; the later rows skip their first entries)
                const double vv = V[i];
                double wv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t j = j0 + (int64_t)u * AWG_WAVES;
                    wv[u] = (j < N && i >= j) ? p.WT[j * ld + i] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) s[u] += wv[u] * vv;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double t = asc_wsum_dpp(s[u]);
                const int64_t j = j0 + (int64_t)u * AWG_WAVES;
                if (lane == 0 && j < N) U[j] = t;
            }
        }
        __syncthreads();
        if (tid == 0) {
            double q = 0.0;
            for (int wv = 0; wv < AWG_WAVES; ++wv) q += red[wv][0];
            s_q = q;
        }
        __syncthreads();
        const double q = s_q;
        double s2 = p.hp.sigma2 - q;
        if (s2 < 0.0) s2 = 0.0;                              // predict_f: max(sigma2, 0)
        const double mu = p.beta + V[N];
        __syncthreads();                                     // (red is reused below)
        double gm[DT], gv[DT];
#pragma unroll
        for (int k = 0; k < DT; ++k) { gm[k] = 0.0; gv[k] = 0.0; }
        for (int64_t j = tid; j < N; j += AWG_THREADS) {    // gradient: sum_j dk*_j/dx (alpha_j, U_j)   (k_grad_finish)
            double rr = 0.0;
#pragma unroll
            for (int k = 0; k < DT; ++k)
                if (k < d) {
                    const double t = xs[k] - p.X[j * d + k];
                    rr += s_il2[k] * (t * t);
                }
            double fac;
            if (p.hp.kern == KERN_MAT52ARD) {
                const double sq = sqrt(5.0) * sqrt(rr);
                fac = -(5.0 / 3.0) * p.hp.sigma2 * (1.0 + sq) * exp(-sq);
            } else {
                fac = -(p.hp.sigma2 * exp(-0.5 * rr));
            }
            const double a = p.alpha[j], uj = U[j];
#pragma unroll
            for (int k = 0; k < DT; ++k)
                if (k < d) {
                    const double dk = fac * (xs[k] - p.X[j * d + k]) * s_il2[k];
                    gm[k] += dk * a;
                    gv[k] += dk * uj;
                }
        }
#pragma unroll
        for (int k = 0; k < DT; ++k)
            if (k < d) {
                const double a = asc_wsum_dpp(gm[k]), b = asc_wsum_dpp(gv[k]);
                if (lane == 0) { red[wave][2 * k] = a; red[wave][2 * k + 1] = b; }
            }
        __syncthreads();
        if (tid < 64) {
            const int k = tid;
            if (k < d) {
                double a = 0.0, b = 0.0;
                for (int wv = 0; wv < AWG_WAVES; ++wv) { a += red[wv][2 * k]; b += red[wv][2 * k + 1]; }
                double dmu, ds2;
                acq_partials(p.ap, mu, s2, dmu, ds2);
                st.Gt[k] = dmu * a + (s2 > 0.0 ? ds2 * (-2.0 * b) : 0.0);
            }
            if (k == 0) st.ft[0] = acq_eval(p.ap, mu, s2);
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");   // (LDS state: lane 0's scalars are visible to the wave)
            // ---- the ascent's bookkeeping for this start point (same code as the batched drivers, on the LDS state)
            if (pass == 0) {
                asc_adopt_one(st, 0, k, d);
                asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
                asc_direction_one(st, 0, k, d, 1, 0, 0, p.lb, p.ub, p.first_step_scale);
            } else {
                asc_step_one(st, 0, k, d, 1, p.lb, p.ub, p.ftol_rel, p.xtol_abs, -1);
            }
            asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
            if (k == 0) {
                const bool time_up = p.max_ticks != 0ull && wall_clock64() - t_begin > p.max_ticks;
                s_go = (st.active[0] != 0 && pass + 1 < p.maxeval && !time_up) ? 1 : 0;
            }
        }
        ++pass;
        __syncthreads();
        if (!s_go) break;
    }
    if (tid < d) p.st.best_X[(int64_t)r * d + tid] = st.best_X[tid];
    if (tid == 0) { p.st.best_f[r] = st.best_f[0]; p.passes[r] = pass; }
}

// (value desc, index asc) over best_f; NaN never wins.  One workgroup.
// packed != nullptr: everything the caller reads back in ONE block -- [best value, best index, best_x (d), f (R), X (R d), passes (R)]
// (eight pageable copies per acquire_max were 0.1 ms of a 0.27 ms call on a small model)
__global__ __launch_bounds__(256) void k_asc_final(AscentState st, int d, int R, Best* __restrict__ best,
                                                   double* __restrict__ best_x, double* __restrict__ packed = nullptr,
                                                   const int* __restrict__ passes = nullptr) {
    __shared__ double sv[256];
    __shared__ long long si[256];
    double v = -INFINITY;
    long long idx = -1;
    for (int r = threadIdx.x; r < R; r += 256) {
        const double f = st.best_f[r];
        if (f > -INFINITY && (idx < 0 || f > v)) { v = f; idx = r; }   // ascending r within a thread: first maximum wins
    }
    sv[threadIdx.x] = v;
    si[threadIdx.x] = idx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const double ov = sv[threadIdx.x + o];
            const long long oi = si[threadIdx.x + o];
            const double mv = sv[threadIdx.x];
            const long long mi = si[threadIdx.x];
            if (oi >= 0 && (mi < 0 || ov > mv || (ov == mv && oi < mi))) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    const long long w = si[0];
    if (threadIdx.x == 0) { best->val = w >= 0 ? sv[0] : -INFINITY; best->idx = w; }
    if (w >= 0 && (int)threadIdx.x < d) best_x[threadIdx.x] = st.best_X[w * d + threadIdx.x];
    if (packed) {
        if (threadIdx.x == 0) { packed[0] = w >= 0 ? sv[0] : -INFINITY; packed[1] = (double)w; }
        if ((int)threadIdx.x < d) packed[2 + threadIdx.x] = w >= 0 ? st.best_X[w * d + threadIdx.x] : 0.0;
        for (int r = threadIdx.x; r < R; r += 256) {
            packed[2 + d + r] = st.best_f[r];
            packed[2 + d + R + (int64_t)R * d + r] = passes ? (double)passes[r] : 0.0;
        }
        for (int64_t e = threadIdx.x; e < (int64_t)R * d; e += 256) packed[2 + d + R + e] = st.best_X[e];
    }
}

}  // namespace bohip
