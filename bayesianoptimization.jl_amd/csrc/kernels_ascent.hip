// kernels_ascent.hip -- SURVEY.md section 8(f) N1, second part: the whole ascent of a small model in ONE launch (k_ascent_wg) and the
// arg-max over the start points (k_asc_final).  State, step functions and the batched kernels: kernels_ascent_step.hip.
#include "common.h"

namespace bohip {

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
