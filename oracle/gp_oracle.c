/*
 * gp_oracle.c -- CPU restatement (plain C, float64) of the GP-posterior + acquisition
 * hot path of jbrea/BayesianOptimization.jl v0.2.5.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the shipped path (libbohip.so) never does.
 *
 * PARITY UNPINNED: the reference is a Julia package whose GP arithmetic lives in the
 * un-vendored GaussianProcesses.jl (compat 0.9-0.12) / ElasticPDMats.jl (0.2.3) packages
 * (reference Project.toml:10,12,22,24; no Manifest).  There is no Julia toolchain in this
 * image, so neither the reference nor those packages can be run to produce golden vectors,
 * and the reference's own tests hold no numeric fixtures.  What IS
 * pinned: every formula that lives in /root/reference itself is restated verbatim below
 * (operation order kept, no FMA contraction: compile with -ffp-contract=off), and the
 * reference's known-answer tests (test/acquisition.jl:11-12, test/acquisitionfunctions.jl:8-11,
 * test/warmstart.jl:64) are re-run against this oracle in tests/test_oracle.py.
 * GaussianProcesses.jl behaviour is restated from its published algorithm; the three
 * unverifiable details are behind named switches (ORACLE_NOISE_EPS, ORACLE_CLAMP_VAR,
 * direct weighted squared distance instead of the Gram trick).
 *
 * Layouts follow Julia: X is d x N column-major (observation i = d contiguous doubles).
 * The Cholesky factor is kept as the column-major UPPER factor U (Julia/ElasticPDMats
 * convention, cK = U'U); the same bytes read row-major are the LOWER factor L = U'.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_NOISE_EPS 2.220446049250313e-16 /* GaussianProcesses.jl update_cK!: exp(2*logNoise)+eps() [UPSTREAM-UNVERIFIED] */
#define ORACLE_CLAMP_VAR 1                     /* predict_f: sigma2 = max(sigma2, 0.0) [implied by acquisitionfunctions.jl:25,48] */

enum { ACQ_EI = 0, ACQ_PI = 1, ACQ_UCB = 2, ACQ_MI = 3, ACQ_MAXMEAN = 4 };
enum { KERN_SEARD = 0, KERN_SEISO = 1, KERN_MAT52ARD = 2 };

/* ---- src/utils.jl:48-49 (verbatim operation order) ---------------------------------- */
/* normal_pdf(mu, s2) = 1 / sqrt(2pi * s2) * exp(-mu^2 / (2 * s2)) */
double oracle_normal_pdf(double mu, double s2) {
    return 1.0 / sqrt(2.0 * M_PI * s2) * exp(-(mu * mu) / (2.0 * s2));
}
/* normal_cdf(mu, s2) = 1 / 2 * (1 + erf(mu / sqrt(2 s2))) */
double oracle_normal_cdf(double mu, double s2) {
    return 1.0 / 2.0 * (1.0 + erf(mu / sqrt(2.0 * s2)));
}

/* ---- src/acquisitionfunctions.jl functors ------------------------------------------- */
/* ExpectedImprovement :47-50.  NOTE: reference computes D*Phi + sqrt(s2)*pdf(D,s2) = D*Phi(z)+phi(z),
 * not the textbook D*Phi(z)+sigma*phi(z); reproduced as written. */
double oracle_ei(double mu, double s2, double tau) {
    if (s2 == 0.0) return mu > tau ? mu - tau : 0.0;
    return (mu - tau) * oracle_normal_cdf(mu - tau, s2) + sqrt(s2) * oracle_normal_pdf(mu - tau, s2);
}
/* ProbabilityOfImprovement :24-27 */
double oracle_pi(double mu, double s2, double tau) {
    if (s2 == 0.0) return mu > tau ? 1.0 : 0.0;
    return oracle_normal_cdf(mu - tau, s2);
}
/* UpperConfidenceBound :96 */
double oracle_ucb(double mu, double s2, double beta_t) { return mu + beta_t * sqrt(s2); }
/* MutualInformation :141 */
double oracle_mi(double mu, double s2, double sqrt_alpha, double gamma_hat) {
    return mu + sqrt_alpha * (sqrt(s2 + gamma_hat) - sqrt(gamma_hat));
}
/* BrochuBetaScaling setparams! :91-95: beta_t = sqrt(2*log(nobs^(D/2+2)*pi^2/(3*delta))) */
double oracle_brochu_beta(int64_t D, int64_t nobs, double delta) {
    if (nobs == 0) nobs = 1;
    return sqrt(2.0 * log(pow((double)nobs, (double)D / 2.0 + 2.0) * (M_PI * M_PI) / (3.0 * delta)));
}
/* generic dispatcher; params: EI/PI {tau}; UCB {beta_t}; MI {sqrt_alpha, gamma_hat}; MaxMean {} (:111) */
double oracle_acq(int acq, const double *p, double mu, double s2) {
    switch (acq) {
    case ACQ_EI: return oracle_ei(mu, s2, p[0]);
    case ACQ_PI: return oracle_pi(mu, s2, p[0]);
    case ACQ_UCB: return oracle_ucb(mu, s2, p[0]);
    case ACQ_MI: return oracle_mi(mu, s2, p[0], p[1]);
    default: return mu;
    }
}

/* ---- covariance functions (GaussianProcesses.jl SEArd / SEIso / Mat52Ard; used at
 *      README.md:24, test/branin.jl:25, test/acquisition.jl:2, BayesianOptimization.jl:259-262)
 * hyper layout: loglen[d] (SEIso uses loglen[0] for all dims), logsig.
 * SEArd:    k = s2 * exp(-0.5 * r),  r = sum_k il2_k (x_k - y_k)^2, il2 = exp(-2 ll), s2 = exp(2 lsig)
 * Mat52Ard: k = s2 * (1 + sqrt(5) R + 5/3 R^2) * exp(-sqrt(5) R), R = sqrt(r)               */
static double wsqdist(int64_t d, const double *x, const double *y, const double *il2) {
    double r = 0.0;
    for (int64_t k = 0; k < d; ++k) {
        double t = x[k] - y[k];
        r += il2[k] * (t * t);
    }
    return r;
}
static double cov_from_r(int kern, double s2, double r) {
    if (kern == KERN_MAT52ARD) {
        double R = sqrt(r), s = sqrt(5.0) * R;
        return s2 * (1.0 + s + 5.0 / 3.0 * r) * exp(-s);
    }
    return s2 * exp(-0.5 * r);
}
void oracle_il2(int kern, int64_t d, const double *loglen, double *il2) {
    for (int64_t k = 0; k < d; ++k) il2[k] = exp(-2.0 * loglen[kern == KERN_SEISO ? 0 : k]);
}

/* A1: cK = K + (exp(2 logNoise) + eps) I.  Output: full symmetric N x N (col-major == row-major). */
void oracle_build_cK(int kern, int64_t d, int64_t N, const double *X, const double *loglen, double logsig,
                     double lognoise, double *cK) {
    double *il2 = (double *)malloc(sizeof(double) * d);
    oracle_il2(kern, d, loglen, il2);
    double s2 = exp(2.0 * logsig), noise = exp(2.0 * lognoise) + ORACLE_NOISE_EPS;
    for (int64_t i = 0; i < N; ++i)
        for (int64_t j = 0; j <= i; ++j) {
            double v = cov_from_r(kern, s2, wsqdist(d, X + d * i, X + d * j, il2));
            if (i == j) v += noise;
            cK[i * N + j] = v;
            cK[j * N + i] = v;
        }
    free(il2);
}

/* A2: Cholesky cK = L L' (row-major lower L == col-major upper U, the ElasticPDMats storage).
 * Row-by-row (Cholesky-Banachiewicz), dot products in index order.  In-place on the lower
 * triangle of A (ld = row stride); the strict upper triangle is zeroed.
 * Returns 0, or the 1-based index of the first non-positive pivot (LAPACK potrf convention). */
int64_t oracle_cholesky(int64_t N, double *A, int64_t ld) {
    for (int64_t i = 0; i < N; ++i) {
        double *Li = A + i * ld;
        for (int64_t j = 0; j <= i; ++j) {
            const double *Lj = A + j * ld;
            double s = Li[j];
            for (int64_t k = 0; k < j; ++k) s -= Li[k] * Lj[k];
            if (i == j) {
                if (!(s > 0.0)) return i + 1;
                Li[j] = sqrt(s);
            } else {
                Li[j] = s / Lj[j];
            }
        }
        for (int64_t j = i + 1; j < N; ++j) Li[j] = 0.0;
    }
    return 0;
}

/* A2': incremental append of p points (ElasticPDMats append!: U12 = U11' \ K12,
 * U22 = chol(K22 - U12'U12)).  L is (N+p) x (N+p) row-major with ld; rows [0,N) hold the old
 * factor; rows [N,N+p) hold, on entry, the new rows of cK (columns 0..N+p-1, lower part used).   */
int64_t oracle_cholesky_append(int64_t N, int64_t p, double *L, int64_t ld) {
    for (int64_t i = N; i < N + p; ++i) {
        double *Li = L + i * ld;
        for (int64_t j = 0; j <= i; ++j) {
            const double *Lj = L + j * ld;
            double s = Li[j];
            for (int64_t k = 0; k < j; ++k) s -= Li[k] * Lj[k];
            if (i == j) {
                if (!(s > 0.0)) return i + 1;
                Li[j] = sqrt(s);
            } else {
                Li[j] = s / Lj[j];
            }
        }
        for (int64_t j = i + 1; j < N + p; ++j) Li[j] = 0.0;
    }
    return 0;
}

/* forward substitution  L v = b  (v may alias b) */
void oracle_trsv_lower(int64_t N, const double *L, int64_t ld, double *v) {
    for (int64_t i = 0; i < N; ++i) {
        const double *Li = L + i * ld;
        double s = v[i];
        for (int64_t k = 0; k < i; ++k) s -= Li[k] * v[k];
        v[i] = s / Li[i];
    }
}
/* back substitution  L' a = v */
void oracle_trsv_lower_t(int64_t N, const double *L, int64_t ld, double *v) {
    for (int64_t i = N - 1; i >= 0; --i) {
        double s = v[i];
        for (int64_t k = i + 1; k < N; ++k) s -= L[k * ld + i] * v[k];
        v[i] = s / L[i * ld + i];
    }
}
/* A3: alpha = cK^{-1} (y - beta) */
void oracle_alpha(int64_t N, const double *L, int64_t ld, const double *y, double beta, double *alpha) {
    for (int64_t i = 0; i < N; ++i) alpha[i] = y[i] - beta;
    oracle_trsv_lower(N, L, ld, alpha);
    oracle_trsv_lower_t(N, L, ld, alpha);
}

/* A4: predict_f at R candidates, one column at a time exactly as
 * mean_var(model, x::Vector) (src/models/gp.jl:2-5) / predict_f(full_cov=false) does:
 *   k*_i = cov(X_i, x*);  mu = beta + k*' alpha;  v = L \ k*;  s2 = max(k(x*,x*) - v'v, 0).
 * work: scratch of N doubles per thread.  nthreads<=1 -> sequential (the reference is single-threaded). */
static void predict_one(int kern, int64_t d, int64_t N, const double *X, const double *il2, double s2f,
                        double beta, const double *L, int64_t ld, const double *alpha, const double *xs,
                        double *work, double *mu, double *var) {
    double m = 0.0;
    for (int64_t i = 0; i < N; ++i) {
        work[i] = cov_from_r(kern, s2f, wsqdist(d, X + d * i, xs, il2));
        m += work[i] * alpha[i];
    }
    *mu = beta + m;
    oracle_trsv_lower(N, L, ld, work);
    double q = 0.0;
    for (int64_t i = 0; i < N; ++i) q += work[i] * work[i];
    double s = s2f - q; /* k(x*,x*) = s2f * f(0) = s2f for all three kernels */
#if ORACLE_CLAMP_VAR
    if (s < 0.0) s = 0.0;
#endif
    *var = s;
}
void oracle_predict(int kern, int64_t d, int64_t N, const double *X, const double *loglen, double logsig,
                    double beta, const double *L, int64_t ld, const double *alpha, const double *Xs,
                    int64_t R, double *mu, double *var, int nthreads) {
    double *il2 = (double *)malloc(sizeof(double) * d);
    oracle_il2(kern, d, loglen, il2);
    double s2f = exp(2.0 * logsig);
    if (nthreads <= 1) {
        double *work = (double *)malloc(sizeof(double) * (N > 0 ? N : 1));
        for (int64_t r = 0; r < R; ++r)
            predict_one(kern, d, N, X, il2, s2f, beta, L, ld, alpha, Xs + d * r, work, mu + r, var + r);
        free(work);
    } else {
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
        {
            double *work = (double *)malloc(sizeof(double) * (N > 0 ? N : 1));
#pragma omp for schedule(static)
            for (int64_t r = 0; r < R; ++r)
                predict_one(kern, d, N, X, il2, s2f, beta, L, ld, alpha, Xs + d * r, work, mu + r, var + r);
            free(work);
        }
#endif
    }
    free(il2);
}

/* Full posterior covariance: predict_f(gp, X; full_cov = true), the input of the reference's joint draw
 * myrand(model, X::Matrix) = rand(gp, X) (src/models/gp.jl:7):  cov = K** - V'V, V = L^-1 K*, no clamp. */
void oracle_predict_cov(int kern, int64_t d, int64_t N, const double *X, const double *loglen, double logsig,
                        double beta, const double *L, int64_t ld, const double *alpha, const double *Xs,
                        int64_t R, double *mu, double *cov) {
    double *il2 = (double *)malloc(sizeof(double) * d);
    double *V = (double *)malloc(sizeof(double) * (N > 0 ? N : 1) * (R > 0 ? R : 1));
    oracle_il2(kern, d, loglen, il2);
    const double s2f = exp(2.0 * logsig);
    for (int64_t r = 0; r < R; ++r) {
        double *v = V + r * N, m = 0.0;
        for (int64_t i = 0; i < N; ++i) {
            v[i] = cov_from_r(kern, s2f, wsqdist(d, X + d * i, Xs + d * r, il2));
            m += v[i] * alpha[i];
        }
        mu[r] = beta + m;
        oracle_trsv_lower(N, L, ld, v);
    }
    for (int64_t r = 0; r < R; ++r)
        for (int64_t s = 0; s <= r; ++s) {
            double q = 0.0;
            for (int64_t i = 0; i < N; ++i) q += V[r * N + i] * V[s * N + i];
            const double c = cov_from_r(kern, s2f, wsqdist(d, Xs + d * r, Xs + d * s, il2)) - q;
            cov[r * R + s] = c;
            cov[s * R + r] = c;
        }
    free(il2);
    free(V);
}

/* A4-A7 fused the way acquire_max would see it if every start point were scored as-is:
 * score every column, keep the best with strict '>' starting from -Inf (src/acquisition.jl:55,62-65)
 * => first maximum wins ties, NaN never wins.  best_idx = -1 if nothing beat -Inf. */
void oracle_score(int kern, int64_t d, int64_t N, const double *X, const double *loglen, double logsig,
                  double beta, const double *L, int64_t ld, const double *alpha, int acq,
                  const double *acq_params, const double *Xs, int64_t R, double *score, double *best_val,
                  int64_t *best_idx, int nthreads) {
    double *mu = (double *)malloc(sizeof(double) * (R > 0 ? R : 1));
    double *var = (double *)malloc(sizeof(double) * (R > 0 ? R : 1));
    oracle_predict(kern, d, N, X, loglen, logsig, beta, L, ld, alpha, Xs, R, mu, var, nthreads);
    double maxf = -INFINITY;
    int64_t maxi = -1;
    for (int64_t r = 0; r < R; ++r) {
        double f = oracle_acq(acq, acq_params, mu[r], var[r]);
        if (score) score[r] = f;
        if (f > maxf) {
            maxf = f;
            maxi = r;
        }
    }
    *best_val = maxf;
    *best_idx = maxi;
    free(mu);
    free(var);
}

/* A8: value and analytic gradient of (mu, s2) w.r.t. x* for SEArd/SEIso (the role of
 * ForwardDiff in wrap_gradient, src/acquisition.jl:11-17):
 *   dk_i/dx_k = -k_i (x_k - X_ki) il2_k;  dmu = (dk)'alpha;  ds2 = -2 (dk)' L^-T v.       */
/* d k(r)/d x_k = fac(r) * il2_k * (x_k - X_ik):  SE: fac = -k;  Mat52: fac = -(5/3) s2 (1 + s) exp(-s), s = sqrt(5 r) */
static double cov_grad_fac(int kern, double s2, double r) {
    if (kern == KERN_MAT52ARD) {
        double s = sqrt(5.0) * sqrt(r);
        return -(5.0 / 3.0) * s2 * (1.0 + s) * exp(-s);
    }
    return -(s2 * exp(-0.5 * r));
}
void oracle_predict_grad(int kern, int64_t d, int64_t N, const double *X, const double *il2, double s2f, double beta,
                         const double *L, int64_t ld, const double *alpha, const double *xs, double *mu,
                         double *var, double *dmu, double *dvar) {
    double *ks = (double *)malloc(sizeof(double) * N), *u = (double *)malloc(sizeof(double) * N);
    double *fac = (double *)malloc(sizeof(double) * N);
    double m = 0.0;
    for (int64_t i = 0; i < N; ++i) {
        double r = wsqdist(d, X + d * i, xs, il2);
        ks[i] = cov_from_r(kern, s2f, r);
        fac[i] = cov_grad_fac(kern, s2f, r);
        m += ks[i] * alpha[i];
        u[i] = ks[i];
    }
    *mu = beta + m;
    oracle_trsv_lower(N, L, ld, u);
    double q = 0.0;
    for (int64_t i = 0; i < N; ++i) q += u[i] * u[i];
    double s = s2f - q;
    int clamped = 0;
#if ORACLE_CLAMP_VAR
    if (s < 0.0) { s = 0.0; clamped = 1; }
#endif
    *var = s;
    oracle_trsv_lower_t(N, L, ld, u); /* u = cK^{-1} k* */
    for (int64_t k = 0; k < d; ++k) {
        double gm = 0.0, gv = 0.0;
        for (int64_t i = 0; i < N; ++i) {
            double dk = fac[i] * (xs[k] - X[d * i + k]) * il2[k];
            gm += dk * alpha[i];
            gv += dk * u[i];
        }
        dmu[k] = gm;
        dvar[k] = clamped ? 0.0 : -2.0 * gv;
    }
    free(ks);
    free(u);
    free(fac);
}
/* d(score)/d(mu), d(score)/d(s2) of the REFERENCE's formulas (not textbook EI).
 * EI_ref = D*Phi(z) + phi(z), z = D/sqrt(s2):  dEI/dmu = Phi(z) + (D - z)/sqrt(s2) * phi(z)... expanded below. */
void oracle_acq_partials(int acq, const double *p, double mu, double s2, double *dmu, double *ds2) {
    const double inv_sqrt_2pi = 0.3989422804014327;
    switch (acq) {
    case ACQ_EI: {
        if (s2 == 0.0) { *dmu = mu > p[0] ? 1.0 : 0.0; *ds2 = 0.0; return; }
        double D = mu - p[0], s = sqrt(s2), z = D / s;
        double Phi = 0.5 * (1.0 + erf(z / sqrt(2.0))), phi = inv_sqrt_2pi * exp(-0.5 * z * z);
        /* f = D Phi(z) + phi(z); dz/dmu = 1/s; dz/ds2 = -z/(2 s2); phi'(z) = -z phi */
        *dmu = Phi + D * phi / s - z * phi / s;
        *ds2 = (D * phi - z * phi) * (-z / (2.0 * s2));
        return;
    }
    case ACQ_PI: {
        if (s2 == 0.0) { *dmu = 0.0; *ds2 = 0.0; return; }
        double D = mu - p[0], s = sqrt(s2), z = D / s, phi = inv_sqrt_2pi * exp(-0.5 * z * z);
        *dmu = phi / s;
        *ds2 = phi * (-z / (2.0 * s2));
        return;
    }
    case ACQ_UCB: *dmu = 1.0; *ds2 = s2 > 0.0 ? p[0] / (2.0 * sqrt(s2)) : 0.0; return;
    case ACQ_MI: *dmu = 1.0; *ds2 = p[0] / (2.0 * sqrt(s2 + p[1])); return;
    default: *dmu = 1.0; *ds2 = 0.0; return;
    }
}
void oracle_score_grad(int kern, int64_t d, int64_t N, const double *X, const double *loglen, double logsig,
                       double beta, const double *L, int64_t ld, const double *alpha, int acq,
                       const double *acq_params, const double *Xs, int64_t R, double *score, double *grad) {
    double *il2 = (double *)malloc(sizeof(double) * d);
    oracle_il2(kern, d, loglen, il2);
    double s2f = exp(2.0 * logsig);
    double *gm = (double *)malloc(sizeof(double) * d), *gv = (double *)malloc(sizeof(double) * d);
    for (int64_t r = 0; r < R; ++r) {
        double mu, var, a, b;
        oracle_predict_grad(kern, d, N, X, il2, s2f, beta, L, ld, alpha, Xs + d * r, &mu, &var, gm, gv);
        score[r] = oracle_acq(acq, acq_params, mu, var);
        oracle_acq_partials(acq, acq_params, mu, var, &a, &b);
        for (int64_t k = 0; k < d; ++k) grad[d * r + k] = a * gm[k] + b * gv[k];
    }
    free(il2); free(gm); free(gv);
}

/* A9 (C5 form): independent draws mu_j + sigma_j z_sj, arg-max per draw (strict '>', first wins).
 * z is S x R row-major, supplied by the caller so any shard can reproduce it. */
void oracle_thompson(int64_t S, int64_t R, const double *mu, const double *var, const double *z,
                     double *best_val, int64_t *best_idx) {
    for (int64_t s = 0; s < S; ++s) {
        double maxf = -INFINITY;
        int64_t maxi = -1;
        for (int64_t r = 0; r < R; ++r) {
            double f = mu[r] + sqrt(var[r]) * z[s * R + r];
            if (f > maxf) { maxf = f; maxi = r; }
        }
        best_val[s] = maxf;
        best_idx[s] = maxi;
    }
}

/* N2: log marginal likelihood and its analytic gradient w.r.t. (logNoise, beta, kernel log-parameters) -- the role of
 * GaussianProcesses.jl update_target_and_dtarget! called from optimizemodel!, reference src/models/gp.jl:59-64
 * (the package source is absent from /root/reference; this is the textbook expression, Rasmussen & Williams eq. 5.9):
 *     mll = -1/2 r'alpha - sum_i log L_ii - N/2 log(2 pi),  r = y - beta,  alpha = cK^-1 r
 *     d mll / d theta = 1/2 tr((alpha alpha' - cK^-1) dcK/dtheta)
 *     dcK/dlogNoise = 2 exp(2 logNoise) I;  d mll / d beta = sum_i alpha_i
 *     SE:    dK_ij/dll_k = K_ij t_k,                      t_k = il2_k (x_ik - x_jk)^2   (SEIso: sum over k)
 *     Mat52: dK_ij/dll_k = 5/3 s2 (1 + s) exp(-s) t_k,    s = sqrt(5 r);     dK_ij/dlogsig = 2 K_ij
 * grad layout: [dlogNoise, dbeta, dll_0 .. dll_{nl-1}, dlogsig], nl = d (Ard) or 1 (Iso).  cK^-1 is formed explicitly
 * column by column (two triangular solves per unit vector): O(N^3), small cases only.
 * Returns 0 or the failing Cholesky pivot. */
int64_t oracle_mll_grad(int kern, int64_t d, int64_t N, const double *X, const double *y, const double *loglen,
                        double logsig, double lognoise, double beta, double *mll, double *grad) {
    double *cK = (double *)malloc(sizeof(double) * N * N), *Ki = (double *)malloc(sizeof(double) * N * N);
    double *alpha = (double *)malloc(sizeof(double) * N), *il2 = (double *)malloc(sizeof(double) * d);
    double *e = (double *)malloc(sizeof(double) * N);
    oracle_build_cK(kern, d, N, X, loglen, logsig, lognoise, cK);
    int64_t info = oracle_cholesky(N, cK, N);
    if (info == 0) {
        oracle_alpha(N, cK, N, y, beta, alpha);
        double m = 0.0;
        for (int64_t i = 0; i < N; ++i) m += -0.5 * (y[i] - beta) * alpha[i] - log(cK[i * N + i]);
        *mll = m - 0.5 * (double)N * log(2.0 * M_PI);
        for (int64_t c = 0; c < N; ++c) {
            for (int64_t i = 0; i < N; ++i) e[i] = (i == c) ? 1.0 : 0.0;
            oracle_trsv_lower(N, cK, N, e);
            oracle_trsv_lower_t(N, cK, N, e);
            for (int64_t i = 0; i < N; ++i) Ki[i * N + c] = e[i];
        }
        oracle_il2(kern, d, loglen, il2);
        const int64_t nl = (kern == KERN_SEISO) ? 1 : d;
        const double s2 = exp(2.0 * logsig);
        for (int64_t k = 0; k < nl + 3; ++k) grad[k] = 0.0;
        for (int64_t i = 0; i < N; ++i) {
            grad[1] += alpha[i];
            grad[0] += 0.5 * (alpha[i] * alpha[i] - Ki[i * N + i]) * 2.0 * exp(2.0 * lognoise);
            for (int64_t j = 0; j <= i; ++j) {
                const double G = (alpha[i] * alpha[j] - Ki[i * N + j]) * (i == j ? 0.5 : 1.0);
                const double r = wsqdist(d, X + d * i, X + d * j, il2);
                const double Kij = cov_from_r(kern, s2, r), fac = -cov_grad_fac(kern, s2, r);
                for (int64_t k = 0; k < d; ++k) {
                    const double t = X[d * i + k] - X[d * j + k];
                    grad[2 + (kern == KERN_SEISO ? 0 : k)] += G * fac * (il2[k] * (t * t));
                }
                grad[2 + nl] += G * 2.0 * Kij;
            }
        }
    }
    free(cK); free(Ki); free(alpha); free(il2); free(e);
    return info;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
