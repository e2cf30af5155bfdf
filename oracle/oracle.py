"""CPU oracle for the GP-posterior + acquisition hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (libbohip.so + the host mirror in bayesianoptimization.jl_amd/) never does.

PARITY UNPINNED (see gp_oracle.c header and DESIGN.md): the reference is Julia, its GP
arithmetic lives in un-vendored GaussianProcesses.jl / ElasticPDMats.jl, and no Julia exists
in this image, so no golden vectors can be produced by the reference itself.  Three
independent restatements are kept so they can at least pin each other:

  * ``COracle``   -- ctypes view of gp_oracle.c (loop order of the Julia generic code;
                     also the timed ``cpu_baseline`` of bench.py),
  * ``NumpyGP``   -- NumPy/SciPy (LAPACK potrf/trtrs = what Julia's LinearAlgebra calls),
  * ``mp_*``      -- mpmath >= 50 digits, used only to bound rounding error of the two above.

Reference call sites restated: src/models/gp.jl:2-18, src/acquisitionfunctions.jl:4-9,24-27,
44-50,91-96,108,111,131-141, src/utils.jl:48-49,101-120, src/acquisition.jl:54-68.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ACQ = {"EI": 0, "PI": 1, "UCB": 2, "MI": 3, "MaxMean": 4}
KERN = {"SEArd": 0, "SEIso": 1, "Mat52Ard": 2}
NOISE_EPS = float(np.finfo(np.float64).eps)  # GaussianProcesses.jl: exp(2 logNoise) + eps()  [UPSTREAM-UNVERIFIED]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "gp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


class COracle:
    """ctypes binding of gp_oracle.c.  X is d x N column-major (Julia) == (N, d) C-contiguous NumPy."""

    def __init__(self):
        self.lib = C.CDLL(build())
        L = self.lib
        L.oracle_cholesky.restype = C.c_int64
        L.oracle_cholesky_append.restype = C.c_int64
        for f in ("oracle_normal_pdf", "oracle_normal_cdf"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_double, C.c_double]
        L.oracle_acq.restype = C.c_double
        L.oracle_acq.argtypes = [C.c_int, _dp, C.c_double, C.c_double]
        L.oracle_brochu_beta.restype = C.c_double
        L.oracle_brochu_beta.argtypes = [C.c_int64, C.c_int64, C.c_double]

    # -- scalar formulas -------------------------------------------------------------------
    def acq(self, name, params, mu, s2):
        p = np.ascontiguousarray(np.atleast_1d(np.asarray(params, dtype=np.float64)))
        if p.size == 0:
            p = np.zeros(1)
        return self.lib.oracle_acq(ACQ[name], _p(p), float(mu), float(s2))

    def brochu_beta(self, D, nobs, delta=0.1):
        return self.lib.oracle_brochu_beta(int(D), int(nobs), float(delta))

    # -- model build -----------------------------------------------------------------------
    def build_cK(self, X, loglen, logsig, lognoise, kern="SEArd"):
        X = np.ascontiguousarray(X, dtype=np.float64)
        N, d = X.shape
        ll = np.ascontiguousarray(np.broadcast_to(np.asarray(loglen, dtype=np.float64), (d,)))
        cK = np.empty((N, N))
        self.lib.oracle_build_cK(C.c_int(KERN[kern]), C.c_int64(d), C.c_int64(N), _p(X), _p(ll),
                                 C.c_double(logsig), C.c_double(lognoise), _p(cK))
        return cK

    def cholesky(self, cK):
        L = np.array(cK, dtype=np.float64, order="C")
        N = L.shape[0]
        info = self.lib.oracle_cholesky(C.c_int64(N), _p(L), C.c_int64(N))
        if info != 0:
            raise np.linalg.LinAlgError(f"not positive definite at pivot {info}")
        return L

    def cholesky_append(self, Lold, newrows):
        """Lold: N x N lower factor; newrows: p x (N+p) rows of cK (lower part used)."""
        N = Lold.shape[0]
        p = newrows.shape[0]
        L = np.zeros((N + p, N + p))
        L[:N, :N] = Lold
        L[N:, :] = newrows
        info = self.lib.oracle_cholesky_append(C.c_int64(N), C.c_int64(p), _p(L), C.c_int64(N + p))
        if info != 0:
            raise np.linalg.LinAlgError(f"not positive definite at pivot {info}")
        return L

    def alpha(self, L, y, beta):
        y = np.ascontiguousarray(y, dtype=np.float64)
        a = np.empty_like(y)
        N = L.shape[0]
        self.lib.oracle_alpha(C.c_int64(N), _p(L), C.c_int64(L.shape[1]), _p(y), C.c_double(beta), _p(a))
        return a

    def mll_grad(self, X, y, loglen, logsig, lognoise, beta, kern="SEArd"):
        """(mll, grad) with grad = [dlogNoise, dbeta, dll..., dlogsig] (O(N^3) explicit inverse: small N only)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        N, d = X.shape
        ll = np.ascontiguousarray(np.broadcast_to(np.asarray(loglen, dtype=np.float64), (d,)))
        nl = 1 if kern == "SEIso" else d
        grad = np.empty(nl + 3)
        mll = C.c_double(0.0)
        self.lib.oracle_mll_grad.restype = C.c_int64
        info = self.lib.oracle_mll_grad(C.c_int(KERN[kern]), C.c_int64(d), C.c_int64(N), _p(X), _p(y), _p(ll),
                                        C.c_double(logsig), C.c_double(lognoise), C.c_double(beta), C.byref(mll), _p(grad))
        if info != 0:
            raise np.linalg.LinAlgError(f"not positive definite at pivot {info}")
        return mll.value, grad

    def fit(self, X, y, loglen, logsig, lognoise, beta, kern="SEArd"):
        cK = self.build_cK(X, loglen, logsig, lognoise, kern)
        L = self.cholesky(cK)
        return L, self.alpha(L, y, beta)

    # -- posterior / scoring ---------------------------------------------------------------
    def predict(self, X, loglen, logsig, beta, L, alpha, Xs, kern="SEArd", nthreads=1):
        X = np.ascontiguousarray(X, dtype=np.float64)
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
        N, d = X.shape
        R = Xs.shape[0]
        ll = np.ascontiguousarray(np.broadcast_to(np.asarray(loglen, dtype=np.float64), (d,)))
        mu = np.empty(R)
        var = np.empty(R)
        self.lib.oracle_predict(C.c_int(KERN[kern]), C.c_int64(d), C.c_int64(N), _p(X), _p(ll),
                                C.c_double(logsig), C.c_double(beta), _p(L), C.c_int64(L.shape[1]),
                                _p(alpha), _p(Xs), C.c_int64(R), _p(mu), _p(var), C.c_int(nthreads))
        return mu, var

    def predict_cov(self, X, loglen, logsig, beta, L, alpha, Xs, kern="SEArd"):
        """(mu, cov) with the full R x R posterior covariance (the joint-draw input of myrand(model, X::Matrix))."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
        N, d = X.shape
        R = Xs.shape[0]
        ll = np.ascontiguousarray(np.broadcast_to(np.asarray(loglen, dtype=np.float64), (d,)))
        mu = np.empty(R)
        cov = np.empty((R, R))
        self.lib.oracle_predict_cov.restype = None
        self.lib.oracle_predict_cov(C.c_int(KERN[kern]), C.c_int64(d), C.c_int64(N), _p(X), _p(ll), C.c_double(logsig),
                                    C.c_double(beta), _p(L), C.c_int64(L.shape[1]), _p(alpha), _p(Xs), C.c_int64(R),
                                    _p(mu), _p(cov))
        return mu, cov

    def score(self, X, loglen, logsig, beta, L, alpha, acq, params, Xs, kern="SEArd", nthreads=1):
        X = np.ascontiguousarray(X, dtype=np.float64)
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
        N, d = X.shape
        R = Xs.shape[0]
        ll = np.ascontiguousarray(np.broadcast_to(np.asarray(loglen, dtype=np.float64), (d,)))
        p = np.ascontiguousarray(np.atleast_1d(np.asarray(params, dtype=np.float64)))
        if p.size == 0:
            p = np.zeros(1)
        score = np.empty(R)
        bv = C.c_double()
        bi = C.c_int64()
        self.lib.oracle_score(C.c_int(KERN[kern]), C.c_int64(d), C.c_int64(N), _p(X), _p(ll),
                              C.c_double(logsig), C.c_double(beta), _p(L), C.c_int64(L.shape[1]), _p(alpha),
                              C.c_int(ACQ[acq]), _p(p), _p(Xs), C.c_int64(R), _p(score), C.byref(bv),
                              C.byref(bi), C.c_int(nthreads))
        return score, bv.value, bi.value

    def score_grad(self, X, loglen, logsig, beta, L, alpha, acq, params, Xs, kern="SEArd"):
        X = np.ascontiguousarray(X, dtype=np.float64)
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
        N, d = X.shape
        R = Xs.shape[0]
        ll = np.ascontiguousarray(np.broadcast_to(np.asarray(loglen, dtype=np.float64), (d,)))
        p = np.ascontiguousarray(np.atleast_1d(np.asarray(params, dtype=np.float64)))
        if p.size == 0:
            p = np.zeros(1)
        score = np.empty(R)
        grad = np.empty((R, d))
        self.lib.oracle_score_grad(C.c_int(KERN[kern]), C.c_int64(d), C.c_int64(N), _p(X), _p(ll), C.c_double(logsig),
                                   C.c_double(beta), _p(L), C.c_int64(L.shape[1]), _p(alpha),
                                   C.c_int(ACQ[acq]), _p(p), _p(Xs), C.c_int64(R), _p(score), _p(grad))
        return score, grad

    def thompson(self, mu, var, z):
        S, R = z.shape
        z = np.ascontiguousarray(z, dtype=np.float64)
        bv = np.empty(S)
        bi = np.empty(S, dtype=np.int64)
        self.lib.oracle_thompson(C.c_int64(S), C.c_int64(R), _p(np.ascontiguousarray(mu)),
                                 _p(np.ascontiguousarray(var)), _p(z), _p(bv), bi.ctypes.data_as(_ip))
        return bv, bi

    def max_threads(self):
        return int(self.lib.oracle_max_threads())


# ------------------------------------------------------------------------------------------
# Independent NumPy / SciPy restatement (LAPACK-backed, like Julia's LinearAlgebra)
# ------------------------------------------------------------------------------------------
def np_normal_pdf(mu, s2):  # src/utils.jl:48
    return 1 / math.sqrt(2 * math.pi * s2) * math.exp(-mu ** 2 / (2 * s2))


def np_normal_cdf(mu, s2):  # src/utils.jl:49
    return 1 / 2 * (1 + math.erf(mu / math.sqrt(2 * s2)))


def np_acq(name, params, mu, s2):
    """src/acquisitionfunctions.jl functors, verbatim."""
    if name == "EI":  # :47-50
        tau = params[0]
        if s2 == 0:
            return mu - tau if mu > tau else 0.0
        return (mu - tau) * np_normal_cdf(mu - tau, s2) + math.sqrt(s2) * np_normal_pdf(mu - tau, s2)
    if name == "PI":  # :24-27
        tau = params[0]
        if s2 == 0:
            return float(mu > tau)
        return np_normal_cdf(mu - tau, s2)
    if name == "UCB":  # :96
        return mu + params[0] * math.sqrt(s2)
    if name == "MI":  # :141
        return mu + params[0] * (math.sqrt(s2 + params[1]) - math.sqrt(params[1]))
    if name == "MaxMean":  # :111
        return mu
    raise KeyError(name)


def brochu_beta(D, nobs, delta=0.1):  # :91-95
    nobs = 1 if nobs == 0 else nobs
    return math.sqrt(2 * math.log(nobs ** (D / 2 + 2) * math.pi ** 2 / (3 * delta)))


def np_cov(kern, X, Y, loglen, logsig):
    """cov(kernel, X, Y) for row-observation arrays X (n,d), Y (m,d)."""
    d = X.shape[1]
    il2 = np.exp(-2.0 * np.broadcast_to(np.asarray(loglen, dtype=np.float64), (d,)))
    diff = X[:, None, :] - Y[None, :, :]
    r = np.einsum("nmk,k->nm", diff * diff, il2)
    s2 = math.exp(2.0 * logsig)
    if kern == "Mat52Ard":
        s = np.sqrt(5.0) * np.sqrt(r)
        return s2 * (1.0 + s + 5.0 / 3.0 * r) * np.exp(-s)
    return s2 * np.exp(-0.5 * r)


class NumpyGP:
    """GPE-like object: fit / append / predict_f with LAPACK Cholesky and triangular solves."""

    def __init__(self, d, loglen, logsig, lognoise, beta, kern="SEArd"):
        self.d, self.kern = d, kern
        self.loglen = np.broadcast_to(np.asarray(loglen, dtype=np.float64), (d,)).copy()
        self.logsig, self.lognoise, self.beta = float(logsig), float(lognoise), float(beta)
        self.X = np.zeros((0, d))
        self.y = np.zeros(0)
        self.L = np.zeros((0, 0))
        self.alpha = np.zeros(0)

    def fit(self, X, y):
        import scipy.linalg as sl

        self.X = np.array(X, dtype=np.float64)
        self.y = np.array(y, dtype=np.float64)
        cK = np_cov(self.kern, self.X, self.X, self.loglen, self.logsig)
        cK[np.diag_indices_from(cK)] += math.exp(2 * self.lognoise) + NOISE_EPS
        self.L = sl.cholesky(cK, lower=True)
        self.alpha = sl.cho_solve((self.L, True), self.y - self.beta)
        return self

    def predict_f(self, Xs):
        import scipy.linalg as sl

        Ks = np_cov(self.kern, self.X, np.asarray(Xs, dtype=np.float64), self.loglen, self.logsig)  # N x R
        mu = self.beta + Ks.T @ self.alpha
        V = sl.solve_triangular(self.L, Ks, lower=True)
        var = np.maximum(math.exp(2 * self.logsig) - np.einsum("nr,nr->r", V, V), 0.0)
        return mu, var


def fit_with_jitter(orc, X, y, loglen, logsig, lognoise, beta, rel, max_tries=10, kern="SEArd"):
    """Oracle twin of bohip_gp_set_jitter (the role of GaussianProcesses.jl's make_posdef! behind update!,
    src/models/gp.jl:11,16 -- UPSTREAM-UNVERIFIED, which is why the device keeps it behind a switch that is off by default):
    factorise; if a pivot fails, add rel x mean(diag cK) to the diagonal and try again, x10 per further try.
    Returns (L, alpha, tries_used, jitter_added)."""
    cK = orc.build_cK(X, loglen, logsig, lognoise, kern)
    mean_diag = math.exp(2.0 * logsig) + math.exp(2.0 * lognoise)
    jit, added = rel * mean_diag, 0.0
    for t in range(max_tries + 1):
        try:
            A = cK if t == 0 else cK + added * np.eye(len(cK))
            L = orc.cholesky(A)
            return L, orc.alpha(L, y, beta), t, added
        except np.linalg.LinAlgError:
            if t == max_tries:
                raise
            added = jit
            jit *= 10.0
    raise AssertionError("unreachable")


# ------------------------------------------------------------------------------------------
# Candidate generator: latin_hypercube_sampling, src/utils.jl:101-120 (NumPy RNG instead of
# Julia's global RNG -- the draws are not reproducible across languages, so candidates are an
# *input* of the hot path; only the stratification property is part of the contract).
# ------------------------------------------------------------------------------------------
def latin_hypercube(lb, ub, n, rng):
    lb = np.asarray(lb, dtype=np.float64)
    ub = np.asarray(ub, dtype=np.float64)
    if lb.shape != ub.shape:
        raise ValueError("mins and maxs should have the same length")
    if not np.all(lb <= ub):
        raise ValueError("mins[i] should not exceed maxs[i]")
    d = lb.size
    out = np.zeros((n, d))
    for i in range(d):
        step = (ub[i] - lb[i]) / n
        col = lb[i] + step * (np.arange(n) + rng.random(n))
        rng.shuffle(col)
        out[:, i] = col
    return out


def argmax_first(scores):
    """acquire_max's reduction (src/acquisition.jl:55,62-65): strict '>' from -Inf, NaN never wins."""
    best, idx = -math.inf, -1
    for i, f in enumerate(scores):
        if f > best:
            best, idx = f, i
    return best, idx


# ------------------------------------------------------------------------------------------
# mpmath evaluators (error bounding only)
# ------------------------------------------------------------------------------------------
def mp_predict(X, y, loglen, logsig, lognoise, beta, Xs, dps=60):
    import mpmath as mp

    mp.mp.dps = dps
    N, d = X.shape
    il2 = [mp.e ** (-2 * mp.mpf(float(l))) for l in np.broadcast_to(loglen, (d,))]
    s2 = mp.e ** (2 * mp.mpf(logsig))
    noise = mp.e ** (2 * mp.mpf(lognoise)) + mp.mpf(NOISE_EPS)

    def k(a, b):
        r = sum(il2[q] * (mp.mpf(float(a[q])) - mp.mpf(float(b[q]))) ** 2 for q in range(d))
        return s2 * mp.e ** (-r / 2)

    K = mp.matrix(N, N)
    for i in range(N):
        for j in range(N):
            K[i, j] = k(X[i], X[j]) + (noise if i == j else 0)
    Lm = mp.cholesky(K)
    rhs = mp.matrix([mp.mpf(float(v)) - mp.mpf(beta) for v in y])
    alpha = mp.cholesky_solve(K, rhs)
    mus, vars_ = [], []
    for xs in Xs:
        ks = mp.matrix([k(X[i], xs) for i in range(N)])
        mu = mp.mpf(beta) + sum(ks[i] * alpha[i] for i in range(N))
        v = mp.lu_solve(Lm, ks)
        var = s2 - sum(v[i] ** 2 for i in range(N))
        mus.append(mu)
        vars_.append(var if var > 0 else mp.mpf(0))
    return mus, vars_


def mp_acq(name, params, mu, s2, dps=60):
    import mpmath as mp

    mp.mp.dps = dps
    mu, s2 = mp.mpf(mu), mp.mpf(s2)
    p = [mp.mpf(float(v)) for v in params]
    cdf = lambda m, v: (1 + mp.erf(m / mp.sqrt(2 * v))) / 2
    pdf = lambda m, v: 1 / mp.sqrt(2 * mp.pi * v) * mp.e ** (-m ** 2 / (2 * v))
    if name == "EI":
        if s2 == 0:
            return mu - p[0] if mu > p[0] else mp.mpf(0)
        return (mu - p[0]) * cdf(mu - p[0], s2) + mp.sqrt(s2) * pdf(mu - p[0], s2)
    if name == "PI":
        if s2 == 0:
            return mp.mpf(1 if mu > p[0] else 0)
        return cdf(mu - p[0], s2)
    if name == "UCB":
        return mu + p[0] * mp.sqrt(s2)
    if name == "MI":
        return mu + p[0] * (mp.sqrt(s2 + p[1]) - mp.sqrt(p[1]))
    return mu
