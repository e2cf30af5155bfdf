"""Model adapter: the six generic functions through which the reference's BO loop touches the GP
(reference src/models/gp.jl:2-18), backed by the device-resident factor in libbohip.

Arrays keep the reference's Julia shapes: a batch of points is ``d x R`` (one point per COLUMN).
``np.asfortranarray`` of such an array has exactly the memory layout the C ABI wants
(every point = d contiguous doubles).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import Best, check

_dp = C.POINTER(C.c_double)


def _ptr(a):
    return a.ctypes.data_as(_dp)


def _cols(x, d):
    """d x R (or length-d vector) -> Fortran-contiguous float64 d x R."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    if x.shape[0] != d:
        raise ValueError(f"expected {d} rows (one point per column), got shape {x.shape}")
    return np.asfortranarray(x)


# ---- GaussianProcesses.jl constructor vocabulary used by the reference --------------------------
class MeanZero:
    beta = 0.0


class MeanConst:
    def __init__(self, beta):
        self.beta = float(beta)


class _Kernel:
    kern = "SEArd"

    def __init__(self, ll, lsigma):
        self.ll = np.atleast_1d(np.asarray(ll, dtype=np.float64)).copy()
        self.lsigma = float(lsigma)


class SEArd(_Kernel):
    """SEArd(ll::Vector, lσ): k = exp(2lσ) exp(-½ Σ (x-y)²/exp(2 ll_k))  (README.md:24)."""
    kern = "SEArd"


class SEIso(_Kernel):
    """SEIso(ll, lσ)  (test/acquisition.jl:2)."""
    kern = "SEIso"


class Mat52Ard(_Kernel):
    """Mat52Ard(ll::Vector, lσ)  (default model, src/BayesianOptimization.jl:259-262)."""
    kern = "Mat52Ard"


class ElasticGPE:
    """Drop-in for ``ElasticGPE(d; mean, kernel, logNoise, capacity)`` (README.md:22-27).

    Owns a ``bohip_gp`` handle: x, y, the Cholesky factor, its inverse and alpha live in HBM and
    survive across ``boptimize_`` calls; ``append_`` extends the factor (reference ``append!``).
    ``model.x`` (d x n) and ``model.y`` are host mirrors, as the reference reads those fields
    directly (src/BayesianOptimization.jl:117-119, src/acquisitionfunctions.jl:136).
    """

    def __init__(self, d, mean=None, kernel=None, logNoise=-2.0, capacity=1024, device=0):
        self.dim = int(d)
        self.mean = mean if mean is not None else MeanZero()
        self.kernel = kernel if kernel is not None else SEArd(np.zeros(d), 0.0)
        self.logNoise = float(logNoise)
        self._lib = _lib.load()
        h = C.c_void_p()
        check(self._lib.bohip_gp_create(self.dim, int(capacity), _lib.KERN[self.kernel.kern], int(device), C.byref(h)))
        self._h = h
        _lib.register(self)
        self._x = np.zeros((self.dim, 0), order="F")
        self._y = np.zeros(0)
        self._push_hyper()

    def close(self):
        """Release the device model now (idempotent; also run for every live model at interpreter exit)."""
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.bohip_gp_destroy(h)

    @classmethod
    def from_data(cls, x, y, mean=None, kernel=None, logNoise=-2.0, **kw):
        """GPE(x, y, mean, kernel[, logNoise]) (test/acquisitionfunctions.jl:4, test/acquisition.jl:2)."""
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x.reshape(1, -1)
        m = cls(x.shape[0], mean=mean, kernel=kernel, logNoise=logNoise, capacity=max(x.shape[1], 1), **kw)
        m.append_(x, y)
        return m

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- hyper-parameters (GP.set_params!) ---------------------------------------------------------
    def _push_hyper(self):
        ll = self.kernel.ll
        if self.kernel.kern != "SEIso" and ll.size != self.dim:
            raise ValueError("kernel length-scale vector must have d entries")
        ll = np.ascontiguousarray(np.broadcast_to(ll, (self.dim,)) if ll.size == 1 else ll)
        check(self._lib.bohip_gp_set_hyper(self._h, _ptr(ll), self.kernel.lsigma, self.logNoise, self.mean.beta))

    def set_params_(self, ll=None, lsigma=None, logNoise=None, beta=None):
        if ll is not None:
            self.kernel.ll = np.atleast_1d(np.asarray(ll, dtype=np.float64)).copy()
        if lsigma is not None:
            self.kernel.lsigma = float(lsigma)
        if logNoise is not None:
            self.logNoise = float(logNoise)
        if beta is not None:
            self.mean = MeanConst(beta)
        self._push_hyper()

    # -- fields the reference reads directly -------------------------------------------------------
    @property
    def x(self):
        return self._x

    @property
    def y(self):
        return self._y

    @property
    def nobs(self):
        return self._y.size

    # -- append! / fit! ----------------------------------------------------------------------------
    def append_(self, x, y):
        x = _cols(x, self.dim)
        y = np.ascontiguousarray(np.atleast_1d(np.asarray(y, dtype=np.float64)))
        if x.shape[1] != y.size:
            raise ValueError("x and y disagree on the number of observations")
        rc = self._lib.bohip_gp_append(self._h, _ptr(x), _ptr(y), y.size)
        if rc in (_lib.OK, _lib.E_NOTPD):  # observations are stored even when the factorisation fails
            self._x = np.asfortranarray(np.concatenate([self._x, x], axis=1))
            self._y = np.concatenate([self._y, y])
        check(rc)
        return self

    def fit_(self):
        check(self._lib.bohip_gp_refit(self._h))
        return self

    def mll(self):
        out = C.c_double()
        check(self._lib.bohip_gp_mll(self._h, C.byref(out)))
        return out.value

    def mll_grad(self):
        """(mll, dlogNoise, dmean, dkern) -- gp.target / gp.dtarget after update_target_and_dtarget!
        (reference src/models/gp.jl:61-63); dkern = [dll..., dlsigma] in the kernel's parameter order."""
        nk = (1 if isinstance(self.kernel, SEIso) else self.dim) + 1
        m, dn, dm = C.c_double(), C.c_double(), C.c_double()
        dk = np.empty(nk)
        check(self._lib.bohip_gp_mll_grad(self._h, C.byref(m), C.byref(dn), C.byref(dm), _ptr(dk)))
        return m.value, dn.value, dm.value, dk

    # -- predict_f / scoring ------------------------------------------------------------------------
    def predict_f(self, xs):
        xs = _cols(xs, self.dim)
        R = xs.shape[1]
        mu = np.empty(R)
        var = np.empty(R)
        check(self._lib.bohip_gp_predict(self._h, _ptr(xs), R, _ptr(mu), _ptr(var)))
        return mu, var

    def predict_cov(self, xs):
        """predict_f(gp, X; full_cov = true): (mu, R x R posterior covariance of the latent f)."""
        xs = _cols(xs, self.dim)
        R = xs.shape[1]
        mu = np.empty(R)
        cov = np.empty((R, R))
        check(self._lib.bohip_gp_predict_cov(self._h, _ptr(xs), R, _ptr(mu), _ptr(cov)))
        return mu, cov

    def score(self, acq, params, xs, want_scores=True):
        """Fused predict + acquisition + arg-max.  Returns (scores or None, best_val, best_idx)."""
        xs = _cols(xs, self.dim)
        R = xs.shape[1]
        p = np.ascontiguousarray(np.atleast_1d(np.asarray(params, dtype=np.float64)))
        if p.size < 2:
            p = np.concatenate([p, np.zeros(2 - p.size)])
        sc = np.empty(R) if want_scores else None
        best = Best()
        check(self._lib.bohip_gp_score(self._h, _lib.ACQ[acq], _ptr(p), _ptr(xs), R,
                                       _ptr(sc) if want_scores else None, C.byref(best)))
        return sc, best.val, best.idx

    def score_grad(self, acq, params, xs):
        xs = _cols(xs, self.dim)
        R = xs.shape[1]
        p = np.ascontiguousarray(np.atleast_1d(np.asarray(params, dtype=np.float64)))
        if p.size < 2:
            p = np.concatenate([p, np.zeros(2 - p.size)])
        sc = np.empty(R)
        grad = np.empty((self.dim, R), order="F")
        check(self._lib.bohip_gp_score_grad(self._h, _lib.ACQ[acq], _ptr(p), _ptr(xs), R, _ptr(sc), _ptr(grad)))
        return sc, grad

    def set_maxtime(self, seconds):
        """NLopt's maxtime for the device ascent (0 = unlimited)."""
        check(self._lib.bohip_gp_set_maxtime(self._h, float(seconds)))

    def set_ascent_stop(self, ftol_abs=0.0, xtol_rel=0.0, stopval=float("inf")):
        """NLopt's ftol_abs / xtol_rel / stopval for the device ascent (0 / 0 / +Inf = off), reference src/acquisition.jl:24-27."""
        check(self._lib.bohip_gp_set_ascent_stop(self._h, float(ftol_abs), float(xtol_rel), float(stopval)))

    def set_jitter(self, rel, max_tries=10):
        """Jitter escalation on a failed factorisation (the role of GaussianProcesses.jl's make_posdef! behind update!,
        src/models/gp.jl:11,16 -- UPSTREAM-UNVERIFIED, off by default): a refit that fails is repeated with
        rel x mean(diag cK) more on the diagonal, x10 per further try; info(INFO_JITTER_STEPS) tells how many it took."""
        check(self._lib.bohip_gp_set_jitter(self._h, float(rel), int(max_tries)))

    def ascend(self, acq, params, lowerbounds, upperbounds, starts, maxeval=2000, ftol_rel=1e-10, xtol_abs=1e-10):
        """Local search of acquire_max on the device (src/acquisition.jl:48-68 with :LD_LBFGS and bounds): every start
        column is refined by a projected L-BFGS ascent, all columns in the same device passes.  Returns
        (f[R], X[d, R], best_f, best_index, best_x[d], evaluations)."""
        starts = _cols(starts, self.dim)
        R = starts.shape[1]
        p = np.ascontiguousarray(np.atleast_1d(np.asarray(params, dtype=np.float64)))
        if p.size < 2:
            p = np.concatenate([p, np.zeros(2 - p.size)])
        lb = np.ascontiguousarray(lowerbounds, dtype=np.float64)
        ub = np.ascontiguousarray(upperbounds, dtype=np.float64)
        if lb.size != self.dim or ub.size != self.dim:
            raise ValueError("bounds must have one entry per input dimension")
        f = np.empty(R)
        X = np.empty((self.dim, R), order="F")
        best = Best()
        bx = np.empty(self.dim)
        ev = C.c_int64(0)
        check(self._lib.bohip_gp_acquire_max(self._h, _lib.ACQ[acq], _ptr(p), _ptr(lb), _ptr(ub), _ptr(starts), R, int(maxeval),
                                             float(ftol_rel), float(xtol_abs), _ptr(X), _ptr(f), C.byref(best), _ptr(bx),
                                             C.byref(ev)))
        return f, X, best.val, best.idx, bx, ev.value

    # -- one process per device: communicator on the handle, exchange inside libbohip (in-library RCCL) ---------
    def comm_init(self, unique_id, rank, nranks):
        buf = C.create_string_buffer(bytes(unique_id), _lib.UNIQUE_ID_BYTES)
        check(self._lib.bohip_gp_comm_init(self._h, buf, _lib.UNIQUE_ID_BYTES, int(rank), int(nranks)))

    def comm_destroy(self):
        check(self._lib.bohip_gp_comm_destroy(self._h))

    def score_sharded_dev(self, acq, params, d_xs_ptr, R_local, col_offset, R_total, d_best_ptr, d_score_ptr=None):
        """Enqueue: score this rank's shard, all-gather the records over RCCL, reduce on the device; the global winner
        lands at d_best_ptr (device or pinned-host address), identical on every rank.  No host synchronisation."""
        p = np.ascontiguousarray(np.atleast_1d(np.asarray(params, dtype=np.float64)))
        if p.size < 2:
            p = np.concatenate([p, np.zeros(2 - p.size)])
        check(self._lib.bohip_gp_score_sharded_dev(self._h, _lib.ACQ[acq], _ptr(p), C.c_void_p(d_xs_ptr), int(R_local),
                                                   int(col_offset), int(R_total),
                                                   C.c_void_p(d_score_ptr) if d_score_ptr else None, C.c_void_p(d_best_ptr)))

    def thompson_sharded(self, xs, S, seed, col_offset, R_total):
        xs = _cols(xs, self.dim)
        out = (Best * S)()
        check(self._lib.bohip_gp_thompson_sharded(self._h, _ptr(xs), xs.shape[1], S, seed, int(col_offset), int(R_total), out))
        return np.array([b.val for b in out]), np.array([b.idx for b in out], dtype=np.int64)

    def synchronize(self):
        check(self._lib.bohip_gp_synchronize(self._h))

    def direct_max(self, acq, params, lowerbounds, upperbounds, maxeval=2000, stopval=float("inf"), maxtime=0.0, seed=0):
        """:GN_DIRECT_L on the device model in one call (bohip_gp_direct_max; reference src/acquisition.jl:7-9, :20-38): the
        dividing-rectangles bookkeeping runs in the library, every iteration's points are one scoring call.  acq = "ThompsonDraw":
        x -> myrand(model, x), one posterior draw per point from the library's counter-based generator keyed by `seed`.
        Returns (best value, best point, evaluations, device calls)."""
        lb = np.ascontiguousarray(lowerbounds, dtype=np.float64)
        ub = np.ascontiguousarray(upperbounds, dtype=np.float64)
        if lb.size != self.dim or ub.size != self.dim:
            raise ValueError("bounds must have one entry per input dimension")
        p = np.zeros(2)
        if params is not None:
            q = np.atleast_1d(np.asarray(params, dtype=np.float64))
            p[:q.size] = q[:2]
        bf = C.c_double(); bx = np.empty(self.dim); ev = C.c_int64(); dc = C.c_int64()
        check(self._lib.bohip_gp_direct_max(self._h, _lib.ACQ[acq], _ptr(p), _ptr(lb), _ptr(ub), int(maxeval), float(stopval),
                                            float(maxtime or 0.0), int(seed), C.byref(bf), _ptr(bx), C.byref(ev), C.byref(dc)))
        return float(bf.value), bx, int(ev.value), int(dc.value)

    def thompson(self, xs, S, seed=0, j0=0):
        xs = _cols(xs, self.dim)
        out = (Best * S)()
        check(self._lib.bohip_gp_thompson(self._h, _ptr(xs), xs.shape[1], S, seed, j0, out))
        return np.array([b.val for b in out]), np.array([b.idx for b in out], dtype=np.int64)

    # -- introspection ------------------------------------------------------------------------------
    def factor(self):
        n = self.nobs
        L = np.zeros((n, n))
        check(self._lib.bohip_gp_get_factor(self._h, _ptr(L)))
        return L

    def alpha(self):
        a = np.zeros(self.nobs)
        check(self._lib.bohip_gp_get_alpha(self._h, _ptr(a)))
        return a

    def info(self, what):
        v = C.c_int64()
        check(self._lib.bohip_gp_info(self._h, what, C.byref(v)))
        return v.value

    def set_batch_hint(self, total_candidates):
        """Score shards of a larger candidate set with the summation schedule of the whole set (bit-identical to the
        unsharded call); 0 clears the hint."""
        check(self._lib.bohip_gp_set_batch_hint(self._h, int(total_candidates)))

    def enable_timing(self, on=True):     # True/1: every stage; 2: only the dominant kernel; 3: as 2, read after the loop
        check(self._lib.bohip_gp_enable_timing(self._h, int(on)))

    def timing(self, cap=64):
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        n = self._lib.bohip_gp_get_timing(self._h, names, ms, cap)
        return [(names[i].decode(), ms[i]) for i in range(min(n, cap))]

    def __repr__(self):
        return (f"ElasticGPE(dim={self.dim}, nobs={self.nobs}, kernel={self.kernel.kern}"
                f"(ll={self.kernel.ll.tolist()}, lσ={self.kernel.lsigma}), mean β={self.mean.beta}, "
                f"logNoise={self.logNoise}) [device-resident, libbohip]")


# ---- the generic functions of reference src/models/gp.jl ----------------------------------------
def mean_var(model, x):
    """gp.jl:2-5 (vector -> scalars) and :8 (d x R matrix -> vectors)."""
    x = np.asarray(x, dtype=np.float64)
    mu, var = model.predict_f(x)
    if x.ndim == 1:
        return float(mu[0]), float(var[0])
    return mu, var


def myrand(model, x, rng=None):
    """gp.jl:6-7.  Vector: one draw from N(mu, sigma^2).  Matrix: ONE JOINT draw from N(mu, Sigma_post) over the columns
    (rand(gp, X) = mu + chol(Sigma)·z with jitter added until the factorisation succeeds -- GaussianProcesses.jl
    make_posdef!, UPSTREAM-UNVERIFIED; only the length is pinned by test/acquisitionfunctions.jl:8)."""
    rng = rng if rng is not None else np.random.default_rng()
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        mu, var = model.predict_f(x)
        return float(mu[0] + math.sqrt(var[0]) * rng.standard_normal())
    if model.nobs == 0:
        raise RuntimeError("myrand on an empty model")
    mu, cov = model.predict_cov(x)
    z = rng.standard_normal(mu.shape)
    jitter, scale = 0.0, max(float(np.max(np.diag(cov))), np.finfo(float).tiny)
    for _ in range(40):
        try:
            Lc = np.linalg.cholesky(cov + jitter * np.eye(len(mu)))
            return mu + Lc @ z
        except np.linalg.LinAlgError:
            jitter = max(10.0 * jitter, 1e-12 * scale)
    raise np.linalg.LinAlgError("posterior covariance could not be made positive definite")


def dims(model):
    """gp.jl:9 -> (D, nobs)."""
    return model.dim, model.nobs


def maxy(model):
    """gp.jl:10."""
    return -math.inf if model.nobs == 0 else float(np.max(model.y))


def update_(model, x, y):
    """gp.jl:11  update!(model::GPE{<:ElasticArray}, x, y) = append!(model, x, y)."""
    return model.append_(x, y)
