"""Acquisition types and the multi-restart search, mirroring reference src/acquisitionfunctions.jl and
src/acquisition.jl.  The functors keep the reference's formulas verbatim (host versions, used for single
points and documentation); batches are scored on the device through ``model.score``.

Where the reference runs R sequential NLopt ascents (src/acquisition.jl:58-66), ``acquire_max`` scores /
ascends ALL R Latin-hypercube starts together on the GPU:
    method :LD_*  -> batched projected L-BFGS driven by the device's analytic gradient (score_grad)
    method :GN_* / :LN_* -> derivative-free: ``maxeval`` candidates scored in one batch
"""
from __future__ import annotations

import math
import warnings
from types import SimpleNamespace

import numpy as np

from .model import ElasticGPE, dims, maxy, mean_var, myrand
from .utils import latin_hypercube_sampling, normal_cdf, normal_pdf


class AbstractAcquisition:
    acq_id = None

    def params(self):
        return []


def setparams_(a, model):                                     # setparams!(a, model) = nothing :3
    f = getattr(a, "_setparams", None)
    return f(model) if f else None


class ProbabilityOfImprovement(AbstractAcquisition):          # :21-28
    acq_id = "PI"

    def __init__(self, tau=-math.inf):
        self.tau = float(tau)

    def __call__(self, mu, s2):
        if s2 == 0:
            return float(mu > self.tau)
        return normal_cdf(mu - self.tau, s2)

    def _setparams(self, model):                              # :44-46
        self.tau = max(maxy(model), self.tau)
        return self.tau

    def params(self):
        return [self.tau]


class ExpectedImprovement(AbstractAcquisition):               # :40-50
    acq_id = "EI"

    def __init__(self, tau=-math.inf):
        self.tau = float(tau)

    def __call__(self, mu, s2):
        if s2 == 0:
            return mu - self.tau if mu > self.tau else 0.0
        return (mu - self.tau) * normal_cdf(mu - self.tau, s2) + math.sqrt(s2) * normal_pdf(mu - self.tau, s2)

    _setparams = ProbabilityOfImprovement._setparams
    params = ProbabilityOfImprovement.params


class BrochuBetaScaling:                                      # :66-68
    def __init__(self, delta=0.1):
        self.delta = float(delta)


class NoBetaScaling:                                          # :72
    pass


class UpperConfidenceBound(AbstractAcquisition):              # :81-96
    acq_id = "UCB"

    def __init__(self, scaling=None, beta_t=1.0):
        self.scaling = scaling if scaling is not None else BrochuBetaScaling(0.1)
        self.beta_t = float(beta_t)

    def __call__(self, mu, s2):
        return mu + self.beta_t * math.sqrt(s2)

    def _setparams(self, model):                              # :91-95 (BrochuBetaScaling only)
        if isinstance(self.scaling, BrochuBetaScaling):
            D, nobs = dims(model)
            nobs = 1 if nobs == 0 else nobs
            self.beta_t = math.sqrt(2 * math.log(nobs ** (D / 2 + 2) * math.pi ** 2 / (3 * self.scaling.delta)))
        return self.beta_t

    def params(self):
        return [self.beta_t]


class ThompsonSamplingSimple(AbstractAcquisition):            # :107-108
    acq_id = "Thompson"


class MaxMean(AbstractAcquisition):                           # :110-111
    acq_id = "MaxMean"

    def __call__(self, mu, s2):
        return mu


class MutualInformation(AbstractAcquisition):                 # :126-141
    acq_id = "MI"

    def __init__(self, alpha=1.0, gamma_hat=0.0):
        self.sqrt_alpha = math.sqrt(alpha)
        self.gamma_hat = float(gamma_hat)

    def __call__(self, mu, s2):
        return mu + self.sqrt_alpha * (math.sqrt(s2 + self.gamma_hat) - math.sqrt(self.gamma_hat))

    def _setparams(self, model):                              # :131-140
        D, nobs = dims(model)
        if nobs == 0:
            self.gamma_hat = 0.0
        else:
            last_x = model.x[:, -1]
            _, s2 = mean_var(model, last_x)
            self.gamma_hat += s2
        return self.gamma_hat

    def params(self):
        return [self.sqrt_alpha, self.gamma_hat]


def acquisitionfunction(a, model, rng=None):
    """:4-9, :108, :111.  x (vector or d x R matrix) -> score(s); batches run fused on the device."""
    if isinstance(a, ThompsonSamplingSimple):
        return lambda x: myrand(model, x, rng)
    if isinstance(a, MaxMean):
        return lambda x: mean_var(model, x)[0]

    def f(x):
        x = np.asarray(x, dtype=np.float64)
        sc, _, _ = model.score(a.acq_id, a.params(), x)
        return float(sc[0]) if x.ndim == 1 else sc

    return f


# ---- src/acquisition.jl ----------------------------------------------------------------------------------
def defaultoptions(model_type, acq_type):                    # :4-9
    if isinstance(acq_type, type) and issubclass(acq_type, ThompsonSamplingSimple):
        return dict(method="GN_DIRECT_L", restarts=1, maxeval=2000)
    return dict(method="LD_LBFGS", restarts=10, maxeval=2000)


ASC_MAX_BT = 30   # kernels_ascent.hip: trial points per line search


def _batched_lbfgs_ascent(fg, X0, lb, ub, maxeval, ftol_rel=1e-10, xtol_abs=1e-10, history=8, ftol_abs=0.0, xtol_rel=0.0,
                          stopval=math.inf):
    """Lock-step projected L-BFGS ascent of R independent d-dimensional problems (role of NLopt :LD_LBFGS
    with lower/upper bounds, src/acquisition.jl:23-35).  fg(X) -> (f[R], G[d,R]) is ONE device call.
    Returns (f, X) at the best point seen per column."""
    d, R = X0.shape
    lbc, ubc = lb.reshape(-1, 1), ub.reshape(-1, 1)
    X = np.clip(X0, lbc, ubc)
    f, G = fg(X)
    evals = 1
    S, Y = [], []
    active = np.isfinite(f)
    best_f, best_X = f.copy(), X.copy()
    while evals < maxeval and active.any():
        # two-loop recursion, vectorised over columns, in the FREE SUBSPACE of every column: a coordinate on a bound with the gradient
        # pushing outward takes no part (kernels_ascent.hip asc_direction_one: 22-35 passes instead of 228-309 on the headline model,
        # where UCB peaks in the corners of the box; with no bound active the arithmetic is unchanged)
        free = ~(((X <= lbc) & (G < 0)) | ((X >= ubc) & (G > 0)))
        q = np.where(free, G, 0.0)
        al = []
        for s_, y_ in zip(reversed(S), reversed(Y)):
            s_, y_ = np.where(free, s_, 0.0), np.where(free, y_, 0.0)
            sy = np.einsum("dr,dr->r", y_, s_)
            rho = np.where(sy > 1e-14, 1.0 / np.where(sy > 1e-14, sy, 1.0), 0.0)     # a pair without curvature in the free subspace is skipped
            a_ = rho * np.einsum("dr,dr->r", s_, q)
            q -= a_ * y_
            al.append((a_, rho))
        if S:
            s_, y_ = np.where(free, S[-1], 0.0), np.where(free, Y[-1], 0.0)
            sy = np.einsum("dr,dr->r", s_, y_)
            yy = np.maximum(np.einsum("dr,dr->r", y_, y_), 1e-300)
            q *= np.where(sy > 1e-14, sy / yy, 1.0)
        for (a_, rho), s_, y_ in zip(reversed(al), S, Y):
            s_, y_ = np.where(free, s_, 0.0), np.where(free, y_, 0.0)
            b_ = rho * np.einsum("dr,dr->r", y_, q)
            q += (a_ - b_) * s_
        D = q                                                     # ascent direction (maximisation: H * grad)
        # bounds: do not push active constraints outward; fall back to the gradient if not an ascent direction
        blocked = ((X <= lbc) & (D < 0)) | ((X >= ubc) & (D > 0))
        D = np.where(blocked, 0.0, D)
        Gp = np.where(((X <= lbc) & (G < 0)) | ((X >= ubc) & (G > 0)), 0.0, G)
        slope = np.einsum("dr,dr->r", Gp, D)
        bad = ~(slope > 0)
        D[:, bad] = Gp[:, bad]
        slope = np.einsum("dr,dr->r", Gp, D)
        if S:
            step = np.ones(R)
        else:                                                     # the FIRST step (kernels_ascent.hip asc_direction_one): L-BFGS-B's unit step to P(x + g),
            step = np.maximum(1.0, 0.1 * np.min(ub - lb + 1e-300) / np.maximum(np.sqrt(np.einsum("dr,dr->r", D, D)), 1e-12))   # but never shorter than a tenth of the box
        step = np.where(active & (slope > 0), step, 0.0)
        accepted = ~active | ~(slope > 0)
        Xn, fn, Gn = X.copy(), f.copy(), G.copy()
        for _ in range(ASC_MAX_BT):                               # backtracking Armijo, all columns per device call
            Xt = np.clip(X + step * D, lbc, ubc)
            ft, Gt = fg(Xt)
            evals += 1
            ok = (~accepted) & np.isfinite(ft) & (ft >= f + 1e-4 * np.einsum("dr,dr->r", Gp, Xt - X))
            Xn[:, ok], fn[ok], Gn[:, ok] = Xt[:, ok], ft[ok], Gt[:, ok]
            accepted |= ok
            if accepted.all() or evals >= maxeval:
                break
            step = np.where(accepted, step, step * 0.5)
        s_ = Xn - X
        y_ = -(Gn - G)                                            # curvature pair for maximisation
        df = fn - f
        moved = np.sqrt(np.einsum("dr,dr->r", s_, s_))
        active = active & (df > ftol_rel * np.maximum(np.abs(fn), 1e-300)) & (moved > xtol_abs)
        # NLopt's other stop tests (kernels_ascent.hip asc_goes_on): ftol_abs, xtol_rel per coordinate, stopval
        active = active & (df > ftol_abs) & ~(fn >= stopval)
        if xtol_rel > 0.0:
            active = active & (np.abs(s_) > xtol_rel * np.abs(Xn)).any(axis=0)
        good = np.einsum("dr,dr->r", s_, y_) > 1e-14
        S.append(np.where(good, s_, 0.0)); Y.append(np.where(good, y_, 0.0))
        if len(S) > history:
            S.pop(0); Y.pop(0)
        X, f, G = Xn, fn, Gn
        better = f > best_f
        best_f[better], best_X[:, better] = f[better], X[:, better]
    return best_f, best_X


def _batched_direct_l(f_batch, lb, ub, maxeval, stopval=math.inf, maxtime=0.0):
    """DIviding RECTangles, locally biased (Gablonsky & Kelley 2001) -- the role of NLopt's :GN_DIRECT_L, which the reference uses
    by default for ThompsonSamplingSimple (src/acquisition.jl:7-9: restarts = 1, maxeval = 2000) -- for MAXIMISATION, with ALL the
    new points of one iteration evaluated in ONE device call: f_batch(X[d, n]) -> f[n].
    The rules are those of NLopt's cdirect.c for this algorithm (its `which_alg = 13`): a rectangle's size is its longest side, the
    potentially optimal set is the upper convex hull of (size, best value of that size) from the incumbent's size up with ONE
    rectangle per size (Jones' epsilon = 0), a cube is trisected along every side -- best sampled value first, so the best points
    end up in the largest boxes -- and any other rectangle along its first longest side only.  What is NOT reproduced is NLopt's
    evaluation ORDER inside an iteration (irrelevant for a deterministic objective, a different random stream for a sampled one).
    `maxtime` > 0: NLopt's wall-clock budget in seconds (the reference's test passes it; checked once per iteration).
    Returns (best value, best point, evaluations)."""
    import time

    deadline = time.monotonic() + maxtime if maxtime and maxtime > 0 else math.inf
    lb = np.asarray(lb, float); ub = np.asarray(ub, float)
    d = lb.size
    span = ub - lb
    to_x = lambda U: lb[:, None] + span[:, None] * U              # noqa: E731  unit cube -> box, columns are points
    C = np.full((d, 1), 0.5)                                      # centres (unit cube)
    Lv = np.zeros((d, 1), dtype=np.int64)                         # level per side: side length 3^-level
    F = np.asarray(f_batch(to_x(C)), float).reshape(-1)
    F = np.where(np.isnan(F), -math.inf, F)
    evals = 1
    while evals < maxeval and not (F.max() >= stopval) and time.monotonic() < deadline:
        size = Lv.min(axis=0)                                     # key of the longest side (smaller = larger rectangle)
        keys = np.unique(size)
        best_of = {}
        for k in keys:                                            # one rectangle per size: the best, first on ties
            idx = np.flatnonzero(size == k)
            best_of[int(k)] = int(idx[np.argmax(F[idx])])
        jmax = int(np.argmax(F))
        # upper hull over (diameter, f) from the incumbent's size towards the larger rectangles
        cand = sorted((k for k in best_of if k <= size[jmax]), reverse=True)          # increasing diameter
        hull = []
        for k in cand:
            pt = (3.0 ** (-k), F[best_of[k]], best_of[k])
            while len(hull) >= 2:
                (x1, y1, _), (x2, y2, _) = hull[-2], hull[-1]
                if (y2 - y1) * (pt[0] - x1) <= (pt[1] - y1) * (x2 - x1):             # the middle point is not above the chord
                    hull.pop()
                else:
                    break
            hull.append(pt)
        chosen = [j for _, _, j in hull]                          # (the first one is the incumbent's own size class)
        # new points of this iteration (capped by the evaluation budget)
        plan, cols = [], []
        for j in chosen:
            lv = Lv[:, j]
            kmin = lv.min()
            longest = np.flatnonzero(lv == kmin)
            dims = longest if longest.size == d else longest[:1]   # a cube: every side; otherwise the first longest side
            if evals + len(cols) + 2 * len(dims) > maxeval:
                dims = dims[: max(0, (maxeval - evals - len(cols)) // 2)]
            if len(dims) == 0:
                continue
            delta = 3.0 ** (-(kmin + 1))
            start = len(cols)
            for i in dims:
                for sgn in (+1.0, -1.0):
                    c = C[:, j].copy(); c[i] += sgn * delta
                    cols.append(c)
            plan.append((j, dims, start))
        if not cols:
            break
        Unew = np.array(cols).T
        Fnew = np.asarray(f_batch(to_x(Unew)), float).reshape(-1)
        Fnew = np.where(np.isnan(Fnew), -math.inf, Fnew)
        evals += Fnew.size
        newL = np.empty((d, Fnew.size), dtype=np.int64)
        for j, dims, start in plan:
            w = np.array([max(Fnew[start + 2 * t], Fnew[start + 2 * t + 1]) for t in range(len(dims))])
            order = np.argsort(-w, kind="stable")                 # best sampled value first
            lv = Lv[:, j].copy()
            for t in order:
                i = dims[t]
                lv[i] += 1                                        # the parent shrinks along i; the two children inherit the levels so far
                newL[:, start + 2 * t] = lv
                newL[:, start + 2 * t + 1] = lv
            Lv[:, j] = lv
        C = np.concatenate([C, Unew], axis=1)
        Lv = np.concatenate([Lv, newL], axis=1)
        F = np.concatenate([F, Fnew])
    jb = int(np.argmax(F))
    return float(F[jb]), to_x(C[:, jb:jb + 1])[:, 0], evals


def direct_l_search(f_batch, lb, ub, maxeval, stopval=math.inf, maxtime=0.0):
    """The same search with the bookkeeping in libbohip (csrc/direct_l.h, bohip_direct_ask / _tell): the caller's f_batch scores each
    iteration's points.  Bit-identical to _batched_direct_l (tests/test_direct_l.py), which stays as its NumPy twin; the NumPy
    bookkeeping was 21 of the 27.7 ms of a default Thompson acquire_max at N = 3000.  Returns (best value, best point, evaluations)."""
    import ctypes as C
    from . import _lib

    lib = _lib.load()
    lb = np.ascontiguousarray(lb, dtype=np.float64); ub = np.ascontiguousarray(ub, dtype=np.float64)
    d = lb.size
    h = C.c_void_p()
    dp = C.POINTER(C.c_double)
    _lib.check(lib.bohip_direct_create(d, lb.ctypes.data_as(dp), ub.ctypes.data_as(dp), int(max(1, maxeval)), float(stopval),
                                       float(maxtime or 0.0), C.byref(h)))
    try:
        X = np.empty((max(2 * d, 64), d))                         # rows = points (d x cap column-major for the library)
        n = C.c_int64()
        while True:
            _lib.check(lib.bohip_direct_ask(h, None, 0, C.byref(n)))         # size of this iteration's batch
            if n.value == 0:
                break
            if n.value > X.shape[0]:
                X = np.empty((2 * n.value, d))
            _lib.check(lib.bohip_direct_ask(h, X.ctypes.data_as(dp), X.shape[0], C.byref(n)))
            F = np.ascontiguousarray(np.asarray(f_batch(X[:n.value].T), dtype=np.float64).reshape(-1))
            _lib.check(lib.bohip_direct_tell(h, F.ctypes.data_as(dp), n.value))
        bf = C.c_double(); bx = np.empty(d); ev = C.c_int64(); it = C.c_int64()
        _lib.check(lib.bohip_direct_best(h, C.byref(bf), bx.ctypes.data_as(dp), C.byref(ev), C.byref(it)))
        return float(bf.value), bx, int(ev.value)
    finally:
        lib.bohip_direct_destroy(h)


# NLopt.Opt properties the reference forwards with setproperty! (src/acquisition.jl:24-27).  The device ascent
# implements the first group; the second is accepted by NLopt but has no counterpart here (a warning says so);
# anything else raises, as setproperty! on an NLopt.Opt does.
_WARNED_METHODS = set()


def _warn_not_a_local_search(method):
    """:LN_* and the non-DIRECT :GN_* methods have no device counterpart: NLopt would run that algorithm (COBYLA, BOBYQA, Nelder-Mead,
    CRS, ISRES ...) from each start, here `maxeval` Latin-hypercube candidates per restart are scored in one batch and the best one is
    returned.  Said once per process and method, since the result is a candidate-set maximum, not that algorithm's."""
    if method in _WARNED_METHODS:
        return
    _WARNED_METHODS.add(method)
    warnings.warn(f"acquire_max: method :{method} is not implemented as such; maxeval Latin-hypercube candidates per restart are "
                  "scored in one device batch instead (use :LD_LBFGS or :GN_DIRECT_L for a search)", stacklevel=3)


_OPTS_USED = {"method", "restarts", "maxeval", "maxtime", "ftol_rel", "xtol_abs", "ftol_abs", "xtol_rel", "stopval"}
_OPTS_NLOPT_ONLY = {"initial_step", "population", "vector_storage", "seed", "local_optimizer", "default_initial_step"}


def _check_options(opts):
    for k in opts:
        if k in _OPTS_USED:
            continue
        if k in _OPTS_NLOPT_ONLY:
            warnings.warn(f"acquisition option {k!r} is an NLopt setting the device ascent does not implement; ignored")
        else:
            raise ValueError(f"unknown acquisition option {k!r} (the reference forwards every key to NLopt.Opt, "
                             "which rejects unknown properties)")


def acquire_max(a, model, lowerbounds, upperbounds, options, rng=None, setparams=True):
    """src/acquisition.jl:48-68: R Latin-hypercube starts (utils.jl:96-120), local search from each, keep the
    best under strict '>' (first maximum wins).  Returns (maxf, maxx).
    setparams=True is the 5-argument method (:48-51: nlopt_setup calls setparams!, :30); the BO loop has already
    called setparams! (src/BayesianOptimization.jl:184) and uses the 4-argument method on its prepared optimiser
    (:185), i.e. setparams=False -- MutualInformation's update is not idempotent."""
    lb = np.asarray(lowerbounds, dtype=np.float64)
    ub = np.asarray(upperbounds, dtype=np.float64)
    opts = options if isinstance(options, dict) else dict(vars(options))
    _check_options(opts)
    if setparams:
        setparams_(a, model)                                      # nlopt_setup :30
    method = str(opts.get("method", "LD_LBFGS")).lstrip(":")
    restarts = int(opts.get("restarts", 10))
    maxeval = int(opts.get("maxeval", 2000))
    maxtime = float(opts.get("maxtime", 0.0) or 0.0)
    maxf, maxx = -math.inf, lb.copy()                             # :55-56
    if model.nobs == 0 or restarts <= 0:
        return maxf, maxx
    derivative = len(method) > 1 and method[1] == "D" and not isinstance(a, ThompsonSamplingSimple)   # :31
    if "DIRECT" in method.upper() and not derivative:
        # :GN_DIRECT_L (the reference's default for ThompsonSamplingSimple, src/acquisition.jl:7-9) and its siblings: dividing
        # rectangles with every iteration's new points in one device call.  DIRECT ignores the start point, so `restarts` runs are
        # `restarts` repetitions (identical for a deterministic acquisition, fresh draws for a sampled one), as with NLopt.
        sval = float(opts.get("stopval", math.inf))
        thompson = isinstance(a, ThompsonSamplingSimple)
        gen = (rng or np.random.default_rng()) if thompson else None
        fused = hasattr(model, "direct_max")                      # device model: the whole search is ONE library call
        if thompson:
            def f_batch(X):                                       # x -> myrand(model, x), one draw per point (src/acquisitionfunctions.jl:108)
                mu, var = model.predict_f(X)
                return np.asarray(mu) + np.sqrt(np.maximum(np.asarray(var), 0.0)) * gen.standard_normal(np.size(mu))
        else:
            def f_batch(X):
                return model.score(a.acq_id, a.params(), X)[0]
        for _ in range(restarts):
            if fused:                                             # (the draws come from the library's counter-based generator,
                seed = int(gen.integers(0, 2 ** 63 - 1)) if thompson else 0   #  keyed by a seed taken from `rng`)
                f, x, _, _ = model.direct_max("ThompsonDraw" if thompson else a.acq_id, None if thompson else a.params(), lb, ub,
                                              max(1, maxeval), sval, maxtime, seed)
            else:
                f, x, _ = direct_l_search(f_batch, lb, ub, max(1, maxeval), sval, maxtime)
            if f > maxf:                                          # :62 strict '>'
                maxf, maxx = f, x
            if not thompson:
                break                                             # deterministic objective: every repetition is the same run
        if not np.isfinite(maxf):
            warnings.warn("acquisition returned no finite value; keeping the lower bounds as maximiser")
        return maxf, maxx
    if isinstance(a, ThompsonSamplingSimple):
        _warn_not_a_local_search(method)
        # one joint draw of the posterior at `maxeval` candidates per restart, arg-max on the device
        n = max(maxeval, 1)
        for _ in range(restarts):
            starts = latin_hypercube_sampling(lb, ub, n, rng)
            seed = int((rng or np.random.default_rng()).integers(0, 2 ** 63 - 1))
            bv, bi = model.thompson(starts, 1, seed=seed)
            if bi[0] >= 0 and bv[0] > maxf:
                maxf, maxx = float(bv[0]), starts[:, int(bi[0])].copy()
        return maxf, maxx
    acq, p = a.acq_id, a.params()
    if derivative:
        starts = latin_hypercube_sampling(lb, ub, restarts, rng)
        # maxeval: NLopt counts objective evaluations per start; the batched ascent spends one evaluation of EVERY
        # start per device pass, so the same number bounds the passes.  No other cap.
        iters = max(2, maxeval)

        ftol, xtol = float(opts.get("ftol_rel", 1e-10)), float(opts.get("xtol_abs", 1e-10))
        # ftol_abs / xtol_rel / stopval with NLopt's defaults (off); the reference's own test passes ftol_abs = eps() (test/acquisition.jl:6,9)
        fabs_, xrel, sval = float(opts.get("ftol_abs", 0.0)), float(opts.get("xtol_rel", 0.0)), float(opts.get("stopval", math.inf))
        if hasattr(model, "ascend"):                              # device model: the whole search runs in libbohip
            if hasattr(model, "set_maxtime"):
                model.set_maxtime(maxtime)                        # NLopt maxtime: per optimize call = per acquire_max here
            if hasattr(model, "set_ascent_stop"):
                model.set_ascent_stop(fabs_, xrel, sval)
            f, X, bf, bi, bx, _ = model.ascend(acq, p, lb, ub, starts, iters, ftol, xtol)
            if bi >= 0:
                return float(bf), np.array(bx)
            warnings.warn("acquisition returned no finite value; keeping the lower bounds as maximiser")
            return maxf, maxx

        def fg(X):                                                # host restatement of the same search (test models)
            return model.score_grad(acq, p, X)

        f, X = _batched_lbfgs_ascent(fg, starts, lb, ub, iters, ftol_rel=ftol, xtol_abs=xtol, ftol_abs=fabs_, xtol_rel=xrel,
                                     stopval=sval)
    else:
        _warn_not_a_local_search(method)
        n = max(restarts, min(maxeval * restarts, 1 << 20))
        X = latin_hypercube_sampling(lb, ub, n, rng)
        f, _, _ = model.score(acq, p, X)
    for j in range(X.shape[1]):                                   # :58-66, strict '>' => first maximum wins
        if f[j] > maxf:
            maxf, maxx = float(f[j]), X[:, j].copy()
    if not np.isfinite(maxf):
        warnings.warn("acquisition returned no finite value; keeping the lower bounds as maximiser")
    return maxf, maxx


def acquire_model_max(o, options=None):                           # :45-47
    return acquire_max(MaxMean(), o.model, o.lowerbounds, o.upperbounds, options or o.acquisitionoptions, o.rng)
