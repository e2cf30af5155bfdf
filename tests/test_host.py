"""CPU tests of the host mirror: the reference's own host-side tests restated (test/utils.jl, test/BayesianOptimization.jl,
test/acquisition.jl option plumbing) plus the pure-host parts of acquisition.py."""
import math
import time

import warnings
import numpy as np
import pytest

import bohip
from bohip import acquisition as A
from bohip.utils import DurationCounter, IterationCounter, init_, isdone, latin_hypercube_sampling, step_


def test_counters_like_reference_test_utils():                      # reference test/utils.jl:4-23
    it = IterationCounter(0, 0, 10)
    for _ in range(10):
        step_(it)
    assert isdone(it) is True
    bohip.maxiterations_(it, 40)
    assert it.N == 40
    init_(it)
    assert it.c == 0 and it.i == 10
    now = time.time()
    du = DurationCounter(now, 0.05, now, now + 0.05)
    time.sleep(0.06)
    assert isdone(du)
    bohip.maxduration_(du, 0.15)
    init_(du)
    assert not isdone(du)
    time.sleep(0.16)
    assert isdone(du)


def test_merge_with_defaults_errors():                               # reference test/BayesianOptimization.jl:16-23
    f = lambda x: x[0] + x[1]
    l, u = [-1, -2], [3, 4]
    with pytest.raises(ValueError):
        bohip.merge_with_defaults(f, l, u, dict(func=lambda x: x[0]))
    with pytest.raises(ValueError):
        bohip.merge_with_defaults(f, l, u, dict(lowerbounds=[], upperbounds=[-1, -100]))
    with pytest.raises(ValueError):
        bohip.merge_with_defaults(f, l, u, dict(hello="world"))
    with pytest.raises(ValueError):
        bohip.merge_with_defaults(f, [1.0], [3.0, 4.0], {})


def test_merge_with_defaults_ordering_with_explicit_model():         # :6-15 (model passed, so no device is needed)
    f = lambda x: x[0] + x[1]
    sentinel = object()
    args, kwargs = bohip.merge_with_defaults(f, [-1, -2], [3, 4], dict(sense=bohip.Max, model=sentinel,
                                                                      acquisition=bohip.ProbabilityOfImprovement(),
                                                                      maxiterations=20))
    assert len(args) == 6 and args[0] is f and args[1] is sentinel
    assert isinstance(args[2], bohip.ProbabilityOfImprovement) and isinstance(args[3], bohip.MAPGPOptimizer)
    assert kwargs["sense"] == bohip.Max and kwargs["maxiterations"] == 20


def test_defaultoptions():                                           # src/acquisition.jl:4-9, test/acquisition.jl:4-9
    assert bohip.defaultoptions(bohip.ElasticGPE, bohip.ExpectedImprovement) == dict(method="LD_LBFGS", restarts=10, maxeval=2000)
    assert bohip.defaultoptions(bohip.ElasticGPE, bohip.ThompsonSamplingSimple) == dict(method="GN_DIRECT_L", restarts=1, maxeval=2000)
    merged = {**bohip.defaultoptions(bohip.ElasticGPE, bohip.MaxMean), **dict(maxtime=3.0, ftol_abs=np.finfo(float).eps)}
    assert merged["maxeval"] == 2000 and merged["maxtime"] == 3.0 and merged["ftol_abs"] == np.finfo(float).eps


def test_host_functors_match_oracle_formulas(orc):
    ei, pi = bohip.ExpectedImprovement(0.5), bohip.ProbabilityOfImprovement(0.5)
    ucb, mi = bohip.UpperConfidenceBound(bohip.NoBetaScaling(), 2.5), bohip.MutualInformation(alpha=1.69, gamma_hat=0.4)
    for mu, s2 in [(0.3, 4.0), (0.7, 0.0), (0.3, 0.0), (-3.0, 0.2), (5.0, 1e-6)]:
        assert ei(mu, s2) == orc.acq("EI", [0.5], mu, s2)
        assert pi(mu, s2) == orc.acq("PI", [0.5], mu, s2)
        assert ucb(mu, s2) == orc.acq("UCB", [2.5], mu, s2)
        assert mi(mu, s2) == pytest.approx(orc.acq("MI", [1.3, 0.4], mu, s2), rel=1e-15)
    assert bohip.ExpectedImprovement().tau == -math.inf               # ExpectedImprovement(; tau = -Inf)


class FakeModel:
    """Just enough of the model contract (dims / y / x) for setparams_ to run on the host."""
    def __init__(self, d, y):
        self.dim, self._y = d, np.asarray(y, float)
        self.x = np.zeros((d, len(y)))
    y = property(lambda s: s._y)
    nobs = property(lambda s: s._y.size)


def test_setparams_rules():                                          # src/acquisitionfunctions.jl:44-46, 91-95
    m = FakeModel(8, np.linspace(-1, 2, 3000))
    ei = bohip.ExpectedImprovement()
    bohip.setparams_(ei, m)
    assert ei.tau == 2.0                                             # test/warmstart.jl:64: tau == maximum(y)
    m2 = FakeModel(8, [0.5])
    bohip.setparams_(ei, m2)
    assert ei.tau == 2.0                                             # monotone: max(maxy, tau)
    ucb = bohip.UpperConfidenceBound()
    bohip.setparams_(ucb, m)
    assert ucb.beta_t == pytest.approx(10.152008469453344, rel=1e-15)  # SURVEY.md A6 probe (N=3000, d=8)
    bohip.setparams_(ucb, FakeModel(2, []))
    assert ucb.beta_t == pytest.approx(math.sqrt(2 * math.log(math.pi ** 2 / 0.3)), rel=1e-15)   # nobs == 0 -> 1
    fixed = bohip.UpperConfidenceBound(bohip.NoBetaScaling(), 3.0)
    bohip.setparams_(fixed, m)
    assert fixed.beta_t == 3.0
    assert bohip.setparams_(bohip.ThompsonSamplingSimple(), m) is None


def test_lhs_initialiser_and_iterators():
    rng = np.random.default_rng(0)
    S = latin_hypercube_sampling([-5.0, 0.0], [10.0, 15.0], 20, rng)
    assert S.shape == (2, 20)
    for k, (lo, hi) in enumerate([(-5, 10), (0, 15)]):
        strata = np.floor((S[k] - lo) / ((hi - lo) / 20)).astype(int)
        assert sorted(strata) == list(range(20))
    it = bohip.ScaledLHSIterator([0.0], [1.0], 7, rng)
    assert len(it) == 7 and len(list(it)) == 7
    sob = bohip.ScaledSobolIterator([-5.0, 0.0], [10.0, 15.0], 10)
    pts = np.array(list(sob))
    assert pts.shape == (10, 2) and np.all(pts >= [-5, 0]) and np.all(pts <= [10, 15])
    assert len(bohip.ScaledSobolIterator([0.0], [1.0], 0)) == 0


def test_batched_lbfgs_on_a_known_concave_problem():
    """The lock-step projected L-BFGS (role of NLopt :LD_LBFGS) on f_r(x) = -|x - c_r|^2 with box bounds."""
    rng = np.random.default_rng(3)
    d, R = 4, 9
    C_ = rng.uniform(-2, 2, (d, R))
    lb, ub = np.full(d, -1.0), np.full(d, 1.0)

    def fg(X):
        return -((X - C_) ** 2).sum(0), -2 * (X - C_)

    f, X = A._batched_lbfgs_ascent(fg, rng.uniform(-1, 1, (d, R)), lb, ub, maxeval=200)
    np.testing.assert_allclose(X, np.clip(C_, -1, 1), atol=1e-6)     # box-constrained maximiser = projection of c


# ---- acquire_max on the host, with the ORACLE standing in for the device model (tests may use the oracle) ------------
class OracleModel:
    """Implements the slice of the ElasticGPE interface that acquisition.py touches (score / score_grad / thompson /
    predict_f / nobs / dim / x / y) on top of the CPU oracle, so the host-side search logic runs without a GPU."""

    def __init__(self, orc, X, y, ll, lsig=0.0, lnoise=-2.0, beta=0.0):
        self.orc, self.X, self._y, self.ll, self.lsig, self.beta = orc, X, y, np.asarray(ll, float), lsig, beta
        self.L, self.al = orc.fit(X, y, ll, lsig, lnoise, beta)
        self.dim = X.shape[1]
        self.calls = []

    y = property(lambda s: s._y)
    x = property(lambda s: np.asfortranarray(s.X.T))
    nobs = property(lambda s: len(s._y))

    def _pad(self, params):
        return list(params) if len(params) else [0.0]

    def predict_f(self, xs):
        xs = np.asarray(xs, float)
        xs = xs.reshape(self.dim, -1)
        return self.orc.predict(self.X, self.ll, self.lsig, self.beta, self.L, self.al, np.ascontiguousarray(xs.T))

    def score(self, acq, params, xs, want_scores=True):
        xs = np.asarray(xs, float).reshape(self.dim, -1)
        self.calls.append(("score", xs.shape[1]))
        sc, bv, bi = self.orc.score(self.X, self.ll, self.lsig, self.beta, self.L, self.al, acq, self._pad(params),
                                    np.ascontiguousarray(xs.T))
        return sc, bv, bi

    def score_grad(self, acq, params, xs):
        xs = np.asarray(xs, float).reshape(self.dim, -1)
        self.calls.append(("score_grad", xs.shape[1]))
        sc, g = self.orc.score_grad(self.X, self.ll, self.lsig, self.beta, self.L, self.al, acq, self._pad(params),
                                    np.ascontiguousarray(xs.T))
        return sc, np.asfortranarray(g.T)


def test_acquire_max_reference_known_answer_on_host(orc):           # reference test/acquisition.jl:2,11-12
    m = OracleModel(orc, np.array([[1.0]]), np.array([2.0]), [1.0])  # GPE([1.0],[2.0],MeanZero(),SEIso(1.0,0.0))
    opts = {**bohip.defaultoptions(type(m), bohip.MaxMean), "restarts": 10}
    maxf, maxx = bohip.acquire_max(bohip.MaxMean(), m, [-5.0], [5.0], opts, np.random.default_rng(0))
    assert maxx == pytest.approx([1.0], abs=1e-5)
    assert maxf == pytest.approx(2 / (1 + math.exp(-4.0)), rel=1e-9)
    assert all(c[0] == "score_grad" and c[1] == 10 for c in m.calls)   # all 10 restarts ascend in lock-step: one call per evaluation


def test_acquire_max_gradient_free_and_bounds(orc):
    rng = np.random.default_rng(4)
    X = rng.random((30, 2)) * 2 - 1
    y = -((X - 0.3) ** 2).sum(1)
    m = OracleModel(orc, X, y, [-0.5, -0.5], 0.0, -3.0, float(y.mean()))
    ei = bohip.ExpectedImprovement()
    f0, x0 = bohip.acquire_max(ei, m, [-1, -1], [1, 1], dict(method="GN_DIRECT_L", restarts=1, maxeval=500), np.random.default_rng(1))
    assert ei.tau == y.max()                                            # setparams! ran (src/acquisition.jl:30)
    # :GN_DIRECT_L (the reference's default for ThompsonSamplingSimple, src/acquisition.jl:7-9): dividing rectangles, every iteration's
    # new points in one batch -- never more than maxeval evaluations in total, the first one the centre of the box
    n_direct = [c[1] for c in m.calls]
    assert all(c[0] == "score" for c in m.calls) and n_direct[0] == 1 and 2 < len(n_direct) < 120 and 400 < sum(n_direct) <= 500
    m.calls.clear()
    from bohip import acquisition as _acq
    _acq._WARNED_METHODS.discard("LN_COBYLA")
    with pytest.warns(UserWarning, match="LN_COBYLA is not implemented as such"):     # said once per process and method
        fl, xl = bohip.acquire_max(ei, m, [-1, -1], [1, 1], dict(method="LN_COBYLA", restarts=1, maxeval=500), np.random.default_rng(1))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        bohip.acquire_max(ei, m, [-1, -1], [1, 1], dict(method="LN_COBYLA", restarts=1, maxeval=500), np.random.default_rng(1))
    m.calls.pop()
    assert m.calls[-1] == ("score", 500)                                # other derivative-free methods: maxeval Latin-hypercube candidates in ONE batch
    assert f0 >= fl * 0.999                                             # 500 evaluations placed by DIRECT-L beat 500 scattered ones here
    f1, x1 = bohip.acquire_max(ei, m, [-1, -1], [1, 1], dict(method="LD_LBFGS", restarts=16, maxeval=60), np.random.default_rng(1))
    assert f1 >= f0 * 0.999                                             # ascent from 16 starts is at least as good as 500 samples
    assert np.all(x1 >= -1) and np.all(x1 <= 1)
    # the maximiser of a tight box sits on the bound and the returned point respects it
    f2, x2 = bohip.acquire_max(bohip.MaxMean(), m, [0.6, 0.6], [1.0, 1.0], dict(method="LD_LBFGS", restarts=4, maxeval=60),
                               np.random.default_rng(2))
    assert x2 == pytest.approx([0.6, 0.6], abs=1e-6)


def test_direct_l_finds_the_known_maxima():
    """The dividing-rectangles search on closed-form objectives: Branin's minimum (the reference's test function, test/branin.jl:1-7)
    to 1e-6 within the reference's default budget of 2000 evaluations, the box respected, the budget never exceeded, NaN never wins,
    stopval honoured."""
    from bohip.acquisition import _batched_direct_l

    def neg_branin(X):
        x1, x2 = X
        return -((x2 - 5.1 / (4 * math.pi ** 2) * x1 ** 2 + 5 / math.pi * x1 - 6) ** 2 + 10 * (1 - 1 / (8 * math.pi)) * np.cos(x1) + 10)

    sizes = []
    f, x, ev = _batched_direct_l(lambda X: (sizes.append(X.shape[1]), neg_branin(X))[1], [-5.0, 0.0], [10.0, 15.0], 2000)
    assert f == pytest.approx(-0.397887, abs=1e-5) and ev <= 2000 and sum(sizes) == ev and len(sizes) < 150
    assert min(np.hypot(*(x - m_)) for m_ in ([-math.pi, 12.275], [math.pi, 2.275], [9.42478, 2.475])) < 1e-2
    for d in (1, 4):
        f, x, ev = _batched_direct_l(lambda X: -np.sum((X - 0.37) ** 2, axis=0), [-1.0] * d, [2.0] * d, 600)
        assert f > -1e-4 and np.all(x >= -1) and np.all(x <= 2) and ev <= 600
    f, x, ev = _batched_direct_l(lambda X: np.where(X[0] > 0.9, np.nan, X[0]), [0.0], [1.0], 100)
    assert np.isfinite(f) and x[0] <= 0.9
    f, x, ev = _batched_direct_l(lambda X: -np.sum(X ** 2, axis=0), [-1.0, -1.0], [1.0, 1.0], 2000, stopval=-0.5)
    assert ev == 1                                                      # the centre already reaches stopval
    f, x, ev = _batched_direct_l(lambda X: -np.sum(X ** 2, axis=0), [-1.0, -1.0], [1.0, 1.0], 7)
    assert ev <= 7


def test_thompson_default_options_run_direct_l_on_posterior_draws(orc):
    """ThompsonSamplingSimple with the reference's defaultoptions (:GN_DIRECT_L, restarts = 1, maxeval = 2000): x -> myrand(model, x)
    is a fresh posterior draw per evaluation (src/acquisitionfunctions.jl:108, src/models/gp.jl:6); the search never asks for more than
    maxeval of them and lands near the posterior's high region."""
    rng = np.random.default_rng(4)
    X = rng.random((40, 2)) * 2 - 1
    y = -((X - 0.3) ** 2).sum(1)
    m = OracleModel(orc, X, y, [-0.5, -0.5], 0.0, -3.0, float(y.mean()))
    asked = []
    pf = m.predict_f
    m.predict_f = lambda xs: (asked.append(np.asarray(xs).reshape(2, -1).shape[1]), pf(xs))[1]
    opts = bohip.defaultoptions(bohip.ElasticGPE, bohip.ThompsonSamplingSimple)
    f, x = bohip.acquire_max(bohip.ThompsonSamplingSimple(), m, [-1, -1], [1, 1], opts, np.random.default_rng(3))
    assert sum(asked) <= 2000 and len(asked) > 3 and np.all(np.abs(x) <= 1)
    mu_x, _ = pf(x.reshape(2, 1))
    assert mu_x[0] > np.quantile(y, 0.75)                               # a draw-maximiser sits where the posterior mean is high


def test_acquire_max_empty_model_returns_reference_initial_state():
    class Empty:
        nobs, dim = 0, 2
        y = np.zeros(0)
        x = np.zeros((2, 0))
    maxf, maxx = bohip.acquire_max(bohip.UpperConfidenceBound(), Empty(), [-1.0, 0.0], [1.0, 2.0], dict(restarts=3))
    assert maxf == -math.inf and list(maxx) == [-1.0, 0.0]              # maxf = -Inf, maxx = lowerbounds (src/acquisition.jl:55-56)


def test_mutual_information_gamma_update(orc):                          # src/acquisitionfunctions.jl:131-140
    rng = np.random.default_rng(8)
    X = rng.random((12, 2)); y = rng.random(12)
    m = OracleModel(orc, X, y, [-0.3, -0.3])
    mi = bohip.MutualInformation()
    _, s2_last = m.predict_f(X[-1])
    bohip.setparams_(mi, m)
    assert mi.gamma_hat == pytest.approx(float(s2_last[0]), rel=1e-12)  # gamma_hat += sigma^2(x_last)
    bohip.setparams_(mi, m)
    assert mi.gamma_hat == pytest.approx(2 * float(s2_last[0]), rel=1e-12)


# ---- the loop calls setparams! exactly once per iteration (src/BayesianOptimization.jl:184-185) ----------------------
class GrowingOracleModel(OracleModel):
    """OracleModel + update!: enough of the model interface for boptimize_ to run on the CPU oracle."""

    def __init__(self, orc, d, ll):
        self.orc, self.X, self._y, self.ll, self.lsig, self.beta = orc, np.zeros((0, d)), np.zeros(0), np.asarray(ll, float), 0.0, 0.0
        self.dim, self.calls = d, []

    def append_(self, x, y):
        x = np.asarray(x, float).reshape(self.dim, -1)
        self.X = np.concatenate([self.X, x.T])
        self._y = np.concatenate([self._y, np.atleast_1d(y)])
        self.L, self.al = self.orc.fit(self.X, self._y, self.ll, self.lsig, -2.0, self.beta)
        return self


def test_loop_calls_setparams_once_per_iteration(orc):
    class CountingMI(bohip.MutualInformation):
        calls = 0

        def _setparams(self, model):
            CountingMI.calls += 1
            return super()._setparams(model)

    m = GrowingOracleModel(orc, 2, [-0.3, -0.3])
    acq = CountingMI()
    k = 3
    o = bohip.BOpt(lambda x: -float(((x - 0.2) ** 2).sum()), m, acq, bohip.NoModelOptimizer(), [-1.0, -1.0], [1.0, 1.0],
                   maxiterations=4 + k, initializer_iterations=4, verbosity=bohip.Silent,
                   acquisitionoptions=dict(method="LD_LBFGS", restarts=3, maxeval=20), rng=np.random.default_rng(0))
    assert CountingMI.calls == 1                                       # ctor: nlopt_setup (src/acquisition.jl:30)
    g0 = acq.gamma_hat
    assert g0 == 0.0                                                   # empty model: gamma_hat = 0 (:133)
    # replay what the reference's flow accumulates: gamma_hat += sigma^2(x_last) ONCE per iteration, before the search
    expected = []
    orig = bohip.MutualInformation._setparams

    def spy(self, model):
        r = orig(self, model)
        expected.append(self.gamma_hat)
        return r

    bohip.MutualInformation._setparams = spy
    try:
        bohip.boptimize_(o)
    finally:
        bohip.MutualInformation._setparams = orig
    assert CountingMI.calls == 1 + k                                   # one per loop iteration, none inside acquire_max
    assert len(expected) == k and acq.gamma_hat == expected[-1]
    inc = np.diff([g0] + expected)
    assert np.all(inc >= 0) and acq.gamma_hat == pytest.approx(float(np.sum(inc)), rel=1e-12)


def test_acquisition_options_are_checked_like_nlopt(orc):
    m = OracleModel(orc, np.array([[1.0]]), np.array([2.0]), [1.0])
    with pytest.raises(ValueError):
        bohip.acquire_max(bohip.MaxMean(), m, [-5.0], [5.0], dict(method="LD_LBFGS", restarts=2, maxevil=10))
    with pytest.warns(UserWarning):                                    # an NLopt setting without a counterpart here
        bohip.acquire_max(bohip.MaxMean(), m, [-5.0], [5.0], dict(method="LD_LBFGS", restarts=2, maxeval=10, initial_step=0.1))
    # ftol_abs / xtol_rel / stopval ARE implemented (the reference's test passes ftol_abs = eps(), test/acquisition.jl:6,9):
    # no warning, and a loose ftol_abs / a reachable stopval stops the search after fewer evaluations than the default
    import warnings as _w
    with _w.catch_warnings():
        _w.simplefilter("error")
        m.calls.clear()
        bohip.acquire_max(bohip.MaxMean(), m, [-5.0], [5.0], dict(method="LD_LBFGS", restarts=2, maxeval=300, ftol_abs=np.finfo(float).eps))
        n_eps = len(m.calls)
        m.calls.clear()
        f_loose, _ = bohip.acquire_max(bohip.MaxMean(), m, [-5.0], [5.0], dict(method="LD_LBFGS", restarts=2, maxeval=300, ftol_abs=0.5))
        n_loose = len(m.calls)
        m.calls.clear()
        f_stop, _ = bohip.acquire_max(bohip.MaxMean(), m, [-5.0], [5.0], dict(method="LD_LBFGS", restarts=2, maxeval=300, stopval=0.1))
        n_stop = len(m.calls)
        m.calls.clear()
        bohip.acquire_max(bohip.MaxMean(), m, [-5.0], [5.0], dict(method="LD_LBFGS", restarts=2, maxeval=300, xtol_rel=0.5))
        n_xrel = len(m.calls)
    assert n_loose <= n_eps and n_stop <= n_eps and n_xrel <= n_eps and f_stop >= 0.1
    # maxeval is honoured as given (no hidden cap): 300 evaluations allowed, the search stops on its own tolerance first
    m.calls.clear()
    bohip.acquire_max(bohip.MaxMean(), m, [-5.0], [5.0], dict(method="LD_LBFGS", restarts=2, maxeval=300))
    assert 2 <= len(m.calls) <= 300
