"""GPU parity tests at the sizes bench.py reports, on factors the device had no part in (round 5).

What round 4's verdict listed as sampling holes, closed here inside the driver-run `-m gpu` suite:
  * C4 (N = 10^4, d = 16): an INDEPENDENT factor (LAPACK dpotrf through SciPy -- what Julia's LinearAlgebra calls behind
    GaussianProcesses.jl's update, reference src/models/gp.jl:11-18 -- itself pinned against the C oracle's own factorisation at a
    size both afford), the WHOLE device factor against it, ALL 640 candidates scored by the oracle on that factor, arg-max asserted;
  * the stress variant (kappa ~ 5e10) on all 4096 candidates, arg-max over the full set;
  * the small-batch path (round 5: k_small_v + k_small_u, kernels_small.hip) at the bench's `default_usage` sizes against the oracle's analytic
    value + gradient, and bohip_gp_acquire_max at that shape against SciPy's L-BFGS-B on the oracle (reference
    src/acquisition.jl:54-68);
  * one seed each of the randomised sweeps tools/fuzz_parity.py / tools/fuzz_large.py.
Tolerances as in tests/test_parity_gpu.py (north_star: 1e-6 relative + the documented cancellation floors; arg-max exact).
"""
import math
import os

import numpy as np
import pytest

from conftest import synth, var_tol

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps
NTH = min(128, os.cpu_count() or 8)


@pytest.fixture(scope="module")
def bohip():
    import bohip as b
    from bohip import _lib

    assert _lib.load().bohip_device_count() > 0, "GPU tests need an MI355X; libbohip has no CPU fallback"
    return b


def make_model(bohip, X, y, ll, lsig, lnoise, beta, kern="SEArd", capacity=None):
    K = {"SEArd": bohip.SEArd, "SEIso": bohip.SEIso, "Mat52Ard": bohip.Mat52Ard}[kern]
    m = bohip.ElasticGPE(X.shape[1], mean=bohip.MeanConst(beta), kernel=K(ll, lsig), logNoise=lnoise,
                         capacity=capacity or max(len(y), 1))
    m.append_(X.T, y)
    return m


def mu_floor(alpha, s2f):
    return 64 * EPS * s2f * np.abs(alpha).sum()


def lapack_fit(orc, X, y, ll, lsig, lnoise, beta):
    """cK by the C oracle's entry loop, its factor by LAPACK dpotrf, alpha by LAPACK substitutions: nothing of the device."""
    import scipy.linalg as sl

    cK = orc.build_cK(X, ll, lsig, lnoise)
    L = np.ascontiguousarray(sl.cholesky(cK, lower=True, overwrite_a=True, check_finite=False))   # (SciPy may hand back Fortran order)
    alpha = sl.cho_solve((L, True), y - beta, check_finite=False)
    return L, alpha


def test_lapack_factor_is_the_c_oracles_factor(orc):
    """The independent factor of the C4 test below against the C restatement's own (row-by-row) factorisation at N = 1500."""
    X, y, _ = synth(1500, 16, 1, seed=4)
    ll = np.full(16, math.log(0.7))
    L_c, a_c = orc.fit(X, y, ll, 0.0, -2.0, 0.0)
    L_l, a_l = lapack_fit(orc, X, y, ll, 0.0, -2.0, 0.0)
    np.testing.assert_allclose(L_l, L_c, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(a_l, a_c, rtol=1e-8, atol=1e-10 * np.abs(a_c).max())


def test_c4_whole_factor_and_all_640_candidates_on_an_independent_factor(bohip, orc):
    """BASELINE configs[3] (N = 10^4, d = 16, SEArd): the device's factor, alpha, mu, sigma^2 and EI against a model the device
    never touched -- LAPACK's factor of the oracle's cK -- over the WHOLE factor and ALL candidates; the device's winner is the
    first arg-max of the oracle's scores."""
    N, d, R = 10000, 16, 640
    X, y, Xs = synth(N, d, R, seed=4)
    ll = np.full(d, math.log(0.7))
    m = make_model(bohip, X, y, ll, 0.0, -2.0, 0.0)
    L, alpha = lapack_fit(orc, X, y, ll, 0.0, -2.0, 0.0)
    Lg = m.factor()
    err = np.abs(Lg - L)
    print(f"C4 factor: max |dL| {err.max():.3e}, max |dL|/|L| where |L| > 1e-6: {(err / np.maximum(np.abs(L), 1e-6)).max():.3e}")
    np.testing.assert_allclose(Lg, L, rtol=1e-9, atol=1e-11)
    ag = m.alpha()
    np.testing.assert_allclose(ag, alpha, rtol=1e-6, atol=1e-9 * np.abs(alpha).max())
    tau = float(y.max())
    sc, bv, bi = m.score("EI", [tau], Xs.T)
    mu, var = m.predict_f(Xs.T)
    mu_o, var_o = orc.predict(X, ll, 0.0, 0.0, L, alpha, Xs, nthreads=NTH)
    sc_o, bv_o, bi_o = orc.score(X, ll, 0.0, 0.0, L, alpha, "EI", [tau], Xs, nthreads=NTH)
    assert np.all(np.abs(mu - mu_o) <= 1e-6 * np.abs(mu_o) + mu_floor(alpha, 1.0))
    assert np.all(np.abs(var - var_o) <= var_tol(var_o, N, 1.0))
    assert np.all(np.abs(sc - sc_o) <= 1e-6 * np.abs(sc_o) + mu_floor(alpha, 1.0) + var_tol(var_o, N, 1.0, rel=0) + 1e-13)
    assert bi == bi_o == int(np.argmax(sc_o)), (bi, bi_o, np.sort(sc_o)[-3:])
    assert abs(bv - bv_o) <= 1e-6 * abs(bv_o)


def test_c4_small_batch_pass_vs_oracle(bohip, orc):
    """The small-batch pass at N = 10^4, d = 16, R = 10 (bench.py's `small_batch_c4`):
    value + analytic gradient against the oracle on the LAPACK factor."""
    N, d = 10000, 16
    X, y, Xs = synth(N, d, 10, seed=4)
    ll = np.full(d, math.log(0.7))
    m = make_model(bohip, X, y, ll, 0.0, -2.0, 0.0)
    L, alpha = lapack_fit(orc, X, y, ll, 0.0, -2.0, 0.0)
    for acq, p in [("EI", [float(y.max())]), ("UCB", [2.0])]:
        sc, g = m.score_grad(acq, p, Xs.T)
        sc_o, g_o = orc.score_grad(X, ll, 0.0, 0.0, L, alpha, acq, p, Xs)
        _, var_o = orc.predict(X, ll, 0.0, 0.0, L, alpha, Xs, nthreads=8)
        fl = mu_floor(alpha, 1.0) + (2.0 * np.sqrt(var_tol(var_o, N, 1.0, rel=0)) if acq == "UCB" else var_tol(var_o, N, 1.0, rel=0)) + 1e-13
        assert np.all(np.abs(sc - sc_o) <= 1e-6 * np.abs(sc_o) + fl), (acq, np.abs(sc - sc_o).max())
        np.testing.assert_allclose(g.T, g_o, rtol=1e-6, atol=1e-8 * np.abs(g_o).max(), err_msg=acq)
        sv, bv, bi = m.score(acq, p, Xs.T)
        np.testing.assert_array_equal(sv, sc)                              # value path == gradient path, bit for bit
        assert bi == int(np.argmax(sc_o)) and bv == sv[bi]
        assert m.score(acq, p, Xs[3:4].T)[0][0] == sv[3]                   # batch == single, bit for bit (test/acquisitionfunctions.jl:8-11)


@pytest.fixture(scope="module")
def headline(bohip, orc):
    """The headline model of bench.py (N = 3000, d = 8, SEArd, l = 0.5, logNoise = -2) with the oracle's own factor."""
    from bench import DIM, synth as bsynth

    X, y = bsynth(0)
    ll = np.full(DIM, math.log(0.5))
    m = make_model(bohip, X, y, ll, 0.0, -2.0, 0.0)
    L, alpha = orc.fit(X, y, ll, 0.0, -2.0, 0.0)
    return m, X, y, ll, L, alpha


@pytest.mark.parametrize("R", [1, 10, 16, 17, 96, 190])
def test_small_batch_pass_at_the_headline_model_vs_oracle(bohip, orc, headline, R):
    """score_grad at N = 3000, d = 8 for the batch sizes of the reference's defaults (R = 10: `default_usage`; 1; one and two passes
    of 16 right-hand sides; the last small-batch size, 96; 190 = the split-K schedule): every value and gradient against the oracle,
    value path == gradient path and batch == single bit for bit, arg-max = the oracle's."""
    m, X, y, ll, L, alpha = headline
    Xs = np.random.default_rng(100 + R).random((R, 8))
    Xs[0] = X[17] + 1e-4                                                  # one candidate beside an observation: sigma^2 cancels
    for acq, p in [("UCB", [10.152008469453344]), ("EI", [float(y.max())]), ("MaxMean", [])]:
        sc, g = m.score_grad(acq, p, Xs.T)
        sc_o, g_o = orc.score_grad(X, ll, 0.0, 0.0, L, alpha, acq, p if p else [0.0], Xs)
        _, var_o = orc.predict(X, ll, 0.0, 0.0, L, alpha, Xs, nthreads=8)
        vt0 = var_tol(var_o, 3000, 1.0, rel=0)
        fl = mu_floor(alpha, 1.0) + (abs(p[0]) * np.sqrt(vt0) if acq == "UCB" else vt0) + 1e-13
        assert np.all(np.abs(sc - sc_o) <= 1e-6 * np.abs(sc_o) + fl), (acq, R, np.abs(sc - sc_o).max())
        good = var_o > 1e3 * var_tol(var_o, 3000, 1.0)
        np.testing.assert_allclose(g.T[good], g_o[good], rtol=1e-6, atol=1e-8 * np.abs(g_o).max(), err_msg=f"{acq} R={R}")
        sv, bv, bi = m.score(acq, p, Xs.T)
        np.testing.assert_array_equal(sv, sc)
        top = np.sort(sc_o)[-2:]
        if R == 1 or top[1] - top[0] > 2 * fl.max() + 1e-6 * abs(top[1]):
            assert bi == int(np.argmax(sc_o))
        assert bv == sv[bi] and bi == int(np.argmax(sv))
        if R <= 96:
            j = R // 2
            assert m.score(acq, p, Xs[j:j + 1].T)[0][0] == sv[j]


@pytest.mark.parametrize("N,d,R,kern", [(600, 40, 5, "SEArd"), (900, 64, 20, "SEArd"), (700, 33, 33, "Mat52Ard"), (300, 1, 7, "SEIso"), (1300, 3, 50, "SEArd")])
def test_small_batch_pass_wide_and_odd_dimensions(bohip, orc, N, d, R, kern):
    """The small-batch kernels are instantiated per dimension class (2, 4, 8, 16, 32, 64) and fetch their gradient records two dimensions at
    a time: odd d, d > 32 (one trip of the final's loop per 32 dimensions), Mat52 (its own gradient factor), several passes of 16 candidates."""
    X, y, Xs = synth(N, d, R, seed=N + d)
    nl = 1 if kern == "SEIso" else d
    ll = np.linspace(0.2, 0.6, nl) if d > 16 else np.linspace(-0.8, -0.4, nl)
    llp = ll if nl > 1 else float(ll[0])
    lsig, lnoise, beta = 0.15, -1.8, 0.1
    L, alpha = orc.fit(X, y, llp, lsig, lnoise, beta, kern=kern)
    m = make_model(bohip, X, y, llp, lsig, lnoise, beta, kern=kern)
    s2f = math.exp(2 * lsig)
    for acq, p in [("EI", [float(np.median(y))]), ("UCB", [2.0])]:
        sc, g = m.score_grad(acq, p, Xs.T)
        sc_o, g_o = orc.score_grad(X, llp, lsig, beta, L, alpha, acq, p, Xs, kern=kern)
        _, var_o = orc.predict(X, llp, lsig, beta, L, alpha, Xs, kern=kern, nthreads=8)
        vt0 = var_tol(var_o, N, s2f, rel=0)
        fl = mu_floor(alpha, s2f) + (2.0 * np.sqrt(vt0) if acq == "UCB" else vt0) + 1e-13
        assert np.all(np.abs(sc - sc_o) <= 1e-6 * np.abs(sc_o) + fl), (acq, np.abs(sc - sc_o).max())
        good = var_o > 1e3 * var_tol(var_o, N, s2f)
        np.testing.assert_allclose(g.T[good], g_o[good], rtol=1e-6, atol=1e-8 * np.abs(g_o).max(), err_msg=f"{acq} d={d}")
        sv, bv, bi = m.score(acq, p, Xs.T)
        np.testing.assert_array_equal(sv, sc)
        assert bi == int(np.argmax(sv)) and bv == sv[bi]
        assert m.score(acq, p, Xs[R // 2:R // 2 + 1].T)[0][0] == sv[R // 2]


def test_default_usage_acquire_max_vs_scipy_on_the_oracle(bohip, orc, headline):
    """bench.py's `default_usage` (the reference's defaults: 10 Latin-hypercube starts, :LD_LBFGS with the box bounds, UCB at
    BrochuBetaScaling's beta_t, reference src/acquisition.jl:4-6,54-68) on the headline model: per start the device's end point is
    a KKT point of the ORACLE's objective carrying the oracle's value, never below its start; SciPy's L-BFGS-B on the oracle from
    the same starts finds no better best-of-starts by more than the plateau tolerance, and the device needs no more passes than
    SciPy needs evaluations per start times two."""
    from scipy.optimize import minimize

    from bench import lhs

    m, X, y, ll, L, alpha = headline
    bt = 10.152008469453344
    starts = np.asfortranarray(lhs(10, seed=7).T)
    lb, ub = np.zeros(8), np.ones(8)
    fd, Xd, bf, bi, bx, ev = m.ascend("UCB", [bt], lb, ub, starts, 2000)
    assert 2 <= ev <= 60, ev                                               # round 3 needed 228 passes here (tightened against SciPy below)
    sc_o, g_o = orc.score_grad(X, ll, 0.0, 0.0, L, alpha, "UCB", [bt], np.ascontiguousarray(Xd.T))
    np.testing.assert_allclose(fd, sc_o, rtol=1e-6, atol=1e-9)
    pg = np.where(((Xd.T <= 0) & (g_o < 0)) | ((Xd.T >= 1) & (g_o > 0)), 0.0, g_o)
    f0, g0 = orc.score_grad(X, ll, 0.0, 0.0, L, alpha, "UCB", [bt], np.ascontiguousarray(starts.T))
    assert np.abs(pg).max() <= 2e-4 * np.abs(g0).max(), (np.abs(pg).max(), np.abs(g0).max())
    assert np.all(fd >= f0 - 1e-12) and bf == fd.max() and bi == int(np.argmax(fd))
    np.testing.assert_array_equal(bx, Xd[:, bi])
    nf, fs = [], []
    for r in range(10):
        cnt = [0]

        def negfg(x):
            cnt[0] += 1
            s, g = orc.score_grad(X, ll, 0.0, 0.0, L, alpha, "UCB", [bt], x[None, :].copy())
            return -float(s[0]), -g[0]

        res = minimize(negfg, starts[:, r], jac=True, method="L-BFGS-B", bounds=[(0, 1)] * 8, options=dict(maxiter=2000, ftol=1e-10, gtol=1e-10))
        nf.append(cnt[0]); fs.append(-res.fun)
    fs = np.array(fs)
    print(f"default_usage: device {ev} passes, best {bf:.6f}; SciPy evaluations per start {nf}, best {fs.max():.6f}")
    # both are stationary points of the same objective from the same starts.  Round 6: the search takes L-BFGS-B's unit first step (to the
    # corner the gradient points at) where the gradient is not small: per start the device ends at SciPy's value or above for >= 8 of the 10
    # starts (rounds 3-5: 4 required, 5 reached; six seeds x 10 starts on the device: 54 of 60, none below 8), the best of the starts is within
    # 1 % of SciPy's (8 % allowed until now), and the passes needed stay within twice SciPy's evaluations for its slowest start
    assert np.sum(fd >= fs - 1e-6 * np.abs(fs)) >= 8, (fd, fs)
    assert bf >= fs.max() - 1e-2 * abs(fs.max()), (bf, fs.max())
    assert ev <= 2 * max(nf), (ev, nf)
    # the host restatement of the same search on the ORACLE's objective ends where the device ends
    from bohip.acquisition import _batched_lbfgs_ascent

    def fg(Z):
        s, g = orc.score_grad(X, ll, 0.0, 0.0, L, alpha, "UCB", [bt], np.ascontiguousarray(Z.T))
        return s, np.asfortranarray(g.T)

    fh, Xh = _batched_lbfgs_ascent(fg, starts, lb, ub, 2000)
    np.testing.assert_allclose(fd, fh, rtol=1e-9, atol=1e-9)


def test_default_usage_ei_acquire_max_vs_scipy_on_the_oracle(bohip, orc, headline):
    """bench.py's `default_usage_ei`: the reference's DEFAULT acquisition (ExpectedImprovement, src/BayesianOptimization.jl:265) through its
    default search (:LD_LBFGS, 10 restarts, maxeval 2000, src/acquisition.jl:4-6) on the headline model, at two incumbents.
    tau = max y (what `boptimize!` uses): EI at a Latin-hypercube start is 1e-20 .. 1e-100 with a gradient to match; SciPy's L-BFGS-B on the
    oracle stops at the first evaluation of every start (its gtol / ftol are absolute at this scale).  The device must do no worse per start
    and may climb (NLopt, called with no tolerance, would): it gets a pass budget, not SciPy's count.
    tau = median y: EI of order 1e-2 .. 1, something to climb from every start: per start no worse than SciPy from the same start (one
    exception), same best of the starts.  No pass bound: a start on the exponential flank of EI (value 1e-8, no curvature pair is ever
    accepted) improves by a constant FACTOR per pass up to maxeval, as NLopt without tolerances would; SciPy's ftol gives it up."""
    from scipy.optimize import minimize

    from bench import lhs

    m, X, y, ll, L, alpha = headline
    lb, ub = np.zeros(8), np.ones(8)
    for seed in (7, 8):
        starts = np.asfortranarray(lhs(10, seed=seed).T)
        for tau, flat in ((float(y.max()), True), (float(np.median(y)), False)):
            fd, Xd, bf, bi, bx, ev = m.ascend("EI", [tau], lb, ub, starts, 2000)
            f0, g0 = orc.score_grad(X, ll, 0.0, 0.0, L, alpha, "EI", [tau], np.ascontiguousarray(starts.T))
            nf, fs = [], []
            for r in range(10):
                cnt = [0]

                def negfg(x):
                    cnt[0] += 1
                    s, g = orc.score_grad(X, ll, 0.0, 0.0, L, alpha, "EI", [tau], x[None, :].copy())
                    return -float(s[0]), -g[0]

                res = minimize(negfg, starts[:, r], jac=True, method="L-BFGS-B", bounds=[(0, 1)] * 8, options=dict(maxiter=2000, ftol=1e-10, gtol=1e-10))
                nf.append(cnt[0]); fs.append(-res.fun)
            fs = np.array(fs)
            print(f"default_usage_ei seed {seed} tau {'max y' if flat else 'median y'}: device {ev} passes, best {bf:.3e}; SciPy evaluations per start {nf}, best {fs.max():.3e}")
            assert np.all(fd >= f0 * (1 - 1e-9)) and bf == fd.max() and bi == int(np.argmax(fd))
            sc_o = orc.score(X, ll, 0.0, 0.0, L, alpha, "EI", [tau], np.ascontiguousarray(Xd.T))[0]
            np.testing.assert_allclose(fd, sc_o, rtol=1e-6, atol=1e-300)
            if flat:
                assert max(nf) == 1 and ev <= 400, (ev, nf)
                assert np.all(fd >= fs * (1 - 1e-6)) and bf >= fs.max() * (1 - 1e-6), (fd, fs)
            else:
                assert np.sum(fd >= fs - 1e-6 * np.abs(fs).max()) >= 9, (fd, fs)
                assert bf >= fs.max() * (1 - 1e-6), (bf, fs.max())


def test_stress_variant_all_4096_candidates(bohip, orc):
    """SURVEY.md 8-D1 stress variant (l_sigma = 5, logNoise = 0, every position observed 5 times, N = 3000, kappa(cK) ~ 5e10 -- the
    case that can break W = L^-1): ALL 4096 candidates against the oracle's substitution on its own factor, arg-max over the full set."""
    d, P, reps, R = 8, 600, 5, 4096
    rng = np.random.default_rng(77)
    Xp = rng.random((P, d))
    X = np.repeat(Xp, reps, axis=0)
    N = len(X)
    y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    Xs = rng.random((R, d))
    Xs[:64] = Xp[:64] + 1e-3 * rng.standard_normal((64, d))
    ll = np.full(d, math.log(0.5))
    lsig, lnoise, beta = 5.0, 0.0, 0.0
    s2f = math.exp(2 * lsig)
    m = make_model(bohip, X, y, ll, lsig, lnoise, beta)
    L, alpha = orc.fit(X, y, ll, lsig, lnoise, beta)
    mu, var = m.predict_f(Xs.T)
    mu_o, var_o = orc.predict(X, ll, lsig, beta, L, alpha, Xs, nthreads=NTH)
    assert np.all(var >= 0) and np.all(var <= s2f * (1 + 1e-12))
    assert np.all(np.abs(var - var_o) <= var_tol(var_o, N, s2f)), np.abs(var - var_o).max()
    assert np.all(np.abs(mu - mu_o) <= 1e-6 * np.abs(mu_o) + mu_floor(alpha, s2f))
    vt0 = var_tol(var_o, N, s2f, rel=0.0)
    for acq, p in [("EI", [float(y.max())]), ("UCB", [10.152008469453344])]:
        sc, bv, bi = m.score(acq, p, Xs.T)
        sc_o, bv_o, bi_o = orc.score(X, ll, lsig, beta, L, alpha, acq, p, Xs, nthreads=NTH)
        floor = mu_floor(alpha, s2f) + (1 + abs(p[0]) if acq == "UCB" else 1.0) * vt0 / (2 * np.sqrt(np.maximum(var_o, vt0)))
        assert np.all(np.abs(sc - sc_o) <= 1e-6 * np.abs(sc_o) + floor), (acq, np.abs(sc - sc_o).max())
        assert bi == bi_o == int(np.argmax(sc_o)), (acq, bi, bi_o, np.sort(sc_o)[-3:])
        assert abs(bv - bv_o) <= 1e-6 * abs(bv_o) + floor[bi_o]


# ---- one seed each of the randomised sweeps (tools/fuzz_parity.py, tools/fuzz_large.py) ---------------------------------------------
def _fuzz_case(bohip, orc, rng, N, d, kern, pieces, Rs, nsub=None):
    nl = 1 if kern == "SEIso" else d
    ll = rng.normal(0.3 if d > 16 else -0.6, 0.25, nl)
    lsig, lnoise, beta = float(rng.normal(0.2, 0.3)), float(rng.uniform(-2.5, -0.5)), float(rng.normal(0, 0.3))
    X, y, _ = synth(N, d, 4, seed=int(rng.integers(1 << 30)))
    llp = ll if nl > 1 else float(ll[0])
    K = {"SEArd": bohip.SEArd, "SEIso": bohip.SEIso, "Mat52Ard": bohip.Mat52Ard}[kern]
    m = bohip.ElasticGPE(d, mean=bohip.MeanConst(beta), kernel=K(llp, lsig), logNoise=lnoise, capacity=max(N // 2, 1))
    pos = 0
    while pos < N:                                                         # capacity growth, incremental appends (p <= 32), refits
        p = int(min(N - pos, rng.choice(pieces)))
        m.append_(X[pos:pos + p].T, y[pos:pos + p]); pos += p
    L, alpha = orc.fit(X, y, llp, lsig, lnoise, beta, kern=kern)
    s2f = math.exp(2 * lsig)
    tag = dict(N=N, d=d, kern=kern)
    np.testing.assert_allclose(m.factor(), L, rtol=1e-8, atol=1e-10 * math.sqrt(s2f), err_msg=str(tag))
    fl = mu_floor(alpha, s2f)
    for R in Rs:
        R = int(R)
        Xs = rng.random((R, d))
        sub = np.arange(R) if nsub is None or R <= nsub else rng.choice(R, nsub, replace=False)
        acq, p = [("EI", [float(y.max())]), ("UCB", [2.0]), ("PI", [float(y.max())]), ("MI", [1.0, 0.3]), ("MaxMean", [])][int(rng.integers(5))]
        sc_o, g_o = orc.score_grad(X, llp, lsig, beta, L, alpha, acq, p, Xs[sub], kern=kern)
        mu_o, var_o = orc.predict(X, llp, lsig, beta, L, alpha, Xs[sub], kern=kern, nthreads=8)
        sc, g = m.score_grad(acq, p, Xs.T)
        sc2, bv, bi = m.score(acq, p, Xs.T)
        mu, var = m.predict_f(Xs.T)
        vt = var_tol(var_o, N, s2f)
        amp = max(1.0, abs(p[0])) if acq in ("UCB", "MI") else 1.0
        vt0 = var_tol(var_o, N, s2f, rel=0)
        sfl = fl + (amp * np.sqrt(vt0) if acq in ("UCB", "MI") else vt0 + 1e-15)
        t = str(dict(tag, R=R, acq=acq))
        assert np.all(np.abs(mu[sub] - mu_o) <= 1e-6 * np.abs(mu_o) + fl), t
        assert np.all(np.abs(var[sub] - var_o) <= vt), t
        assert np.all(np.abs(sc[sub] - sc_o) <= 1e-6 * np.abs(sc_o) + sfl), t
        np.testing.assert_array_equal(sc, sc2, err_msg=t)
        if np.isfinite(sc2).any():
            assert bi == int(np.argmax(np.where(np.isnan(sc2), -np.inf, sc2))) and sc2[bi] == bv, t
        good = var_o > 1e3 * vt
        if good.any():
            np.testing.assert_allclose(g.T[sub][good], g_o[good], rtol=1e-5, atol=1e-7 * (np.abs(g_o[good]).max() + 1e-300), err_msg=t)


def test_fuzz_parity_one_seed(bohip, orc):
    """tools/fuzz_parity.py, SEED = 5, one trial per size: model sizes and batch sizes around every tile / chunk / path boundary,
    random kernel / dimension / hyper-parameters, models grown in random pieces; everything against the oracle."""
    rng = np.random.default_rng(5)
    Rs_all = [1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 200, 257, 300, 513, 700]
    for N in [1, 2, 63, 127, 128, 129, 255, 256, 257, 383, 384, 385, 640, 1000, 1153, 1600]:
        d = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17]))
        kern = str(rng.choice(["SEArd", "SEIso", "Mat52Ard"]))
        _fuzz_case(bohip, orc, rng, N, d, kern, [1, 3, 32, 33, 200], rng.choice(Rs_all, size=3, replace=False))


def test_fuzz_large_one_seed(bohip, orc):
    """tools/fuzz_large.py, SEED = 5: N up to 4000 grown across tile boundaries, d up to 64, R across the candidate-chunk boundary and
    the row-wise / split-K / whole-K thresholds, against the oracle on a 300-candidate subset."""
    rng = np.random.default_rng(5)
    for N, d, R in [(2049, 33, 8193), (3000, 64, 97), (4000, 8, 20000)]:
        kern = str(rng.choice(["SEArd", "Mat52Ard"]))
        _fuzz_case(bohip, orc, rng, N, d, kern, [1, 31, 32, 500, 1500], [R], nsub=300)
