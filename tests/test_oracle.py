"""CPU tests of the oracle: the reference's own known-answer tests restated, the committed golden
vectors, and mutual agreement of the three independent restatements (C / NumPy-LAPACK / mpmath)."""
import math

import numpy as np
import pytest

from conftest import load_golden, synth, var_tol
from oracle.oracle import (NumpyGP, argmax_first, brochu_beta, latin_hypercube, mp_acq, mp_predict, np_acq)


# ---- reference test/acquisition.jl:11-12: MaxMean on a 1-observation GP peaks at the observation ----
def test_ref_known_answer_maxmean_single_observation(orc):
    X = np.array([[1.0]]); y = np.array([2.0])
    L, alpha = orc.fit(X, y, [1.0], 0.0, -2.0, 0.0)                  # GPE([1.0],[2.0],MeanZero(),SEIso(1.0,0.0))
    rng = np.random.default_rng(0)
    starts = latin_hypercube([-5.0], [5.0], 10, rng)                 # acquire_max(opt, [-5.], [5.], 10)
    grid = np.concatenate([starts, np.linspace(-5, 5, 2001).reshape(-1, 1)])
    sc, bv, bi = orc.score(X, [1.0], 0.0, 0.0, L, alpha, "MaxMean", [], grid)
    assert grid[bi, 0] == pytest.approx(1.0, abs=5e-3)
    # closed form: mu(x) = k(x,1) * 2 / (1 + sn2 + eps)
    sn2 = math.exp(-4.0) + np.finfo(float).eps
    ell2 = math.exp(2.0)
    for x, m in zip(grid[:50, 0], sc[:50]):
        assert m == pytest.approx(math.exp(-0.5 * (x - 1) ** 2 / ell2) * 2 / (1 + sn2), rel=1e-13)


# ---- reference test/acquisitionfunctions.jl:4-11: batched == single, bit-exact (all but Thompson) ----
@pytest.mark.parametrize("acq,params", [("PI", [0.4]), ("EI", [0.4]), ("UCB", [brochu_beta(3, 4)]), ("MI", [1.0, 0.0])])
def test_ref_batch_equals_single_bitexact(orc, acq, params):
    rng = np.random.default_rng(1)
    X = rng.random((4, 3)); y = rng.random(4)                        # GPE(rand(3,4), rand(4), MeanZero(), SEIso(0.,0.))
    L, alpha = orc.fit(X, y, [0.0], 0.0, -2.0, 0.0, kern="SEIso")
    x = rng.random((2, 3))
    batch, _, _ = orc.score(X, [0.0], 0.0, 0.0, L, alpha, acq, params, x, kern="SEIso")
    single, _, _ = orc.score(X, [0.0], 0.0, 0.0, L, alpha, acq, params, x[:1], kern="SEIso")
    assert len(batch) == 2
    assert batch[0] == single[0]


# ---- reference test/warmstart.jl:64 and src/acquisitionfunctions.jl:44-46: tau = max(maxy, tau) ----
def test_ref_tau_rule():
    y = np.array([-3.0, -1.5, -2.0])
    tau = -math.inf
    tau = max(float(y.max()), tau)
    assert tau == y.max()
    tau = max(-5.0, tau)  # monotone: never decreases
    assert tau == y.max()


def test_reference_ei_is_not_textbook_ei(orc):
    g = load_golden("acq_formulas")
    mu, s2, tau, ei = g["ref_vs_textbook"]
    assert orc.acq("EI", [tau], mu, s2) == ei
    assert ei == pytest.approx(0.30492, abs=1e-5)
    z = (mu - tau) / math.sqrt(s2)
    textbook = (mu - tau) * 0.5 * (1 + math.erf(z / math.sqrt(2))) + math.sqrt(s2) * math.exp(-z * z / 2) / math.sqrt(2 * math.pi)
    assert textbook == pytest.approx(0.70187, abs=1e-5)


def test_acq_formula_table_c_numpy_mpmath(orc):
    g = load_golden("acq_formulas")
    for m, s, tau, ei, pi, ucb, mi in g["table"]:
        assert orc.acq("EI", [tau], m, s) == ei
        assert orc.acq("PI", [tau], m, s) == pi
        assert orc.acq("UCB", [2.5], m, s) == ucb
        assert orc.acq("MI", [1.3, 0.4], m, s) == mi
        # independent NumPy restatement: same libm erf/exp -> expect (near) bit equality
        assert np_acq("EI", [tau], m, s) == pytest.approx(ei, rel=1e-15, abs=1e-300)
        assert np_acq("PI", [tau], m, s) == pytest.approx(pi, rel=1e-15, abs=1e-300)
        # 60-digit evaluation: 1 + erf(.) cancels in the left tail -> absolute floor eps*|D|
        assert abs(float(mp_acq("EI", [tau], m, s)) - ei) <= 4e-16 * max(1.0, abs(m - tau)) + 1e-15 * abs(ei)
    for D, n, b in g["brochu"]:
        assert orc.brochu_beta(int(D), int(n)) == b == brochu_beta(int(D), int(n))
    assert brochu_beta(8, 3000) == pytest.approx(10.152008469453344, rel=1e-15)  # SURVEY.md A6 probe


def test_sigma2_zero_branches(orc):
    assert orc.acq("EI", [0.5], 0.7, 0.0) == pytest.approx(0.2, rel=1e-15)
    assert orc.acq("EI", [0.5], 0.3, 0.0) == 0.0
    assert orc.acq("PI", [0.5], 0.7, 0.0) == 1.0
    assert orc.acq("PI", [0.5], 0.5, 0.0) == 0.0


@pytest.mark.parametrize("name", ["n1_seiso_maxmean", "n2_seard", "readme_d2_rep5", "branin_shaped", "n256_d8_r512", "ties_n256"])
def test_golden_vectors_reproduce(orc, name):
    g = load_golden(name)
    X, y, Xs = g["X"], g["y"], g["Xs"]
    L, alpha = orc.fit(X, y, g["loglen"], float(g["logsig"]), float(g["lognoise"]), float(g["beta"]))
    mu, var = orc.predict(X, g["loglen"], float(g["logsig"]), float(g["beta"]), L, alpha, Xs)
    np.testing.assert_array_equal(mu, g["mu"])
    np.testing.assert_array_equal(var, g["var"])
    np.testing.assert_array_equal(np.diag(L), g["Ldiag"])
    for acq in ("EI", "PI", "UCB", "MI", "MaxMean"):
        if f"{acq}_score" not in g:
            continue
        sc, bv, bi = orc.score(X, g["loglen"], float(g["logsig"]), float(g["beta"]), L, alpha, acq, g[f"{acq}_params"], Xs)
        np.testing.assert_array_equal(sc, g[f"{acq}_score"])
        assert bi == g[f"{acq}_best_idx"][0] and bv == g[f"{acq}_best"][0]
        assert (bv, bi) == argmax_first(sc)


def test_ties_first_maximum_wins(orc):
    g = load_golden("ties_n256")
    sc = g["EI_score"]
    bi = int(g["EI_best_idx"][0])
    dup = np.flatnonzero(sc == sc[bi])
    assert len(dup) >= 2 and bi == dup.min()                          # duplicated column, smallest index kept


def test_c_vs_numpy_lapack_vs_mpmath(orc):
    X, y, Xs = synth(120, 5, 40, seed=3)
    ll = np.linspace(-0.8, 0.2, 5)
    L, alpha = orc.fit(X, y, ll, 0.4, -1.5, 0.25)
    mu, var = orc.predict(X, ll, 0.4, 0.25, L, alpha, Xs)
    ngp = NumpyGP(5, ll, 0.4, -1.5, 0.25).fit(X, y)
    mu_n, var_n = ngp.predict_f(Xs)
    s2f = math.exp(0.8)
    np.testing.assert_allclose(L, ngp.L, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(mu, mu_n, rtol=1e-10, atol=1e-11)
    assert np.all(np.abs(var - var_n) <= var_tol(var_n, 120, s2f, rel=1e-9))
    mus, vars_ = mp_predict(X[:40], y[:40], ll, 0.4, -1.5, 0.25, Xs[:5])
    L2, a2 = orc.fit(X[:40], y[:40], ll, 0.4, -1.5, 0.25)
    mu2, var2 = orc.predict(X[:40], ll, 0.4, 0.25, L2, a2, Xs[:5])
    for i in range(5):
        assert float(mus[i]) == pytest.approx(mu2[i], rel=1e-11, abs=1e-12)
        assert abs(float(vars_[i]) - var2[i]) <= var_tol(var2[i], 40, s2f, rel=1e-10)


def test_incremental_append_matches_full_factor(orc):
    X, y, _ = synth(90, 4, 1, seed=5)
    ll = np.full(4, -0.5)
    cK = orc.build_cK(X, ll, 0.0, -2.0)
    Lfull = orc.cholesky(cK)
    L60 = orc.cholesky(cK[:60, :60])
    Lapp = orc.cholesky_append(L60, cK[60:, :])
    np.testing.assert_allclose(Lapp, Lfull, rtol=1e-13, atol=1e-15)


def test_not_positive_definite_reports_pivot(orc):
    A = np.eye(5); A[3, 3] = -1.0
    with pytest.raises(np.linalg.LinAlgError, match="pivot 4"):
        orc.cholesky(A)


@pytest.mark.parametrize("kern", ["SEArd", "Mat52Ard"])
def test_gradient_matches_finite_difference(orc, kern):
    X, y, Xs = synth(60, 3, 6, seed=7)
    ll = np.array([-0.3, 0.1, -0.6])
    L, alpha = orc.fit(X, y, ll, 0.2, -2.0, 0.1, kern=kern)
    for acq, p in [("EI", [y.max()]), ("UCB", [2.0]), ("PI", [y.max()]), ("MI", [1.0, 0.3]), ("MaxMean", [])]:
        sc, grad = orc.score_grad(X, ll, 0.2, 0.1, L, alpha, acq, p, Xs, kern=kern)
        h = 1e-6
        for k in range(3):
            Xp, Xm = Xs.copy(), Xs.copy()
            Xp[:, k] += h; Xm[:, k] -= h
            fp, _, _ = orc.score(X, ll, 0.2, 0.1, L, alpha, acq, p, Xp, kern=kern)
            fm, _, _ = orc.score(X, ll, 0.2, 0.1, L, alpha, acq, p, Xm, kern=kern)
            np.testing.assert_allclose(grad[:, k], (fp - fm) / (2 * h), rtol=2e-5, atol=1e-8)


def test_latin_hypercube_stratification():
    rng = np.random.default_rng(11)
    lb, ub = np.array([-5.0, 0.0, 2.0]), np.array([10.0, 15.0, 2.0])
    S = latin_hypercube(lb, ub, 37, rng)
    assert S.shape == (37, 3)
    for k in range(2):
        strata = np.floor((S[:, k] - lb[k]) / ((ub[k] - lb[k]) / 37)).astype(int)
        assert sorted(strata) == list(range(37))                     # exactly one point per stratum (src/utils.jl:113-118)
    assert np.all(S[:, 2] == 2.0)
    with pytest.raises(ValueError):
        latin_hypercube([1.0], [0.0], 3, rng)


def test_thompson_oracle_argmax(orc):
    rng = np.random.default_rng(2)
    mu, var, z = rng.standard_normal(50), rng.random(50), rng.standard_normal((7, 50))
    bv, bi = orc.thompson(mu, var, z)
    f = mu + np.sqrt(var) * z
    np.testing.assert_array_equal(bi, f.argmax(1))
    np.testing.assert_array_equal(bv, f.max(1))


def test_multithreaded_oracle_is_bit_identical(orc):
    X, y, Xs = synth(200, 6, 64, seed=9)
    ll = np.full(6, -0.7)
    L, alpha = orc.fit(X, y, ll, 0.0, -2.0, 0.0)
    a = orc.score(X, ll, 0.0, 0.0, L, alpha, "EI", [y.max()], Xs, nthreads=1)
    b = orc.score(X, ll, 0.0, 0.0, L, alpha, "EI", [y.max()], Xs, nthreads=4)
    np.testing.assert_array_equal(a[0], b[0])
    assert a[1:] == b[1:]


@pytest.mark.parametrize("kern", ["SEArd", "SEIso", "Mat52Ard"])
def test_oracle_mll_gradient_matches_central_differences(kern):
    """oracle_mll_grad (the checker for bohip_gp_mll_grad) against central differences of its own mll."""
    from oracle.oracle import COracle

    o = COracle()
    rng = np.random.default_rng(3)
    N, d = 60, 3
    X = rng.random((N, d))
    y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    nl = 1 if kern == "SEIso" else d
    th = np.concatenate([[-1.2, 0.2], rng.normal(-0.5, 0.3, nl), [0.3]])

    def f(t):
        return o.mll_grad(X, y, t[2:2 + nl] if nl > 1 else t[2], t[2 + nl], t[0], t[1], kern)

    m, g = f(th)
    fd = np.array([(f(th + 1e-6 * e)[0] - f(th - 1e-6 * e)[0]) / 2e-6 for e in np.eye(len(th))])
    np.testing.assert_allclose(g, fd, rtol=1e-6, atol=1e-6)
    # textbook mll from the oracle factor
    L, alpha = o.fit(X, y, th[2:2 + nl] if nl > 1 else th[2], th[2 + nl], th[0], th[1], kern)
    ref = -0.5 * (y - th[1]) @ alpha - np.log(np.diag(L)).sum() - 0.5 * N * np.log(2 * np.pi)
    assert m == pytest.approx(ref, rel=1e-12)


def test_oracle_full_covariance_consistent_with_variance_path():
    """oracle_predict_cov (checker of bohip_gp_predict_cov): diagonal == the per-candidate variance, symmetric, and
    equal to the NumPy restatement K** - K*' cK^-1 K*."""
    from oracle.oracle import COracle, np_cov

    o = COracle()
    X, y, Xs = synth(80, 3, 17, seed=8)
    ll = np.array([-0.3, -0.6, 0.1])
    L, alpha = o.fit(X, y, ll, 0.2, -1.0, 0.1)
    mu, cov = o.predict_cov(X, ll, 0.2, 0.1, L, alpha, Xs)
    mu1, var1 = o.predict(X, ll, 0.2, 0.1, L, alpha, Xs)
    np.testing.assert_array_equal(mu, mu1)
    np.testing.assert_allclose(np.diag(cov), var1, rtol=1e-12, atol=1e-13)
    np.testing.assert_array_equal(cov, cov.T)
    cK = L @ L.T
    Ks = np_cov("SEArd", X, Xs, ll, 0.2)
    ref = np_cov("SEArd", Xs, Xs, ll, 0.2) - Ks.T @ np.linalg.solve(cK, Ks)
    np.testing.assert_allclose(cov, ref, rtol=1e-8, atol=1e-10)
