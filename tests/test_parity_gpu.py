"""GPU parity tests: the HIP path (through the C ABI in libbohip.so) against the CPU oracle.

Tolerances (north_star: mu / sigma^2 / EI within 1e-6 relative, arg-max indices bit-exact):
  mu, scores : |d| <= 1e-6 |ref| + floor, floor = eps-level multiple of the cancelling terms
  sigma^2    : |d| <= 1e-6 |ref| + 64 N eps s_f^2   (see conftest.var_tol)
  arg-max    : exact index equality with the oracle's strict-'>' first-maximum rule
"""
import ctypes as C
import math
import os

import numpy as np
import pytest

from conftest import load_golden, synth, var_tol

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


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


def check_scores(sc, ref, floor):
    assert np.all(np.abs(sc - ref) <= 1e-6 * np.abs(ref) + floor), np.abs(sc - ref).max()


# ---- committed golden vectors ---------------------------------------------------------------------
@pytest.mark.parametrize("name", ["n1_seiso_maxmean", "n2_seard", "readme_d2_rep5", "branin_shaped", "n256_d8_r512", "ties_n256"])
def test_golden(bohip, name):
    g = load_golden(name)
    X, y, Xs = g["X"], g["y"], g["Xs"]
    N = len(y)
    s2f = math.exp(2 * float(g["logsig"]))
    m = make_model(bohip, X, y, g["loglen"], float(g["logsig"]), float(g["lognoise"]), float(g["beta"]))
    mu, var = m.predict_f(Xs.T)
    fl = mu_floor(g["alpha"], s2f)
    assert np.all(np.abs(mu - g["mu"]) <= 1e-6 * np.abs(g["mu"]) + fl)
    assert np.all(np.abs(var - g["var"]) <= var_tol(g["var"], N, s2f))
    np.testing.assert_allclose(np.diag(m.factor()), g["Ldiag"], rtol=1e-10)
    np.testing.assert_allclose(m.factor()[-1], g["Lrow_last"], rtol=1e-8, atol=1e-10 * math.sqrt(s2f))
    for acq in ("EI", "PI", "UCB", "MI", "MaxMean"):
        if f"{acq}_score" not in g:
            continue
        sc, bv, bi = m.score(acq, g[f"{acq}_params"], Xs.T)
        ref = g[f"{acq}_score"]
        sfloor = fl + 64 * N * EPS * s2f * (1 + (abs(g[f"{acq}_params"][0]) if acq in ("UCB",) else 0)) + 1e-15
        if acq == "UCB" or acq == "MI":  # sqrt(sigma^2) amplifies the sigma^2 floor near observations
            sfloor = sfloor + np.sqrt(var_tol(g["var"], N, s2f, rel=0)) * max(1.0, abs(g[f"{acq}_params"][0]))
        check_scores(sc, ref, sfloor)
        assert bi == g[f"{acq}_best_idx"][0], (acq, bi, g[f"{acq}_best_idx"][0])
        assert sc[bi] == bv


def test_ties_smallest_index_wins_on_device(bohip):
    g = load_golden("ties_n256")
    m = make_model(bohip, g["X"], g["y"], g["loglen"], float(g["logsig"]), float(g["lognoise"]), float(g["beta"]))
    sc, bv, bi = m.score("EI", g["EI_params"], g["Xs"].T)
    # duplicated columns are scored by identical instruction sequences -> bit-identical values
    np.testing.assert_array_equal(sc[:100], sc[101:201])
    assert bi == g["EI_best_idx"][0] and bi == int(np.flatnonzero(sc == sc.max()).min())


# ---- reference test/acquisitionfunctions.jl:8-11: batched == single, BIT-exact, on the device ------
@pytest.mark.parametrize("acq,params", [("PI", [0.4]), ("EI", [0.4]), ("UCB", [2.0]), ("MI", [1.0, 0.0]), ("MaxMean", [])])
def test_batch_equals_single_bitexact_on_device(bohip, acq, params):
    rng = np.random.default_rng(1)
    X = rng.random((4, 3)); y = rng.random(4)
    m = make_model(bohip, X, y, [0.0], 0.0, -2.0, 0.0, kern="SEIso")
    x = rng.random((3, 2))
    batch, _, _ = m.score(acq, params, x)
    single, _, _ = m.score(acq, params, x[:, :1])
    assert len(batch) == 2 and batch[0] == single[0]
    mid = rng.random((3, 20)); mid[:, 13] = x[:, 0]
    assert m.score(acq, params, mid)[0][13] == single[0]           # independent of position / batch size (same code path)
    # batches above 32 candidates run on the MFMA engine (different summation order): equal to rounding, and
    # again bit-identical among themselves wherever the column sits
    big = rng.random((3, 700)); big[:, 333] = x[:, 0]; big[:, 600] = x[:, 0]
    sb = m.score(acq, params, big)[0]
    assert sb[333] == sb[600]
    assert sb[333] == pytest.approx(single[0], rel=1e-12, abs=1e-14)


# ---- seeded synthetic cases against the oracle, every acquisition ------------------------------------
_ARGMAX_EXEMPT = []   # (N, d, acquisition) cases of test_seeded_vs_oracle that took the near-tie exemption: printed, and bounded below
@pytest.mark.parametrize("N,d,R,lsig,lnoise,beta", [
    (1, 1, 5, 0.0, -2.0, 0.0), (7, 2, 3, 0.5, -1.0, 0.3), (127, 3, 129, 0.0, -2.0, 0.0), (128, 4, 128, 0.0, -2.0, -1.0),
    (129, 5, 257, 0.3, -1.5, 0.0), (500, 2, 1000, 5.0, 0.0, 0.0), (1000, 8, 1500, 0.0, -2.0, 0.0), (700, 16, 300, 0.0, -2.0, 0.2)])   # = _SEEDED_SHAPES below
def test_seeded_vs_oracle(bohip, orc, N, d, R, lsig, lnoise, beta):
    X, y, Xs = synth(N, d, R, seed=N + d)
    ll = np.linspace(-0.9, -0.3, d)
    L, alpha = orc.fit(X, y, ll, lsig, lnoise, beta)
    m = make_model(bohip, X, y, ll, lsig, lnoise, beta)
    s2f = math.exp(2 * lsig)
    np.testing.assert_allclose(m.factor(), L, rtol=1e-9, atol=1e-11 * math.sqrt(s2f))
    np.testing.assert_allclose(m.alpha(), alpha, rtol=1e-6, atol=1e-9 * np.abs(alpha).max())
    mu_o, var_o = orc.predict(X, ll, lsig, beta, L, alpha, Xs, nthreads=8)
    mu, var = m.predict_f(Xs.T)
    fl = mu_floor(alpha, s2f)
    assert np.all(np.abs(mu - mu_o) <= 1e-6 * np.abs(mu_o) + fl)
    assert np.all(np.abs(var - var_o) <= var_tol(var_o, N, s2f))
    tau = float(y.max())
    for acq, p in [("EI", [tau]), ("PI", [tau]), ("UCB", [orc.brochu_beta(d, N)]), ("MI", [1.0, 0.3]), ("MaxMean", [])]:
        sc_o, bv_o, bi_o = orc.score(X, ll, lsig, beta, L, alpha, acq, p, Xs, nthreads=8)
        sc, bv, bi = m.score(acq, p, Xs.T)
        amp = max(1.0, abs(p[0])) if acq in ("UCB", "MI") else 1.0
        floor = fl + amp * np.sqrt(var_tol(var_o, N, s2f, rel=0)) if acq in ("UCB", "MI") else fl + var_tol(var_o, N, s2f, rel=0) + 1e-15
        check_scores(sc, sc_o, floor)
        # arg-max: exact unless the oracle's own top two are closer than the documented floor
        top2 = np.sort(sc_o)[-2:] if R > 1 else np.array([-np.inf, sc_o[0]])
        if top2[1] - top2[0] > 4 * np.max(floor):
            assert bi == bi_o, (acq, bi, bi_o)
        else:
            _ARGMAX_EXEMPT.append((N, d, acq))
        assert sc[bi] == bv
    print(f"test_seeded_vs_oracle: arg-max exemptions so far (oracle's top two closer than the floor): {len(_ARGMAX_EXEMPT)} {_ARGMAX_EXEMPT}")


_SEEDED_SHAPES = [(1, 1, 5, 0.0, -2.0, 0.0), (7, 2, 3, 0.5, -1.0, 0.3), (127, 3, 129, 0.0, -2.0, 0.0), (128, 4, 128, 0.0, -2.0, -1.0),
                  (129, 5, 257, 0.3, -1.5, 0.0), (500, 2, 1000, 5.0, 0.0, 0.0), (1000, 8, 1500, 0.0, -2.0, 0.0), (700, 16, 300, 0.0, -2.0, 0.2)]


def test_seeded_vs_oracle_argmax_exemptions_are_rare(orc):
    """Of the 40 (shape, acquisition) cases of test_seeded_vs_oracle, how many skip the exact arg-max assertion because the ORACLE's own top
    two scores are closer than the documented floor?  The criterion looks at oracle values only, so it is recomputed HERE from the same
    shapes and seeds (no module-level state: the bound holds under -k and xdist too) and bounded: the exemption must stay the exception."""
    exempt = []
    for N, d, R, lsig, lnoise, beta in _SEEDED_SHAPES:
        X, y, Xs = synth(N, d, R, seed=N + d)
        ll = np.linspace(-0.9, -0.3, d)
        L, alpha = orc.fit(X, y, ll, lsig, lnoise, beta)
        s2f = math.exp(2 * lsig)
        _, var_o = orc.predict(X, ll, lsig, beta, L, alpha, Xs, nthreads=8)
        fl = mu_floor(alpha, s2f)
        for acq, p in [("EI", [float(y.max())]), ("PI", [float(y.max())]), ("UCB", [orc.brochu_beta(d, N)]), ("MI", [1.0, 0.3]), ("MaxMean", [])]:
            sc_o, _, _ = orc.score(X, ll, lsig, beta, L, alpha, acq, p, Xs, nthreads=8)
            amp = max(1.0, abs(p[0])) if acq in ("UCB", "MI") else 1.0
            floor = fl + amp * np.sqrt(var_tol(var_o, N, s2f, rel=0)) if acq in ("UCB", "MI") else fl + var_tol(var_o, N, s2f, rel=0) + 1e-15
            top2 = np.sort(sc_o)[-2:] if R > 1 else np.array([-np.inf, sc_o[0]])
            if not top2[1] - top2[0] > 4 * np.max(floor):
                exempt.append((N, d, acq))
    print(f"arg-max exemptions: {len(exempt)} of 40: {exempt}")
    assert len(exempt) <= 4, exempt


def test_mat52ard_and_seiso_kernels(bohip, orc):
    X, y, Xs = synth(300, 6, 400, seed=21)
    for kern, ll in (("Mat52Ard", np.linspace(-0.5, 0.2, 6)), ("SEIso", np.array([-0.4]))):
        L, alpha = orc.fit(X, y, ll, 0.1, -2.0, 0.0, kern=kern)
        m = make_model(bohip, X, y, ll, 0.1, -2.0, 0.0, kern=kern)
        sc_o, _, bi_o = orc.score(X, ll, 0.1, 0.0, L, alpha, "EI", [y.max()], Xs, kern=kern, nthreads=8)
        sc, _, bi = m.score("EI", [y.max()], Xs.T)
        check_scores(sc, sc_o, mu_floor(alpha, math.exp(0.2)) + 1e-12)
        assert bi == bi_o


# ---- edge cases -------------------------------------------------------------------------------------
def test_empty_and_ragged_inputs(bohip):
    m = bohip.ElasticGPE(3)
    assert bohip.dims(m) == (3, 0) and bohip.maxy(m) == -math.inf
    with pytest.raises(bohip.BohipError):                    # predict with no observations is a state error, not a crash
        m.predict_f(np.zeros((3, 2)))
    m.append_(np.zeros((3, 0)), np.zeros(0))                 # p == 0 append is a no-op
    X, y, Xs = synth(50, 3, 10, seed=2)
    m.append_(X.T, y)
    sc, bv, bi = m.score("EI", [0.0], np.zeros((3, 0)))      # R == 0 -> (-Inf, -1), like acquire_max's initial state
    assert len(sc) == 0 and bv == -math.inf and bi == -1
    with pytest.raises(ValueError):
        m.predict_f(np.zeros((4, 2)))                        # wrong d
    with pytest.raises(ValueError):
        m.append_(np.zeros((3, 2)), np.zeros(3))             # x / y length mismatch


def test_c_abi_error_codes_on_device(bohip):
    """Every misuse returns a status code and a message; nothing aborts, the handle stays usable."""
    from bohip import _lib

    lib = _lib.load()
    h = C.c_void_p()
    assert lib.bohip_gp_create(2, 16, 0, 99, C.byref(h)) == _lib.E_ARG                 # device ordinal out of range
    assert lib.bohip_gp_create(2, 16, 0, 0, C.byref(h)) == _lib.OK
    X = (C.c_double * 4)(0.1, 0.2, 0.7, 0.9); y = (C.c_double * 2)(1.0, 2.0)
    mu = (C.c_double * 2)(); var = (C.c_double * 2)(); best = _lib.Best()
    assert lib.bohip_gp_predict(h, X, 2, mu, var) == _lib.E_STATE                      # no observations yet
    assert b"no observations" in lib.bohip_last_error()
    assert lib.bohip_gp_append(h, None, None, 2) == _lib.E_ARG
    assert lib.bohip_gp_append(h, X, y, 2) == _lib.OK
    assert lib.bohip_gp_score(h, 7, None, X, 2, None, C.byref(best)) == _lib.E_ARG      # unknown acquisition id
    assert lib.bohip_gp_score(h, _lib.ACQ["EI"], None, X, 2, None, C.byref(best)) == _lib.E_ARG   # EI needs tau
    assert lib.bohip_gp_predict(h, X, -1, mu, var) == _lib.E_ARG
    assert lib.bohip_gp_predict(h, X, 2, mu, var) == _lib.OK and var[0] >= 0
    n = C.c_int64(); d = C.c_int64(); m = C.c_double()
    assert lib.bohip_gp_dims(h, C.byref(d), C.byref(n)) == _lib.OK and (d.value, n.value) == (2, 2)
    assert lib.bohip_gp_maxy(h, C.byref(m)) == _lib.OK and m.value == 2.0
    v = C.c_int64()
    assert lib.bohip_gp_info(h, 42, C.byref(v)) == _lib.E_ARG
    assert lib.bohip_gp_info(h, _lib.INFO_CAPACITY, C.byref(v)) == _lib.OK and v.value >= 16
    lib.bohip_gp_destroy(h)
    lib.bohip_gp_destroy(None)                                                           # destroy(NULL) is a no-op


def test_nan_and_inf_never_win(bohip):
    X, y, Xs = synth(40, 2, 20, seed=4)
    m = make_model(bohip, X, y, [-0.5, -0.5], 0.0, -2.0, 0.0)
    Xs[7] = np.nan
    sc, bv, bi = m.score("EI", [y.max()], Xs.T)
    assert math.isnan(sc[7]) and bi != 7 and bi == int(np.nanargmax(sc))
    sc, bv, bi = m.score("EI", [y.max()], np.full((2, 3), np.nan))
    assert bi == -1 and bv == -math.inf                      # `f > maxf` is false for NaN (src/acquisition.jl:62)


def test_not_positive_definite_is_reported_not_fatal(bohip):
    X = np.zeros((40, 2)); y = np.arange(40.0)               # forty identical points, almost no noise
    m = bohip.ElasticGPE(2, kernel=bohip.SEArd([0.0, 0.0], 10.0), logNoise=-30.0)
    with pytest.raises(bohip.NotPositiveDefinite):
        m.append_(X.T, y)
    assert m.info(0) >= 2                                    # failing pivot index is retrievable
    m.set_params_(logNoise=-2.0)                             # handle stays usable
    m.fit_()
    assert np.all(np.isfinite(m.predict_f(np.ones((2, 1)))[0]))


def test_capacity_growth_and_incremental_append(bohip, orc):
    X, y, Xs = synth(400, 4, 64, seed=8)
    ll = np.full(4, -0.6)
    m = bohip.ElasticGPE(4, kernel=bohip.SEArd(ll, 0.0), logNoise=-2.0, capacity=100)
    m.append_(X[:90].T, y[:90])
    for lo, hi in [(90, 91), (91, 96), (96, 130), (130, 255), (255, 256), (256, 257), (257, 400)]:
        m.append_(X[lo:hi].T, y[lo:hi])                      # crosses tile (128) and capacity (100, 200, 400) boundaries
        assert bohip.dims(m) == (4, hi)
    L, alpha = orc.fit(X, y, ll, 0.0, -2.0, 0.0)
    np.testing.assert_allclose(m.factor(), L, rtol=1e-9, atol=1e-12)
    sc_o, _, bi_o = orc.score(X, ll, 0.0, 0.0, L, alpha, "EI", [y.max()], Xs)
    sc, _, bi = m.score("EI", [y.max()], Xs.T)
    check_scores(sc, sc_o, mu_floor(alpha, 1.0) + 1e-13)
    assert bi == bi_o
    assert m.info(1) >= 400
    np.testing.assert_array_equal(m.x, X.T)
    np.testing.assert_array_equal(m.y, y)


def test_incremental_append_is_used_and_matches_refit(bohip, orc):
    """A2': single / few-point appends extend the factor on the device (no refit), across a tile boundary
    (alpha row moves from a padding row into a new tile) and with duplicated columns (repetitions > 1)."""
    X, y, Xs = synth(140, 3, 50, seed=14)
    ll = np.array([-0.4, -0.7, -0.2])
    m = bohip.ElasticGPE(3, kernel=bohip.SEArd(ll, 0.3), logNoise=-1.5, mean=bohip.MeanConst(0.2), capacity=400)
    m.append_(X[:120].T, y[:120])
    refits0 = m.info(2)
    n = 120
    for p_ in (1, 1, 5, 1, 2, 1, 1, 8):                       # 120 -> 140, crossing 127/128
        m.append_(X[n:n + p_].T, y[n:n + p_]); n += p_
    assert n == 140 and m.info(2) == refits0 and m.info(3) == 8
    L, alpha = orc.fit(X, y, ll, 0.3, -1.5, 0.2)
    np.testing.assert_allclose(m.factor(), L, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(m.alpha(), alpha, rtol=1e-7, atol=1e-10 * np.abs(alpha).max())
    mu_o, var_o = orc.predict(X, ll, 0.3, 0.2, L, alpha, Xs)
    mu, var = m.predict_f(Xs.T)
    np.testing.assert_allclose(mu, mu_o, rtol=1e-6, atol=mu_floor(alpha, math.exp(0.6)))
    assert np.all(np.abs(var - var_o) <= var_tol(var_o, 140, math.exp(0.6)))
    # README-style repetitions: the same point appended 5 times (rank-deficient up to the noise term)
    xr = np.repeat(Xs[:1].T, 5, axis=1); yr = np.linspace(0.1, 0.5, 5)
    m.append_(xr, yr)
    X2 = np.concatenate([X, xr.T]); y2 = np.concatenate([y, yr])
    L2, a2 = orc.fit(X2, y2, ll, 0.3, -1.5, 0.2)
    np.testing.assert_allclose(m.factor(), L2, rtol=1e-8, atol=1e-11)
    m2 = make_model(bohip, X2, y2, ll, 0.3, -1.5, 0.2)        # full refit of the same data
    sc_i, _, bi_i = m.score("EI", [y2.max()], Xs.T)
    sc_f, _, bi_f = m2.score("EI", [y2.max()], Xs.T)
    np.testing.assert_allclose(sc_i, sc_f, rtol=1e-7, atol=1e-12)
    assert bi_i == bi_f


def test_alpha_after_a_long_run_of_appends(bohip, orc):
    """A2' / A3: since round 5 an append updates alpha incrementally (alpha_new = [alpha_old + W21' u2; W22' u2], O(N p)) instead of
    recomputing W'(W r).  600 observations appended one by one, then in blocks of 3 and 32, to a model of 300 -- what a BO loop does
    between two hyper-parameter updates -- without a refit: alpha, the factor and the posterior still match the oracle's from-scratch
    fit, and a full refit of the same handle moves alpha by rounding only."""
    X, y, Xs = synth(1100, 5, 80, seed=77)
    ll = np.linspace(-0.7, -0.3, 5)
    m = bohip.ElasticGPE(5, kernel=bohip.SEArd(ll, 0.2), logNoise=-1.8, mean=bohip.MeanConst(0.1), capacity=1200)
    m.append_(X[:300].T, y[:300])
    refits0 = m.info(2)
    n = 300
    for p_ in [1] * 600 + [3] * 24 + [32] * 4:
        m.append_(X[n:n + p_].T, y[n:n + p_]); n += p_
    assert n == 1100 and m.info(2) == refits0
    L, alpha = orc.fit(X, y, ll, 0.2, -1.8, 0.1)
    np.testing.assert_allclose(m.factor(), L, rtol=1e-8, atol=1e-11)
    a_inc = m.alpha().copy()
    np.testing.assert_allclose(a_inc, alpha, rtol=1e-7, atol=1e-10 * np.abs(alpha).max())
    mu_o, var_o = orc.predict(X, ll, 0.2, 0.1, L, alpha, Xs)
    mu, var = m.predict_f(Xs.T)
    np.testing.assert_allclose(mu, mu_o, rtol=1e-6, atol=mu_floor(alpha, math.exp(0.4)))
    assert np.all(np.abs(var - var_o) <= var_tol(var_o, 1100, math.exp(0.4)))
    m.fit_()                                                  # the same data factored from scratch on the device
    np.testing.assert_allclose(a_inc, m.alpha(), rtol=1e-8, atol=1e-11 * np.abs(alpha).max())


@pytest.mark.parametrize("kern,N,d,R", [("SEArd", 200, 3, 40), ("SEArd", 1000, 8, 300), ("Mat52Ard", 300, 6, 130), ("SEIso", 130, 2, 5)])
def test_score_grad_vs_oracle(bohip, orc, kern, N, d, R):
    """A8: analytic d(score)/dx through the reference's formulas, against the oracle's analytic gradient."""
    X, y, Xs = synth(N, d, R, seed=31 + d)
    ll = np.array([-0.5]) if kern == "SEIso" else np.linspace(-0.8, -0.2, d)
    L, alpha = orc.fit(X, y, ll, 0.1, -2.0, 0.05, kern=kern)
    m = make_model(bohip, X, y, ll, 0.1, -2.0, 0.05, kern=kern)
    tau = float(y.max())
    for acq, p in [("EI", [tau]), ("UCB", [2.5]), ("PI", [tau]), ("MI", [1.0, 0.3]), ("MaxMean", [])]:
        sc_o, g_o = orc.score_grad(X, ll, 0.1, 0.05, L, alpha, acq, p, Xs, kern=kern)
        sc, g = m.score_grad(acq, p, Xs.T)
        assert g.shape == (d, R)
        np.testing.assert_allclose(sc, sc_o, rtol=1e-6, atol=mu_floor(alpha, math.exp(0.2)) + 1e-12)
        scale = np.abs(g_o).max()
        np.testing.assert_allclose(g.T, g_o, rtol=1e-6, atol=1e-9 * scale + 1e-12)
        # value path and gradient path agree on the scores bit-for-bit
        np.testing.assert_array_equal(sc, m.score(acq, p, Xs.T)[0])


def test_hyperparameter_change_refits(bohip, orc):
    X, y, Xs = synth(150, 3, 30, seed=6)
    m = make_model(bohip, X, y, np.zeros(3), 0.0, -2.0, 0.0)
    before = m.predict_f(Xs.T)[0]
    m.set_params_(ll=np.array([-0.7, -0.2, 0.1]), lsigma=0.4, logNoise=-1.0, beta=0.5)
    L, alpha = orc.fit(X, y, [-0.7, -0.2, 0.1], 0.4, -1.0, 0.5)
    mu_o, var_o = orc.predict(X, [-0.7, -0.2, 0.1], 0.4, 0.5, L, alpha, Xs)
    mu, var = m.predict_f(Xs.T)
    assert not np.allclose(mu, before)
    np.testing.assert_allclose(mu, mu_o, rtol=1e-6, atol=mu_floor(alpha, math.exp(0.8)))
    assert np.all(np.abs(var - var_o) <= var_tol(var_o, 150, math.exp(0.8)))
    # mll against the textbook expression evaluated with the oracle factor
    ref = -0.5 * (y - 0.5) @ alpha - np.log(np.diag(L)).sum() - 0.5 * len(y) * math.log(2 * math.pi)
    assert m.mll() == pytest.approx(ref, rel=1e-9)


@pytest.mark.parametrize("kern,N,d", [("SEArd", 300, 5), ("SEIso", 130, 3), ("Mat52Ard", 257, 4), ("SEArd", 40, 1)])
def test_mll_gradient_vs_oracle(bohip, orc, kern, N, d):
    """bohip_gp_mll_grad (cK^-1 = W'W on the MFMA engine + one reduction pass) against the oracle's explicit-inverse
    gradient, in the reference's parameter order [logNoise; mean; kernel]."""
    X, y, _ = synth(N, d, 4, seed=21)
    rng = np.random.default_rng(5)
    nl = 1 if kern == "SEIso" else d
    ll = rng.normal(-0.6, 0.2, nl)
    lsig, lnoise, beta = 0.3, -1.5, 0.2
    m = make_model(bohip, X, y, ll if nl > 1 else float(ll[0]), lsig, lnoise, beta, kern=kern)
    mll_o, g_o = orc.mll_grad(X, y, ll if nl > 1 else float(ll[0]), lsig, lnoise, beta, kern=kern)
    mll, dn, dm, dk = m.mll_grad()
    g = np.concatenate([[dn, dm], dk])
    assert mll == pytest.approx(mll_o, rel=1e-9)
    # each entry is a sum of ~N^2/2 signed terms of magnitude up to |G| |dK|: absolute floor relative to the largest entry
    np.testing.assert_allclose(g, g_o, rtol=1e-6, atol=1e-8 * np.abs(g_o).max())
    assert mll == m.mll()


def test_mll_gradient_full_size_matches_central_differences(bohip):
    """N = 3000 (BASELINE configs[1]): the oracle's O(N^3) explicit inverse is too slow, so the analytic gradient is
    checked against central differences of the device mll itself."""
    X, y, _ = synth(3000, 8, 4, seed=0)
    ll = np.full(8, math.log(0.5))
    m = make_model(bohip, X, y, ll, 0.0, -2.0, 0.0)
    mll, dn, dm, dk = m.mll_grad()
    h = 1e-5

    def at(**kw):
        base = dict(ll=ll.copy(), lsigma=0.0, logNoise=-2.0, beta=0.0)
        base.update(kw)
        m.set_params_(**base)
        return m.mll()

    fd_n = (at(logNoise=-2.0 + h) - at(logNoise=-2.0 - h)) / (2 * h)
    fd_m = (at(beta=h) - at(beta=-h)) / (2 * h)
    fd_s = (at(lsigma=h) - at(lsigma=-h)) / (2 * h)
    e0 = np.zeros(8); e0[0] = h
    fd_l0 = (at(ll=ll + e0) - at(ll=ll - e0)) / (2 * h)
    scale = max(abs(dn), abs(dm), np.abs(dk).max())
    for a, b in [(dn, fd_n), (dm, fd_m), (dk[-1], fd_s), (dk[0], fd_l0)]:
        assert abs(a - b) <= 1e-5 * scale, (a, b)


@pytest.mark.parametrize("kern,N,d,R", [("SEArd", 300, 4, 70), ("Mat52Ard", 200, 3, 5), ("SEIso", 129, 2, 257)])
def test_full_posterior_covariance_vs_oracle(bohip, orc, kern, N, d, R):
    """bohip_gp_predict_cov (K** - V'V, V'V on the MFMA engine) against the oracle; the diagonal must agree with the
    variance path wherever that one is not clamped, and the matrix must be exactly symmetric."""
    X, y, Xs = synth(N, d, R, seed=31)
    nl = 1 if kern == "SEIso" else d
    ll = np.full(nl, -0.4) if nl > 1 else -0.4
    lsig, lnoise, beta = 0.2, -1.0, 0.1
    m = make_model(bohip, X, y, ll, lsig, lnoise, beta, kern=kern)
    L, alpha = orc.fit(X, y, ll, lsig, lnoise, beta, kern=kern)
    mu_o, cov_o = orc.predict_cov(X, ll, lsig, beta, L, alpha, Xs, kern=kern)
    mu, cov = m.predict_cov(Xs.T)
    s2f = math.exp(2 * lsig)
    np.testing.assert_allclose(mu, mu_o, rtol=1e-6, atol=mu_floor(alpha, s2f))
    assert np.all(np.abs(cov - cov_o) <= var_tol(cov_o, N, s2f)), np.abs(cov - cov_o).max()
    np.testing.assert_array_equal(cov, cov.T)
    _, var = m.predict_f(Xs.T)
    pos = np.diag(cov) > 0
    assert np.all(np.abs(np.diag(cov)[pos] - var[pos]) <= var_tol(var[pos], N, s2f))


def test_joint_draw_has_posterior_moments(bohip):
    """myrand(model, X::Matrix) = one joint draw (reference src/models/gp.jl:7): length pinned by
    test/acquisitionfunctions.jl:8; here also the empirical mean and covariance of many draws."""
    X, y, Xs = synth(120, 2, 6, seed=4)
    m = make_model(bohip, X, y, np.array([-0.5, -0.5]), 0.0, -1.0, 0.0)
    rng = np.random.default_rng(0)
    draws = np.stack([bohip.myrand(m, Xs.T, rng) for _ in range(4000)])
    assert draws.shape == (4000, 6)
    mu, cov = m.predict_cov(Xs.T)
    sd = np.sqrt(np.diag(cov))
    assert np.all(np.abs(draws.mean(0) - mu) <= 5 * sd / math.sqrt(4000))
    emp = np.cov(draws.T)
    assert np.all(np.abs(emp - cov) <= 0.15 * np.outer(sd, sd) + 1e-12)
    assert isinstance(bohip.myrand(m, Xs[0], rng), float)


def test_thompson_draws_match_oracle(bohip, orc):
    from bohip import _lib

    lib = _lib.load()
    X, y, Xs = synth(200, 4, 500, seed=10)
    ll = np.full(4, -0.5)
    m = make_model(bohip, X, y, ll, 0.0, -2.0, 0.0)
    mu, var = m.predict_f(Xs.T)
    S, seed, j0 = 16, 42, 1000
    z = np.array([[lib.bohip_thompson_normal(seed, s, j0 + j) for j in range(500)] for s in range(S)])
    bv_o, bi_o = orc.thompson(mu, var, z)
    bv, bi = m.thompson(Xs.T, S, seed=seed, j0=j0)
    np.testing.assert_array_equal(bi, bi_o)                  # host/device z agree to an ulp; winners are well separated
    np.testing.assert_allclose(bv, bv_o, rtol=1e-12)
    # sharding invariance: the same global columns split in two give the same winners
    bv1, bi1 = m.thompson(Xs[:250].T, S, seed=seed, j0=j0)
    bv2, bi2 = m.thompson(Xs[250:].T, S, seed=seed, j0=j0 + 250)
    glob = np.where(bv1 >= bv2, bi1, bi2 + 250)
    np.testing.assert_array_equal(glob, bi)


@pytest.mark.parametrize("N,d,maxeval", [(200, 4, 600), (1000, 8, 2000), (60, 1, 90)])
def test_direct_l_in_one_library_call(bohip, orc, N, d, maxeval):
    """:GN_DIRECT_L as ONE library call (bohip_gp_direct_max; reference src/acquisition.jl:7-9, :20-38): the search must evaluate exactly
    the points its NumPy twin asks for when the twin scores through the same device entry points -- same best value, point and
    evaluation count, for a deterministic acquisition and for x -> myrand(model, x) with the library's generator -- and the value it
    reports is the ORACLE's score at the point it reports."""
    from bohip import _lib
    from bohip.acquisition import _batched_direct_l

    lib = _lib.load()
    X, y, _ = synth(N, d, 4, seed=3 * N + d)
    ll = np.linspace(-0.9, -0.3, d)
    m = make_model(bohip, X, y, ll, 0.0, -2.0, 0.1)
    lb, ub = np.full(d, -0.2), np.full(d, 1.1)
    L, alpha = orc.fit(X, y, ll, 0.0, -2.0, 0.1)
    for acq, p in [("UCB", [2.0]), ("EI", [float(np.median(y))])]:
        calls = []
        f1, x1, e1 = _batched_direct_l(lambda Z: (calls.append(Z.shape[1]), m.score(acq, p, Z)[0])[1], lb, ub, maxeval)
        f2, x2, e2, dc = m.direct_max(acq, p, lb, ub, maxeval)
        assert (f1, e1, len(calls)) == (f2, e2, dc) and np.array_equal(x1, x2) and e2 <= maxeval
        sc_o, _, _ = orc.score(X, ll, 0.0, 0.1, L, alpha, acq, p, x2[None, :])
        assert abs(f2 - sc_o[0]) <= 1e-6 * abs(sc_o[0]) + 1e-9
        assert np.all(x2 >= lb) and np.all(x2 <= ub)
    # one posterior draw per evaluated point, z = bohip_thompson_normal(seed, 0, e) for the e-th evaluation
    seed = 77
    cnt = [0]

    def f_draw(Z):
        mu, var = m.predict_f(Z)
        z = np.array([lib.bohip_thompson_normal(seed, 0, cnt[0] + j) for j in range(Z.shape[1])])
        cnt[0] += Z.shape[1]
        return mu + np.sqrt(np.maximum(var, 0.0)) * z
    f1, x1, e1 = _batched_direct_l(f_draw, lb, ub, maxeval)
    f2, x2, e2, dc = m.direct_max("ThompsonDraw", None, lb, ub, maxeval, seed=seed)
    assert (f1, e1) == (f2, e2) and np.array_equal(x1, x2) and cnt[0] == e2
    f3, x3, _, _ = m.direct_max("ThompsonDraw", None, lb, ub, maxeval, seed=seed + 1)
    assert f3 != f2                                                     # another seed, another draw
    # stopval / maxtime / error paths
    f4, _, e4, _ = m.direct_max("UCB", [2.0], lb, ub, maxeval, stopval=-1e300)
    assert e4 == 1
    with pytest.raises(bohip.BohipError):
        m.direct_max("UCB", [2.0], ub, lb, maxeval)                     # lower bound above upper bound
    # acquire_max takes this route for a device model
    from bohip.acquisition import acquire_max, ThompsonSamplingSimple, defaultoptions
    f5, x5 = acquire_max(ThompsonSamplingSimple(), m, lb, ub, defaultoptions(type(m), ThompsonSamplingSimple), rng=np.random.default_rng(1),
                         setparams=False)
    assert np.isfinite(f5) and np.all(x5 >= lb) and np.all(x5 <= ub)
    mu5, _ = m.predict_f(x5[:, None])
    assert mu5[0] > np.quantile(y, 0.5)


@pytest.mark.parametrize("N,d", [(3600, 6), (4900, 8)])
def test_mid_size_refit_with_the_inverse_in_group_form_vs_oracle(bohip, orc, N, d):
    """28 ... 39 row tiles: the sizes whose inverse queues take the group form since the end of round 6.  The whole
    factor, alpha and the posterior at 300 candidates against the oracle; W against L (L W = I); the executor form ran and nothing timed out."""
    from bohip import _lib

    X, y, Xs = synth(N, d, 300, seed=7 * N)
    ll = np.linspace(-0.9, -0.4, d)
    L, alpha = orc.fit(X, y, ll, 0.1, -2.0, 0.2)
    m = make_model(bohip, X, y, ll, 0.1, -2.0, 0.2)
    m.fit_()
    assert m.info(_lib.INFO_CHOL_FORM) == 4 and m.info(_lib.INFO_CHOL_FALLBACKS) == 0
    s2f = math.exp(0.2)
    np.testing.assert_allclose(m.factor(), L, rtol=1e-9, atol=1e-11 * math.sqrt(s2f))
    np.testing.assert_allclose(m.alpha(), alpha, rtol=1e-6, atol=1e-9 * np.abs(alpha).max())
    mu_o, var_o = orc.predict(X, ll, 0.1, 0.2, L, alpha, Xs, nthreads=8)
    mu, var = m.predict_f(Xs.T)
    assert np.all(np.abs(mu - mu_o) <= 1e-6 * np.abs(mu_o) + mu_floor(alpha, s2f))
    assert np.all(np.abs(var - var_o) <= var_tol(var_o, N, s2f))
    # sigma^2 = k** - |W k*|^2 uses W = L^-1 from the inverse queues: check it against the factor directly on random vectors
    Kstar = np.exp(-0.5 * (((Xs[:4, None, :] - X[None, :, :]) / np.exp(ll)) ** 2).sum(-1)) * s2f     # (4, N)
    v = np.linalg.solve(L, Kstar.T)                                                                 # L^-1 k*
    np.testing.assert_allclose(s2f - (v ** 2).sum(0), var[:4], rtol=1e-7, atol=1e-9)


# ---- BASELINE.json full sizes: size-independent properties + a bounded oracle sample ------------------
def test_full_size_c2_properties(bohip, orc):
    N, d, R = 3000, 8, 4096
    X, y, Xs = synth(N, d, R, seed=0)
    ll = np.full(d, math.log(0.5))
    m = make_model(bohip, X, y, ll, 0.0, -2.0, 0.0)
    tau = float(y.max())
    sc, bv, bi = m.score("EI", [tau], Xs.T)
    mu, var = m.predict_f(Xs.T)
    assert np.all(np.isfinite(sc)) and np.all(var >= 0) and np.all(var <= 1.0 + 1e-12)     # 0 <= sigma^2 <= s_f^2
    assert bi == int(np.argmax(sc)) and bv == sc[bi]
    # determinism: same call twice -> bit-identical; permuted candidates -> permuted scores, same winner
    sc2, bv2, bi2 = m.score("EI", [tau], Xs.T)
    np.testing.assert_array_equal(sc, sc2)
    perm = np.random.default_rng(1).permutation(R)
    scp, bvp, bip = m.score("EI", [tau], Xs[perm].T)
    np.testing.assert_array_equal(scp, sc[perm])
    assert perm[bip] == bi and bvp == bv
    # sharding invariance: 8 shards of 512 (the 8-GPU partition) reduce to the same winner
    from bohip.dist import reduce_best, shard_bounds
    recs = []
    m.set_batch_hint(R)     # every rank announces the size of the WHOLE set -> the summation schedule of the unsharded call
    for g in range(8):
        lo, hi = shard_bounds(R, 8, g)
        s_g, v_g, i_g = m.score("EI", [tau], Xs[lo:hi].T)
        np.testing.assert_array_equal(s_g, sc[lo:hi])
        recs.append((v_g, i_g + lo))
    assert reduce_best(*zip(*recs)) == (bv, bi)
    m.set_batch_hint(0)
    # without the hint a 512-candidate shard takes the split-K schedule: same numbers to rounding, same winner here
    s_0, v_0, i_0 = m.score("EI", [tau], Xs[:512].T)
    np.testing.assert_allclose(s_0, sc[:512], rtol=1e-9, atol=1e-300)
    # at an observation the posterior mean interpolates within the noise level and sigma^2 collapses
    mu_x, var_x = m.predict_f(X[:64].T)
    assert np.all(var_x < 0.05)
    # factor identity L L' = cK on a row sample (checks Cholesky without an O(N^3) CPU run)
    Lg = m.factor()
    cK = orc.build_cK(X, ll, 0.0, -2.0)
    rows = np.random.default_rng(2).choice(N, 24, replace=False)
    np.testing.assert_allclose(Lg[rows] @ Lg.T, cK[rows], rtol=0, atol=1e-11)
    # the oracle on ALL 4096 candidates at full N (every host core; ~10 s of one core): every score, mu and sigma^2 within the
    # stated tolerance, and the device's winner IS the first arg-max of the ORACLE's scores (north_star: "argmax indices bit-exact")
    L, alpha = orc.fit(X, y, ll, 0.0, -2.0, 0.0)
    nth = min(128, os.cpu_count() or 8)
    sc_o, bv_o, bi_o = orc.score(X, ll, 0.0, 0.0, L, alpha, "EI", [tau], Xs, nthreads=nth)
    check_scores(sc, sc_o, mu_floor(alpha, 1.0) + 1e-13)
    assert bi == bi_o == int(np.argmax(sc_o)), (bi, bi_o, np.sort(sc_o)[-3:])
    assert abs(bv - bv_o) <= 1e-6 * abs(bv_o)
    mu_o, var_o = orc.predict(X, ll, 0.0, 0.0, L, alpha, Xs, nthreads=nth)
    assert np.all(np.abs(var - var_o) <= var_tol(var_o, N, 1.0))
    assert np.all(np.abs(mu - mu_o) <= 1e-6 * np.abs(mu_o) + mu_floor(alpha, 1.0))


def test_small_batch_path_agrees_with_mfma_path(bohip, orc):
    """Up to min(256, 90 + 300000/N) candidates take the row-wise small-batch kernels (the reference's default: 10
    L-BFGS restarts); larger batches the MFMA engine.  Same numbers to rounding, both against the oracle, values and gradients."""
    X, y, Xs = synth(900, 5, 600, seed=23)                                 # 600 > 256: the MFMA engine at any N
    ll = np.linspace(-0.7, -0.3, 5)
    L, alpha = orc.fit(X, y, ll, 0.2, -2.0, 0.1)
    m = make_model(bohip, X, y, ll, 0.2, -2.0, 0.1)
    tau = float(y.max())
    sc_big, g_big = m.score_grad("EI", [tau], Xs.T)                       # MFMA engine
    for lo, n in [(0, 1), (3, 7), (10, 8), (40, 9), (100, 32), (50, 33), (20, 70), (100, 96), (300, 255), (5, 256)]:   # row-wise path
        sc, g = m.score_grad("EI", [tau], Xs[lo:lo + n].T)
        np.testing.assert_allclose(sc, sc_big[lo:lo + n], rtol=1e-11, atol=1e-14)
        np.testing.assert_allclose(g, g_big[:, lo:lo + n], rtol=1e-9, atol=1e-12 * np.abs(g_big).max())
        sc_o, g_o = orc.score_grad(X, ll, 0.2, 0.1, L, alpha, "EI", [tau], Xs[lo:lo + n])
        np.testing.assert_allclose(sc, sc_o, rtol=1e-6, atol=mu_floor(alpha, math.exp(0.4)) + 1e-12)
        np.testing.assert_allclose(g.T, g_o, rtol=1e-6, atol=1e-9 * np.abs(g_o).max())
        mu, var = m.predict_f(Xs[lo:lo + n].T)
        mu_o, var_o = orc.predict(X, ll, 0.2, 0.1, L, alpha, Xs[lo:lo + n])
        np.testing.assert_allclose(mu, mu_o, rtol=1e-6, atol=mu_floor(alpha, math.exp(0.4)))
        assert np.all(np.abs(var - var_o) <= var_tol(var_o, 900, math.exp(0.4)))
        _, bv, bi = m.score("EI", [tau], Xs[lo:lo + n].T)
        assert bi == int(np.argmax(sc)) and bv == sc[bi]


@pytest.mark.parametrize("N,d,R", [(1500, 5, 300), (1100, 3, 700), (3000, 8, 257)])
def test_split_k_path_vs_oracle_and_whole_k(bohip, orc, N, d, R):
    """A few hundred candidates leave the whole-K MFMA jobs on a handful of CUs, so the contraction index is cut into
    slices (k_gemm_nt batched over slices + k_split_combine_v/u).  Against the oracle, and against the whole-K schedule
    (forced through the batch hint) to rounding; values and gradients; bit-deterministic and batch-independent."""
    X, y, Xs = synth(N, d, R, seed=33)
    ll = np.linspace(-0.8, -0.4, d)
    L, alpha = orc.fit(X, y, ll, 0.1, -2.0, 0.05)
    m = make_model(bohip, X, y, ll, 0.1, -2.0, 0.05)
    tau = float(y.max())
    s2f = math.exp(0.2)
    sc, g = m.score_grad("UCB", [2.0], Xs.T)
    sub = np.random.default_rng(1).choice(R, 120, replace=False)
    sc_o, g_o = orc.score_grad(X, ll, 0.1, 0.05, L, alpha, "UCB", [2.0], Xs[sub])
    _, var_o = orc.predict(X, ll, 0.1, 0.05, L, alpha, Xs[sub])
    floor = mu_floor(alpha, s2f) + 2.0 * np.sqrt(var_tol(var_o, N, s2f, rel=0))
    check_scores(sc[sub], sc_o, floor)
    good = var_o > 1e3 * var_tol(var_o, N, s2f)
    np.testing.assert_allclose(g.T[sub][good], g_o[good], rtol=1e-5, atol=1e-7 * np.abs(g_o).max())
    sv, bv, bi = m.score("UCB", [2.0], Xs.T)
    np.testing.assert_array_equal(sv, sc)                                  # value path == gradient path, bit for bit
    assert bi == int(np.argmax(sv)) and bv == sv[bi]
    np.testing.assert_array_equal(m.score("UCB", [2.0], Xs[40:R - 3].T)[0], sv[40:R - 3])   # batch-independent (same schedule)
    m.set_batch_hint(1 << 20)                                              # whole-K schedule
    sw, gw = m.score_grad("UCB", [2.0], Xs.T)
    m.set_batch_hint(0)
    np.testing.assert_allclose(sc, sw, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(g, gw, rtol=1e-7, atol=1e-10 * np.abs(gw).max())


def test_candidate_chunking_is_invisible(bohip, orc):
    """R larger than one K*' chunk (65536 rows at this N; two equal chunks of 35328 here): chunk boundaries, 64-wide
    tile padding and the per-chunk gradient buffers must not show in the results."""
    X, y, Xs = synth(200, 3, 70001, seed=17)
    ll = np.array([-0.4, -0.9, -0.1])
    L, alpha = orc.fit(X, y, ll, 0.0, -2.0, 0.1)
    m = make_model(bohip, X, y, ll, 0.0, -2.0, 0.1)
    tau = float(y.max())
    sc_o, bv_o, bi_o = orc.score(X, ll, 0.0, 0.1, L, alpha, "EI", [tau], Xs, nthreads=8)
    sc, bv, bi = m.score("EI", [tau], Xs.T)
    check_scores(sc, sc_o, mu_floor(alpha, 1.0) + 1e-13)
    assert bi == bi_o and sc[bi] == bv
    from bohip import _lib
    ch = m.info(_lib.INFO_SCORE_CHUNK)
    assert 0 < ch < 70001 and m.info(_lib.INFO_SCORE_LAUNCHES) == 2           # the batch really was cut
    for lo, hi in [(0, ch), (ch - 192, ch + 208), (ch, 70001)]:               # any sub-batch on the same (MFMA) path reproduces its slice bit-for-bit
        np.testing.assert_array_equal(m.score("EI", [tau], Xs[lo:hi].T)[0], sc[lo:hi])
    sg, g = m.score_grad("EI", [tau], Xs.T)
    np.testing.assert_array_equal(sg, sc)
    lo, hi = ch - 192, ch + 208
    _, g_o = orc.score_grad(X, ll, 0.0, 0.1, L, alpha, "EI", [tau], Xs[lo:hi])
    np.testing.assert_allclose(g[:, lo:hi].T, g_o, rtol=1e-6, atol=1e-9 * np.abs(g_o).max())
    bvs, bis = m.thompson(Xs.T, 4, seed=3)
    mu, var = m.predict_f(Xs.T)
    lib = _lib.load()
    for s_ in range(4):
        f = mu[bis[s_]] + math.sqrt(var[bis[s_]]) * lib.bohip_thompson_normal(3, s_, int(bis[s_]))
        assert f == pytest.approx(bvs[s_], rel=1e-12)


def test_full_size_c4_properties(bohip, orc):
    """BASELINE configs[3]: N=10000, d=16 (blocked-Cholesky path).  An O(N^3) CPU factorisation is out of reach
    for a test, so the factor is pinned by L L' = cK on sampled rows, W by L-solves against the oracle's
    substitution on the SAME factor, and the scoring path by the oracle's predict on a candidate sample."""
    from oracle.oracle import np_cov

    N, d, R = 10000, 16, 640
    X, y, Xs = synth(N, d, R, seed=4)
    ll = np.full(d, math.log(0.7))
    m = make_model(bohip, X, y, ll, 0.0, -2.0, 0.0)
    Lg = m.factor()
    assert np.all(np.diag(Lg) > 0) and np.all(np.triu(Lg, 1) == 0)
    rows = np.random.default_rng(1).choice(N, 16, replace=False)
    cK_rows = np_cov("SEArd", X[rows], X, ll, 0.0)
    cK_rows[np.arange(16), rows] += math.exp(-4.0) + EPS
    np.testing.assert_allclose(Lg[rows] @ Lg.T, cK_rows, rtol=0, atol=2e-11)
    alpha = m.alpha()
    np.testing.assert_allclose(Lg @ (Lg.T @ alpha), y, rtol=0, atol=1e-8)          # cK alpha = y - beta
    tau = float(y.max())
    sc, bv, bi = m.score("EI", [tau], Xs.T)
    mu, var = m.predict_f(Xs.T)
    assert bi == int(np.argmax(sc)) and np.all(var >= 0) and np.all(var <= 1 + 1e-12)
    sel = np.unique(np.concatenate([[bi], np.arange(0, R, 40)]))
    mu_o, var_o = orc.predict(X, ll, 0.0, 0.0, Lg, alpha, Xs[sel], nthreads=8)      # substitution on the same factor
    assert np.all(np.abs(mu[sel] - mu_o) <= 1e-6 * np.abs(mu_o) + mu_floor(alpha, 1.0))
    assert np.all(np.abs(var[sel] - var_o) <= var_tol(var_o, N, 1.0))
    sc_o, _, _ = orc.score(X, ll, 0.0, 0.0, Lg, alpha, "EI", [tau], Xs[sel], nthreads=8)
    check_scores(sc[sel], sc_o, mu_floor(alpha, 1.0) + var_tol(var_o, N, 1.0, rel=0) + 1e-13)
    np.testing.assert_array_equal(sc, m.score("EI", [tau], Xs.T)[0])               # deterministic


def test_device_resident_entry_point_matches_host_entry_point(bohip):
    torch = pytest.importorskip("torch")
    from bohip import _lib
    from bohip.dist import allgather_best

    lib = _lib.load()
    X, y, Xs = synth(300, 8, 1000, seed=12)
    m = make_model(bohip, X, y, np.full(8, -0.6), 0.0, -2.0, 0.0)
    sc, bv, bi = m.score("EI", [y.max()], Xs.T)
    dXs = torch.from_numpy(Xs).cuda()
    dsc = torch.empty(1000, dtype=torch.float64, device="cuda")
    rec = torch.zeros(2, dtype=torch.int64, device="cuda")
    _lib.check(lib.bohip_gp_set_stream(m._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    p = (C.c_double * 2)(float(y.max()), 0.0)
    _lib.check(lib.bohip_gp_score_dev(m._h, _lib.ACQ["EI"], p, C.c_void_p(dXs.data_ptr()), 1000,
                                      C.c_void_p(dsc.data_ptr()), C.c_void_p(rec.data_ptr())))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dsc.cpu().numpy(), sc)
    assert allgather_best(rec, 0, 1) == (bv, bi)
    _lib.check(lib.bohip_gp_set_stream(m._h, None))


def test_device_ascent_matches_host_restatement(bohip, orc):
    """bohip_gp_acquire_max (lock-step projected L-BFGS on the device) against the NumPy restatement of the
    same search driven by the same device score_grad: same maximisers to optimiser tolerance, never worse than the start,
    inside the box, first-maximum-wins over the starts."""
    from bohip.acquisition import _batched_lbfgs_ascent

    X, y, _ = synth(120, 2, 4, seed=12)
    m = make_model(bohip, X, y, np.array([-1.0, -0.7]), 0.3, -2.0, 0.0)
    lb, ub = np.zeros(2), np.ones(2)
    rng = np.random.default_rng(3)
    for acq, p in [("UCB", [2.0]), ("EI", [float(y.max())]), ("MaxMean", [])]:
        for R in (1, 7, 40):                      # row-wise path and MFMA path
            starts = np.asfortranarray(rng.random((2, R)))
            f0, _ = m.score_grad(acq, p, starts)
            f, Xb, bf, bi, bx, ev = m.ascend(acq, p, lb, ub, starts, maxeval=200)
            fh, Xh = _batched_lbfgs_ascent(lambda Z: m.score_grad(acq, p, Z), starts, lb, ub, 200)
            assert ev >= 1 and ev <= 200
            assert np.all(f >= f0 - 1e-12) and np.all(Xb >= lb[:, None]) and np.all(Xb <= ub[:, None])
            # value at the returned point is the returned value
            fchk, _ = m.score_grad(acq, p, Xb)
            np.testing.assert_allclose(fchk, f, rtol=1e-9, atol=1e-12)
            # both searches climb to the same local maxima (same algorithm, different reduction order)
            np.testing.assert_allclose(f, fh, rtol=1e-5, atol=1e-8)
            j = int(np.argmax(f))                 # numpy argmax = first maximum
            assert bi == j and bf == f[j]
            np.testing.assert_array_equal(bx, Xb[:, j])


def test_free_running_ascent_follows_the_lock_step_trajectories():
    """csrc/kernels_ascent.hip k_asc_step: every start point on its own schedule (no host decision between two evaluation
    passes) against the lock-step driver (BOHIP_ASC_LOCKSTEP=1).  A start point's arithmetic never looks at another one, so with
    room to converge the end points and values are identical bit for bit, and the free-running form needs no more passes."""
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT

    code = r'''
import json, sys
sys.path.insert(0, %r)
import numpy as np, bohip
out = {}
for N, d, R in ((700, 3, 10), (1100, 2, 37), (900, 4, 300), (800, 9, 13), (600, 16, 16), (650, 5, 1)):
    rng = np.random.default_rng(N)
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, -0.9), 0.2), logNoise=-2.0, capacity=N)
    m.append_(X.T, y)
    starts = np.asfortranarray(rng.random((d, R)))
    for acq, p in (("UCB", [2.0]), ("EI", [float(y.max())])):
        f, Xb, bf, bi, bx, ev = m.ascend(acq, p, np.zeros(d), np.ones(d), starts, maxeval=3000)
        out["%%d-%%s" %% (N, acq)] = dict(f=f.tolist(), X=Xb.tolist(), bf=bf, bi=int(bi), bx=bx.tolist(), ev=int(ev))
print("RESULT" + json.dumps(out))
''' % ROOT
    res = {}
    # round 6: with <= 16 start points and d <= 16 the step runs in the gradient kernel's last workgroup, four start points per wave
    # (kernels_small.hip SmallFold); BOHIP_ASC_LOCKSTEP=2 keeps it a launch of its own (k_asc_step) -- the same trajectories again
    for name, env in (("free", {}), ("own_launch", {"BOHIP_ASC_LOCKSTEP": "2"}), ("lockstep", {"BOHIP_ASC_LOCKSTEP": "1"})):
        o = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert o.returncode == 0, (name, o.stderr[-2000:])
        res[name] = json.loads([l for l in o.stdout.splitlines() if l.startswith("RESULT")][-1][6:])
    for key, a in res["free"].items():
        b = res["lockstep"][key]
        assert a["f"] == b["f"] and a["X"] == b["X"], key
        assert (a["bf"], a["bi"], a["bx"]) == (b["bf"], b["bi"], b["bx"]), key
        assert 1 <= a["ev"] <= b["ev"], (key, a["ev"], b["ev"])
        assert a == res["own_launch"][key], key


def test_one_workgroup_per_start_ascent_against_the_batched_driver():
    """csrc/kernels_ascent.hip k_ascent_wg (models up to 256 observations: the whole acquire_max is ONE launch, one workgroup per start
    point) against the free-running batched driver (BOHIP_ASC_WG_NMAX=0).  Same search code, different summation orders in the
    posterior: same maxima to optimiser tolerance, same winner, and the returned value is the score at the returned point."""
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT

    code = r'''
import json, sys
sys.path.insert(0, %r)
import numpy as np, bohip
out = {}
for N, d, R, kern in ((40, 2, 10, "SEArd"), (200, 3, 33, "SEArd"), (256, 8, 10, "Mat52Ard"), (130, 12, 17, "SEArd"), (1, 1, 5, "SEArd")):
    rng = np.random.default_rng(N)
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    K = getattr(bohip, kern)
    m = bohip.ElasticGPE(d, mean=bohip.MeanConst(0.3), kernel=K(np.full(d, -0.9), 0.2), logNoise=-2.0, capacity=N)
    m.append_(X.T, y)
    starts = np.asfortranarray(rng.random((d, R)) * 1.4 - 0.2)      # partly outside the box
    for acq, p in (("UCB", [2.0]), ("EI", [float(y.max())]), ("PI", [float(y.max())]), ("MaxMean", [])):
        f, Xb, bf, bi, bx, ev = m.ascend(acq, p, np.zeros(d), np.ones(d), starts, maxeval=400)
        fchk, _ = m.score_grad(acq, p, Xb)
        out["%%d-%%s" %% (N, acq)] = dict(f=f.tolist(), X=Xb.tolist(), bf=bf, bi=int(bi), bx=bx.tolist(), ev=int(ev), fchk=fchk.tolist())
print("RESULT" + json.dumps(out))
''' % ROOT
    res = {}
    for name, env in (("wg", {}), ("batched", {"BOHIP_ASC_WG_NMAX": "0"})):
        o = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert o.returncode == 0, (name, o.stderr[-2000:])
        res[name] = json.loads([l for l in o.stdout.splitlines() if l.startswith("RESULT")][-1][6:])
    for key, a in res["wg"].items():
        b = res["batched"][key]
        fa, fb = np.array(a["f"]), np.array(b["f"])
        Xa = np.array(a["X"])
        assert np.all(Xa >= 0.0) and np.all(Xa <= 1.0), key
        np.testing.assert_allclose(np.array(a["fchk"]), fa, rtol=1e-9, atol=1e-12, err_msg=key)   # value at the returned point
        np.testing.assert_allclose(fa, fb, rtol=1e-5, atol=1e-8, err_msg=key)                     # same local maxima
        assert a["bf"] == pytest.approx(b["bf"], rel=1e-6, abs=1e-9), key
        assert a["bf"] == fa[a["bi"]] and a["bx"] == Xa[:, a["bi"]].tolist(), key                # first maximum wins
        assert 1 <= a["ev"] <= 400, key


def test_device_ascent_on_the_split_k_and_whole_k_schedules(bohip):
    """Restart counts beyond the row-wise path: 300 (split-K) and 1500 (whole-K jobs) starts at N = 1100."""
    from bohip.acquisition import _batched_lbfgs_ascent

    X, y, _ = synth(1100, 3, 4, seed=14)
    m = make_model(bohip, X, y, np.array([-1.0, -0.8, -0.6]), 0.2, -2.0, 0.0)
    lb, ub = np.zeros(3), np.ones(3)
    rng = np.random.default_rng(4)
    for R in (300, 1500):
        starts = np.asfortranarray(rng.random((3, R)))
        f0, _ = m.score_grad("UCB", [2.0], starts)
        # (maxeval: enough for BOTH drivers to converge -- under a cap they differ by construction: the free-running device form spends a pass
        # per start and trial, the lock-step host form a pass per trial of the slowest line search)
        f, Xb, bf, bi, bx, ev = m.ascend("UCB", [2.0], lb, ub, starts, maxeval=400)
        fh, Xh = _batched_lbfgs_ascent(lambda Z: m.score_grad("UCB", [2.0], Z), starts, lb, ub, 400)
        assert ev < 400
        assert np.all(f >= f0 - 1e-12) and np.all(Xb >= 0) and np.all(Xb <= 1)
        fchk, _ = m.score_grad("UCB", [2.0], Xb)
        np.testing.assert_allclose(fchk, f, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(f, fh, rtol=1e-5, atol=1e-8)
        j = int(np.argmax(f))
        assert bi == j and bf == f[j]


def test_device_ascent_respects_bounds_and_edge_cases(bohip):
    X, y, _ = synth(60, 3, 4, seed=2)
    m = make_model(bohip, X, y, np.zeros(3), 0.0, -1.0, 0.0)
    lb, ub = np.array([0.2, 0.0, 0.5]), np.array([0.4, 1.0, 0.5])     # third coordinate pinned
    starts = np.asfortranarray(np.random.default_rng(0).random((3, 5)) * 2 - 0.5)   # partly outside the box
    f, Xb, bf, bi, bx, ev = m.ascend("UCB", [1.5], lb, ub, starts, maxeval=50)
    assert np.all(Xb >= lb[:, None] - 0) and np.all(Xb <= ub[:, None] + 0) and np.all(Xb[2] == 0.5)
    # maxeval = 1: only the clipped starts are evaluated
    f1, X1, *_ = m.ascend("UCB", [1.5], lb, ub, starts, maxeval=1)
    np.testing.assert_array_equal(X1, np.clip(starts, lb[:, None], ub[:, None]))
    # a NaN start never wins and does not poison the others
    starts[:, 1] = np.nan
    f2, X2, bf2, bi2, bx2, _ = m.ascend("UCB", [1.5], lb, ub, starts, maxeval=50)
    assert bi2 != 1 and np.isfinite(bf2)
    with pytest.raises(ValueError):
        m.ascend("UCB", [1.5], lb[:2], ub, starts)


@pytest.mark.parametrize("N", [220, 600])
def test_device_ascent_against_scipy_lbfgsb_on_the_oracle(bohip, orc, N):
    """Independent check of bohip_gp_acquire_max (role of NLopt :LD_LBFGS at reference src/acquisition.jl:59): SciPy's
    L-BFGS-B (the Fortran code NLopt's LBFGS descends from) maximises the ORACLE's value + analytic gradient from the same
    starts under the same box; nothing of the device is inside that loop.  Per start the device's end point must be a
    KKT point of the oracle's objective with the oracle's value; over the starts the best maximum and maximiser agree.
    N = 220 runs the one-launch form (k_ascent_wg, models of <= 256 observations), N = 600 the free-running batched driver
    (k_asc_step: the one behind the headline model's `default_usage`); the test below repeats N = 220 with the batched driver forced."""
    from scipy.optimize import minimize

    X, y, _ = synth(N, 3, 1, seed=40)
    ll = np.array([-0.9, -0.6, -0.75])
    lsig, lnoise, beta = 0.1, -2.0, 0.2
    L, alpha = orc.fit(X, y, ll, lsig, lnoise, beta)
    m = make_model(bohip, X, y, ll, lsig, lnoise, beta)
    R = 16
    starts = np.random.default_rng(41).random((3, R))
    lb, ub = np.zeros(3), np.ones(3)
    tau = float(np.median(y))          # an incumbent the posterior mean exceeds somewhere: EI is not ~1e-50 everywhere
    for acq, p in [("EI", [tau]), ("UCB", [2.0]), ("MaxMean", [])]:
        f, Xd, bf, bi, bx, ev = m.ascend(acq, p, lb, ub, starts, maxeval=2000, ftol_rel=1e-13, xtol_abs=1e-13)
        assert 2 <= ev <= 2000 and np.all(Xd >= 0) and np.all(Xd <= 1)
        cnt = [0]

        def negfg(x):
            cnt[0] += 1
            sc, g = orc.score_grad(X, ll, lsig, beta, L, alpha, acq, p if p else [0.0], x[None, :].copy())
            return -float(sc[0]), -g[0]

        fs, xs, nf = np.empty(R), np.empty((3, R)), []
        for r in range(R):
            cnt[0] = 0
            res = minimize(negfg, starts[:, r], jac=True, method="L-BFGS-B", bounds=[(0.0, 1.0)] * 3,
                           options=dict(maxiter=2000, maxfun=4000, ftol=1e-15, gtol=1e-12))
            fs[r], xs[:, r] = -res.fun, res.x
            nf.append(cnt[0])
        # evaluation passes (all starts advance in every pass: the slowest start decides) against SciPy's evaluations for its slowest start:
        # UCB / MaxMean need fewer.  (EI has no bound: a start on the exponential flank of EI -- no curvature pair is ever accepted there --
        # improves by a constant FACTOR per pass for as long as maxeval lets it, as NLopt's search without tolerances would; SciPy's ftol,
        # absolute below 1, leaves it.  DESIGN 6c.)
        if acq != "EI":
            assert ev <= max(nf) + 10, (acq, ev, max(nf))
        sc_o, g_o = orc.score_grad(X, ll, lsig, beta, L, alpha, acq, p if p else [0.0], np.ascontiguousarray(Xd.T))
        scale = max(np.abs(fs).max(), 1e-300)
        np.testing.assert_allclose(f, sc_o, rtol=1e-6, atol=1e-9 * scale)           # the device reports the oracle's value there
        pg = np.where(((Xd.T <= 0) & (g_o < 0)) | ((Xd.T >= 1) & (g_o > 0)), 0.0, g_o)   # projected gradient (maximisation)
        g0 = orc.score_grad(X, ll, lsig, beta, L, alpha, acq, p if p else [0.0], np.ascontiguousarray(starts.T))[1]
        # per start a KKT point -- every one of them (until round 6 a start whose line search failed twelve halvings in a row was given up
        # short of it, 0.2-0.5 % of EI starts: the line search now has 30 trial points, tools/ascent_kkt_margin.py reads 0 of 576)
        kkt = np.abs(pg).max(1) <= 2e-4 * np.abs(g0).max()
        assert kkt.all(), (acq, np.abs(pg).max(1), np.abs(g0).max())
        assert np.all(f >= orc.score(X, ll, lsig, beta, L, alpha, acq, p if p else [0.0], np.ascontiguousarray(starts.T))[0] - 1e-12 * scale)
        # and per start no worse than SciPy from the same start (one exception allowed: two searches may fall into neighbouring maxima)
        assert np.sum(f >= fs - 1e-6 * scale) >= R - 1, (acq, f, fs)
        same = np.abs(f - fs) <= 1e-6 * scale                                        # same local maximum as SciPy from that start
        assert same.mean() >= 0.75, (acq, same.mean())
        sharp = same & (fs >= 0.5 * fs.max()) if fs.max() > 0 else same                # plateaus (EI ~ 0) have no unique maximiser
        np.testing.assert_allclose(Xd[:, sharp], xs[:, sharp], atol=1e-3)
        assert bf == pytest.approx(fs.max(), rel=1e-6, abs=1e-9 * scale)            # acquire_max's answer: the best over the starts
        np.testing.assert_allclose(bx, xs[:, int(np.argmax(fs))], atol=2e-4)


def test_scipy_check_of_the_batched_ascent_driver_at_small_N():
    """The N = 220 case above with BOHIP_ASC_WG_NMAX=0: the free-running batched driver (k_asc_step) instead of the one-launch form.
    The switch is read once per process, hence the nested pytest run."""
    import subprocess
    import sys

    from conftest import ROOT

    if os.environ.get("BOHIP_NESTED_PYTEST"):
        pytest.skip("already inside the nested run")
    env = dict(os.environ, BOHIP_ASC_WG_NMAX="0", BOHIP_NESTED_PYTEST="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_parity_gpu.py") + "::test_device_ascent_against_scipy_lbfgsb_on_the_oracle"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("N", [120, 600])
def test_device_ascent_nlopt_stop_criteria(bohip, N):
    """ftol_abs / xtol_rel / stopval (bohip_gp_set_ascent_stop): the NLopt properties the reference forwards with setproperty!
    (src/acquisition.jl:24-27; its own test passes ftol_abs = eps(), test/acquisition.jl:6,9).  Both device forms (N = 120: one
    launch per acquire_max; N = 600: the batched driver) against the host restatement of the same search with the same settings,
    and the properties each criterion promises."""
    from bohip.acquisition import _batched_lbfgs_ascent

    X, y, _ = synth(N, 3, 1, seed=77)
    m = make_model(bohip, X, y, np.array([-1.0, -0.8, -0.6]), 0.2, -2.0, 0.0)
    lb, ub = np.zeros(3), np.ones(3)
    starts = np.asfortranarray(np.random.default_rng(5).random((3, 12)))
    fg = lambda Z: m.score_grad("UCB", [2.0], Z)     # noqa: E731
    f0, _ = fg(starts)
    base = m.ascend("UCB", [2.0], lb, ub, starts, maxeval=400)
    eps = float(np.finfo(float).eps)
    runs = {}
    for name, kw in [("eps", dict(ftol_abs=eps)), ("loose", dict(ftol_abs=1e-2)), ("xrel", dict(xtol_rel=0.2)),
                     ("stop", dict(stopval=float(0.5 * (f0.mean() + base[0].max()))))]:
        m.set_ascent_stop(**kw)
        runs[name] = m.ascend("UCB", [2.0], lb, ub, starts, maxeval=400)
        fh, Xh = _batched_lbfgs_ascent(fg, starts, lb, ub, 400, **kw)
        np.testing.assert_allclose(runs[name][0], fh, rtol=1e-5, atol=1e-8, err_msg=name)      # device == host restatement
    m.set_ascent_stop()                                                    # NLopt's defaults: all three off
    again = m.ascend("UCB", [2.0], lb, ub, starts, maxeval=400)
    np.testing.assert_array_equal(again[0], base[0])                       # and the default search is what it was
    np.testing.assert_allclose(runs["eps"][0], base[0], rtol=1e-9)         # ftol_abs = eps() changes nothing visible (the reference's setting)
    for name in ("loose", "xrel", "stop"):
        assert runs[name][5] <= base[5] and np.all(runs[name][0] >= f0 - 1e-12), name     # stops no later, never below the start
    assert runs["loose"][5] < base[5] and np.all(base[0] - runs["loose"][0] <= 0.5)       # gave up within a few ftol_abs of the maximum
    sv = float(0.5 * (f0.mean() + base[0].max()))                          # half-way up from the average start value
    reached = base[0] >= sv                                                # start points whose ascent can reach stopval at all
    assert np.all(runs["stop"][0][reached] >= sv) and np.any(runs["stop"][0] < base[0] - 1e-9)   # stopped AT the first value >= stopval


def test_dataflow_cholesky_matches_the_launch_chained_one(bohip, orc):
    """csrc/kernels_chol.hip, csrc/kernels_exec.hip: the dataflow forms of the factorisation (persistent chain workgroups +
    flags) against the launch-chained one -- form 1 (panel followers, T <= 46 by default), form 2 (flagged row solves + K = 128
    window updates), form 2 with left-looking window updates, and the executor form (one persistent kernel pulling tile tasks;
    the default for 47..96 row tiles), each forced at sizes the test can afford.
    Same factor to rounding, same alpha and posterior; BOHIP_CHOL_DF_STRICT turns a timed-out flag into an error instead of the
    silent fall-back.  Subprocesses because the switches are read once per process."""
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT

    code = r'''
import json, sys
sys.path.insert(0, %r)
import numpy as np, bohip
out = {}
for N, d in ((130, 2), (1000, 4), (3000, 8)):
    rng = np.random.default_rng(N)
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, -0.6), 0.1), logNoise=-2.0, capacity=N)
    m.append_(X.T, y)
    L = m.factor()
    mu, var = m.predict_f(X[:50].T + 0.01)
    out[str(N)] = dict(Lsum=float(np.abs(L).sum()), Ldiag=np.diag(L)[::97].tolist(), Llast=L[-1, ::211].tolist(),
                       alpha=m.alpha()[::113].tolist(), mu=mu.tolist(), var=var.tolist(), refits=m.info(2))
print("RESULT" + json.dumps(out))
''' % ROOT
    variants = {
        "chained": dict(BOHIP_CHOL_DATAFLOW="0"),
        "form1": dict(BOHIP_CHOL_DATAFLOW="2", BOHIP_CHOL_DF2_MIN="999", BOHIP_CHOL_EXEC="0"),
        "form2": dict(BOHIP_CHOL_DATAFLOW="2", BOHIP_CHOL_DF2_MIN="4", BOHIP_CHOL_DF2_LL="0", BOHIP_CHOL_EXEC="0"),
        "form2-left-looking": dict(BOHIP_CHOL_DATAFLOW="2", BOHIP_CHOL_DF2_MIN="4", BOHIP_CHOL_DF2_LL="1", BOHIP_CHOL_EXEC="0"),
        # csrc/kernels_exec.hip: the chain + one persistent task-executor kernel (the default from 47 row tiles on)
        "executor": dict(BOHIP_CHOL_DATAFLOW="2", BOHIP_CHOL_EXEC="1", BOHIP_CHOL_EXEC_MIN="4"),
        # the same without the inverse queues (W = L^-1 level by level after the factorisation), and with short pieces
        "executor-inverse-after": dict(BOHIP_CHOL_DATAFLOW="2", BOHIP_CHOL_EXEC="1", BOHIP_CHOL_EXEC_MIN="4", BOHIP_CHOL_INV_G="0"),
        "executor-inverse-pieces-of-2": dict(BOHIP_CHOL_DATAFLOW="2", BOHIP_CHOL_EXEC="1", BOHIP_CHOL_EXEC_MIN="4", BOHIP_CHOL_INV_G="2"),
        # the inverse queues in their GROUP form (the default from 28 row tiles on) forced at every size, groups of 8 and of 3 blocks
        "executor-inverse-groups": dict(BOHIP_CHOL_DATAFLOW="2", BOHIP_CHOL_EXEC="1", BOHIP_CHOL_EXEC_MIN="4", BOHIP_CHOL_INV_GRP_MIN="0"),
        "executor-inverse-groups-of-3": dict(BOHIP_CHOL_DATAFLOW="2", BOHIP_CHOL_EXEC="1", BOHIP_CHOL_EXEC_MIN="4", BOHIP_CHOL_INV_GRP_MIN="0",
                                             BOHIP_CHOL_INV_G="3"),
    }
    res = {}
    for name, env in variants.items():
        o = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BOHIP_CHOL_DF_STRICT="1", **env), capture_output=True,
                           text=True, timeout=600)
        assert o.returncode == 0, (name, o.stderr[-2000:])
        res[name] = json.loads([l for l in o.stdout.splitlines() if l.startswith("RESULT")][-1][6:])
    for name in variants:
        if name == "chained":
            continue
        for N in res["chained"]:
            a, b = res["chained"][N], res[name][N]
            assert a["Lsum"] == pytest.approx(b["Lsum"], rel=1e-12), (name, N)
            for k in ("Ldiag", "Llast", "alpha", "mu"):
                np.testing.assert_allclose(b[k], a[k], rtol=1e-9, atol=1e-12, err_msg=f"{name} N={N} {k}")
            np.testing.assert_allclose(b["var"], a["var"], rtol=1e-7, atol=1e-12, err_msg=f"{name} N={N} var")
