"""Independent published-algorithm check of the oracle against scikit-learn's GaussianProcessRegressor (CPU).

What this bounds and what it does not.  The reference's GP arithmetic lives in GaussianProcesses.jl / ElasticPDMats.jl,
which are neither vendored under /root/reference nor runnable here (no Julia), so parity with THAT package stays unpinned
(DESIGN.md section 7).  scikit-learn is a separate implementation of the same published algorithm (Rasmussen & Williams,
GPML, Algorithm 2.1: L = chol(K + s_n^2 I), alpha = L'\\(L\\y), mu = k*'alpha, v = L\\k*, var = k** - v'v,
log p(y|X) = -1/2 y'alpha - sum log L_ii - n/2 log 2 pi).  Agreement here says the oracle restates the ALGORITHM the
reference's call sites (src/models/gp.jl:2-18) delegate to, with the kernel conventions of README.md:24-25
(SEArd(ll, lsig): k = exp(2 lsig) exp(-1/2 sum ((x-x')/exp(ll))^2); Mat52Ard likewise) -- not that the Julia package
rounds the same way.

Mapping:  SEArd(ll, lsig)   -> ConstantKernel(exp(2 lsig)) * RBF(length_scale=exp(ll))
          Mat52Ard(ll, lsig) -> ConstantKernel(exp(2 lsig)) * Matern(length_scale=exp(ll), nu=2.5)
          logNoise           -> WhiteKernel(exp(2 logNoise) + eps)   (the +eps is ORACLE_NOISE_EPS, see gp_oracle.c)
          MeanConst(beta)    -> fit on y - beta, add beta back to the prediction
scikit-learn's predictive variance of a kernel with a WhiteKernel term includes the noise on k**; the reference's
`predict_f` is the LATENT variance, so the white-noise level is subtracted before comparing.
"""
import math
import warnings

import numpy as np
import pytest

from conftest import load_golden, synth
from oracle.oracle import NOISE_EPS

sk_gp = pytest.importorskip("sklearn.gaussian_process")
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern, WhiteKernel  # noqa: E402

GOLDEN_CASES = ["n1_seiso_maxmean", "n2_seard", "readme_d2_rep5", "branin_shaped", "n256_d8_r512", "ties_n256"]


def sk_fit(X, y, loglen, logsig, lognoise, beta, kern="SEArd"):
    d = X.shape[1]
    ell = np.exp(np.broadcast_to(np.asarray(loglen, dtype=np.float64), (d,)))
    noise = math.exp(2.0 * lognoise) + NOISE_EPS
    base = Matern(length_scale=ell, nu=2.5) if kern == "Mat52Ard" else RBF(length_scale=ell)
    k = ConstantKernel(math.exp(2.0 * logsig)) * base + WhiteKernel(noise)
    gpr = sk_gp.GaussianProcessRegressor(kernel=k, alpha=0.0, optimizer=None, normalize_y=False, copy_X_train=True)
    gpr.fit(X, y - beta)
    return gpr, noise


def check_against_sklearn(orc, X, y, loglen, logsig, lognoise, beta, Xs, kern="SEArd", rtol=1e-9):
    N = X.shape[0]
    s2f = math.exp(2.0 * logsig)
    L, alpha = orc.fit(X, y, loglen, logsig, lognoise, beta, kern=kern)
    mu, var = orc.predict(X, loglen, logsig, beta, L, alpha, Xs, kern=kern)
    gpr, noise = sk_fit(X, y, loglen, logsig, lognoise, beta, kern)
    # --- Cholesky factor and alpha
    scale_L = np.abs(gpr.L_).max()
    assert np.abs(L - gpr.L_).max() <= rtol * scale_L * max(1.0, np.sqrt(N)), "L"
    cond_slack = max(1.0, (np.abs(np.diag(L)).max() / np.abs(np.diag(L)).min()) ** 2)
    assert np.abs(alpha - gpr.alpha_).max() <= rtol * cond_slack * max(np.abs(gpr.alpha_).max(), 1e-300), "alpha"
    # --- posterior mean and LATENT variance
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")   # sklearn warns when it clamps a (rounding-)negative predictive variance
        mu_s, std_s = gpr.predict(Xs, return_std=True)
    mu_s = mu_s + beta
    var_s = std_s ** 2 - noise
    mu_floor = 64 * np.finfo(float).eps * s2f * np.abs(alpha).sum()
    assert np.all(np.abs(mu - mu_s) <= 1e-9 * np.abs(mu_s) + mu_floor), "mu"
    # sigma^2 = s_f^2 - v'v cancels near observations (and sklearn carries the noise term through the subtraction):
    # relative bound + the float64 floor of the cancellation
    var_floor = 64 * N * np.finfo(float).eps * (s2f + noise)
    assert np.all(np.abs(var - np.maximum(var_s, 0.0)) <= 1e-9 * np.abs(var_s) + var_floor), "var"
    # --- log marginal likelihood
    mll, _ = orc.mll_grad(X, y, loglen, logsig, lognoise, beta, kern=kern)
    mll_s = gpr.log_marginal_likelihood_value_
    assert abs(mll - mll_s) <= 1e-9 * abs(mll_s) + 1e-9 * N, "mll"
    return gpr


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_golden_case_vs_sklearn(orc, name):
    g = load_golden(name)
    X, y, Xs = g["X"], g["y"], g["Xs"]
    check_against_sklearn(orc, X, y, g["loglen"], float(g["logsig"]), float(g["lognoise"]), float(g["beta"]), Xs)
    # and the committed vectors themselves (they were written by the same oracle: this pins the FILES to sklearn too)
    gpr, noise = sk_fit(X, y, g["loglen"], float(g["logsig"]), float(g["lognoise"]), float(g["beta"]))
    s2f = math.exp(2.0 * float(g["logsig"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mu_s, std_s = gpr.predict(Xs, return_std=True)
    N = X.shape[0]
    assert np.all(np.abs(g["mu"] - (mu_s + float(g["beta"]))) <= 1e-9 * np.abs(mu_s) + 64 * np.finfo(float).eps * s2f * np.abs(g["alpha"]).sum())
    assert np.all(np.abs(g["var"] - np.maximum(std_s ** 2 - noise, 0.0)) <= 1e-9 * np.abs(std_s ** 2) + 64 * N * np.finfo(float).eps * (s2f + noise))
    assert np.abs(g["Ldiag"] - np.diag(gpr.L_)).max() <= 1e-9 * np.abs(np.diag(gpr.L_)).max()
    assert np.abs(g["Lrow_last"] - gpr.L_[-1]).max() <= 1e-9 * np.abs(gpr.L_).max() * max(1.0, np.sqrt(N))


def test_c2_shaped_sample_vs_sklearn(orc):
    """BASELINE configs[1] shape (N=3000, d=8) with the synthetic recipe of BASELINE.md; 64 candidates."""
    X, y, Xs = synth(3000, 8, 64, seed=0)
    check_against_sklearn(orc, X, y, np.full(8, math.log(0.5)), 0.0, -2.0, 0.0, Xs)


def test_matern52_vs_sklearn(orc):
    """Mat52Ard is the default model kernel of `optimize` (src/BayesianOptimization.jl:259-262)."""
    X, y, Xs = synth(200, 3, 40, seed=3)
    check_against_sklearn(orc, X, y, np.array([0.2, -0.3, 0.1]), 0.4, -1.5, 0.25, Xs, kern="Mat52Ard")


def test_stress_variant_vs_sklearn(orc):
    """README kernel (lsig = 5, logNoise = 0) with every position observed five times: kappa ~ 5e10."""
    rng = np.random.default_rng(5)
    pos = rng.random((60, 2)) * 10 - 5
    X = np.repeat(pos, 5, axis=0)
    y = -(((X - 1) ** 2).sum(1) + rng.standard_normal(len(X)))
    Xs = rng.random((32, 2)) * 10 - 5
    # alpha = K^-1 y inherits the conditioning: compare it through K alpha = y instead of entry by entry
    L, alpha = orc.fit(X, y, [0.0, 0.0], 5.0, 0.0, 0.0)
    gpr, noise = sk_fit(X, y, [0.0, 0.0], 5.0, 0.0, 0.0)
    K = gpr.L_ @ gpr.L_.T
    assert np.abs(K @ alpha - y).max() <= 1e-6 * np.abs(y).max()
    assert np.abs(L - gpr.L_).max() <= 1e-7 * np.abs(gpr.L_).max()
    mu, var = orc.predict(X, [0.0, 0.0], 5.0, 0.0, L, alpha, Xs)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mu_s, std_s = gpr.predict(Xs, return_std=True)
    s2f = math.exp(10.0)
    assert np.all(np.abs(mu - mu_s) <= 1e-6 * np.abs(mu_s) + 1e-6)
    assert np.all(np.abs(var - np.maximum(std_s ** 2 - noise, 0.0)) <= 1e-6 * np.abs(var) + 64 * len(X) * np.finfo(float).eps * s2f * 50)
