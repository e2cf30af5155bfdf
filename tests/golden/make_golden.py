"""Generates the committed golden fixtures in this directory.

The reference (Julia) cannot run in the build image and holds no numeric fixtures of its own
(SURVEY.md 8c), so these vectors come from the project's own float64 oracle (oracle/gp_oracle.c),
cross-checked here against the independent NumPy/LAPACK restatement and, for the small cases, an
mpmath 60-digit evaluation before anything is written.  PARITY UNPINNED at the GaussianProcesses.jl
boundary -- see oracle/gp_oracle.c.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle import COracle, NumpyGP, latin_hypercube, mp_predict, mp_acq, brochu_beta  # noqa: E402

orc = COracle()


def gp_case(name, X, y, ll, lsig, lnoise, beta, Xs, acqs, mp_check=False):
    d = X.shape[1]
    ll = np.broadcast_to(np.asarray(ll, dtype=np.float64), (d,)).copy()
    L, alpha = orc.fit(X, y, ll, lsig, lnoise, beta)
    mu, var = orc.predict(X, ll, lsig, beta, L, alpha, Xs)
    ngp = NumpyGP(d, ll, lsig, lnoise, beta).fit(X, y)
    mu_n, var_n = ngp.predict_f(Xs)
    s2f = np.exp(2 * lsig)
    floor = 64 * X.shape[0] * np.finfo(float).eps * s2f
    assert np.allclose(mu, mu_n, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(alpha).sum() * s2f * 1e-6)), name
    assert np.all(np.abs(var - var_n) <= 1e-8 * np.abs(var_n) + floor), name
    if mp_check:
        mus, vars_ = mp_predict(X, y, ll, lsig, lnoise, beta, Xs[:8])
        for i in range(len(mus)):
            assert abs(float(mus[i]) - mu[i]) <= 1e-9 * max(1.0, abs(mu[i])), (name, i)
            assert abs(float(vars_[i]) - var[i]) <= 1e-8 * abs(var[i]) + floor, (name, i)
    out = dict(X=X, y=y, loglen=ll, logsig=lsig, lognoise=lnoise, beta=beta, Xs=Xs, mu=mu, var=var,
               Ldiag=np.diag(L).copy(), Lrow_last=L[-1].copy(), alpha=alpha)
    for acq, params in acqs.items():
        sc, bv, bi = orc.score(X, ll, lsig, beta, L, alpha, acq, params, Xs)
        out[f"{acq}_params"] = np.asarray(params, dtype=np.float64)
        out[f"{acq}_score"] = sc
        out[f"{acq}_best"] = np.array([bv])
        out[f"{acq}_best_idx"] = np.array([bi], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "N", X.shape[0], "R", Xs.shape[0], "var range", var.min(), var.max())


def main():
    rng = np.random.default_rng(0)
    # (1) analytic N=1 (reference test/acquisition.jl:2,11-12: GPE([1.0],[2.0],MeanZero(),SEIso(1.0,0.0)))
    X = np.array([[1.0]]); y = np.array([2.0])
    Xs = np.linspace(-5, 5, 101).reshape(-1, 1)
    gp_case("n1_seiso_maxmean", X, y, [1.0], 0.0, -2.0, 0.0, Xs, {"MaxMean": [], "EI": [2.0], "UCB": [1.0]}, mp_check=True)
    # (2) N=2 closed form territory
    X = np.array([[0.0, 0.0], [1.0, 0.5]]); y = np.array([0.3, -0.2])
    Xs = rng.random((32, 2)) * 2 - 0.5
    gp_case("n2_seard", X, y, [0.1, -0.2], 0.3, -1.0, 0.1, Xs, {"EI": [0.3], "PI": [0.3], "UCB": [brochu_beta(2, 2)], "MI": [1.0, 0.25]}, mp_check=True)
    # (3) README-shaped: d=2, SEArd([0,0],5.), MeanConst, logNoise=0, every position observed 5 times
    pos = rng.random((40, 2)) * 10 - 5
    X = np.repeat(pos, 5, axis=0)
    y = -(((X - 1) ** 2).sum(1) + rng.standard_normal(len(X)))
    Xs = latin_hypercube([-5, -5], [5, 5], 64, rng)
    gp_case("readme_d2_rep5", X, y, [0.0, 0.0], 5.0, 0.0, 0.0, Xs, {"UCB": [brochu_beta(2, len(X))], "EI": [y.max()]})
    # (4) branin-test-shaped hyper-parameters: SEArd([0,0],5.), MeanConst(-10), logNoise=-2 (test/branin.jl:24-26)
    X = rng.random((60, 2)) * 15 - np.array([5.0, 0.0])
    y = -np.sin(X[:, 0]) * 10 - (X[:, 1] - 5) ** 2 * 0.3
    Xs = latin_hypercube([-5, 0], [10, 15], 96, rng)
    gp_case("branin_shaped", X, y, [0.0, 0.0], 5.0, -2.0, -10.0, Xs, {"EI": [y.max()], "PI": [y.max()], "MI": [1.0, 0.7], "MaxMean": []})
    # (5) mid-size seeded case: N=256, d=8, R=512 (BASELINE.md synthetic recipe)
    X = rng.random((256, 8)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(256)
    Xs = latin_hypercube(np.zeros(8), np.ones(8), 512, rng)
    gp_case("n256_d8_r512", X, y, np.full(8, np.log(0.5)), 0.0, -2.0, 0.0, Xs,
            {"EI": [y.max()], "UCB": [brochu_beta(8, 256)], "PI": [y.max()], "MI": [1.0, 0.1], "MaxMean": []})
    # (6) argmax tie cases: duplicated candidate columns -> identical scores, first index must win
    Xs_t = np.concatenate([Xs[:100], Xs[40:41], Xs[:100]], axis=0)
    gp_case("ties_n256", X, y, np.full(8, np.log(0.5)), 0.0, -2.0, 0.0, Xs_t, {"EI": [y.max()], "MaxMean": []})
    # (7) acquisition formula sweeps incl. tails, sigma^2 == 0 branches, and the non-textbook EI
    mus = np.concatenate([np.linspace(-10, 10, 81), [0.3, 0.3, 0.5, 0.7]])
    s2s = np.concatenate([np.full(81, 1.0), [4.0, 0.0, 0.0, 0.0]])
    rows = []
    for tau in (0.0, 0.5):
        for m, s in zip(mus, s2s):
            ei = orc.acq("EI", [tau], m, s); pi = orc.acq("PI", [tau], m, s)
            assert abs(ei - float(mp_acq("EI", [tau], m, s))) <= 1e-12 * max(1.0, abs(ei)) + 1e-16
            rows.append((m, s, tau, ei, pi, orc.acq("UCB", [2.5], m, s), orc.acq("MI", [1.3, 0.4], m, s)))
    betas = np.array([[D, n, brochu_beta(D, n)] for D in (1, 2, 8, 16) for n in (0, 1, 10, 500, 3000, 10000)])
    np.savez_compressed(os.path.join(HERE, "acq_formulas.npz"), table=np.array(rows), brochu=betas,
                        ref_vs_textbook=np.array([0.3, 4.0, 0.5, orc.acq("EI", [0.5], 0.3, 4.0)]))
    print("acq_formulas", len(rows), "rows; EI(0.3,4,0.5) =", orc.acq("EI", [0.5], 0.3, 4.0))


if __name__ == "__main__":
    main()
