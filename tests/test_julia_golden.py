"""Consumes tests/golden/julia_outputs.txt -- the outputs of julia/gen_golden.jl, i.e. of the REAL GaussianProcesses.jl /
BayesianOptimization.jl on the inputs in tests/golden/julia_inputs.txt -- when someone with a Julia toolchain has
produced it.  Present: the oracle (CPU) and the device (GPU) are compared against the reference itself and parity is
pinned.  Absent (the state of this repository: no Julia in the build image or on the GPU box): the tests report
"parity unpinned" and check that the documentation says so too."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, var_tol

sys.path.insert(0, GOLDEN)
from gio import read_cases, write_cases  # noqa: E402

INPUTS = os.path.join(GOLDEN, "julia_inputs.txt")
OUTPUTS = os.path.join(GOLDEN, "julia_outputs.txt")
HAVE = os.path.exists(OUTPUTS)
UNPINNED = ("parity unpinned: tests/golden/julia_outputs.txt is absent -- run julia/gen_golden.jl where a Julia toolchain "
            "exists and commit its output")
EPS = np.finfo(np.float64).eps


def test_inputs_file_matches_the_committed_goldens():
    c = read_cases(INPUTS)
    assert set(c) == {"n1_seiso_maxmean", "n2_seard", "readme_d2_rep5", "branin_shaped", "n256_d8_r512", "ties_n256",
                      "c2_shaped_sample"}
    for name in ("n2_seard", "n256_d8_r512", "readme_d2_rep5"):
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        for k in ("X", "y", "Xs", "loglen"):
            np.testing.assert_array_equal(c[name][k], g[k])             # text round trip is bit-exact
        assert float(c[name]["logsig"]) == float(g["logsig"])
    assert c["c2_shaped_sample"]["X"].shape == (3000, 8) and c["c2_shaped_sample"]["Xs"].shape == (64, 8)


def test_format_round_trip(tmp_path):
    a = {"k": {"s": "SEArd", "x": np.array([[1.0, -0.1], [np.pi, 1e-300]]), "z": np.float64(0.1), "e": np.zeros(0),
               "i": np.array([np.inf, -np.inf])}}
    p = tmp_path / "t.txt"
    write_cases(p, a)
    b = read_cases(p)
    assert b["k"]["s"] == "SEArd" and b["k"]["z"].shape == () and b["k"]["e"].shape == (0,)
    np.testing.assert_array_equal(b["k"]["x"], a["k"]["x"])
    np.testing.assert_array_equal(b["k"]["i"], a["k"]["i"])


def test_parity_status_is_stated_honestly():
    """while the Julia outputs are absent, DESIGN.md and the oracle header must say PARITY UNPINNED; once they exist the
    claim may be dropped"""
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    header = open(os.path.join(ROOT, "oracle", "gp_oracle.c")).read()[:6000]
    if not HAVE:
        assert "PARITY UNPINNED" in design.upper() and "PARITY UNPINNED" in header.upper()
        assert "gen_golden.jl" in design


def _oracle_case(orc, c):
    X, y, Xs = c["X"], c["y"], c["Xs"]
    d = X.shape[1]
    ll = np.broadcast_to(np.atleast_1d(c["loglen"]), (d,)).copy()
    lsig, lnoise = float(c["logsig"]), float(c["lognoise"])
    beta = 0.0 if c["mean"] == "MeanZero" else float(c["beta"])
    L, alpha = orc.fit(X, y, ll, lsig, lnoise, beta, kern=c["kern"])
    return X, y, Xs, ll, lsig, lnoise, beta, L, alpha


def _compare(name, c, j, mu, var, alpha, Ldiag, Lrow, scores, bests):
    N = len(c["y"])
    s2f = float(np.exp(2 * float(c["logsig"])))
    floor_mu = 64 * EPS * s2f * np.abs(j["alpha"]).sum()
    assert np.all(np.abs(mu - j["mu"]) <= 1e-6 * np.abs(j["mu"]) + floor_mu), name
    assert np.all(np.abs(var - j["var"]) <= var_tol(j["var"], N, s2f)), name
    np.testing.assert_allclose(alpha, j["alpha"], rtol=1e-6, atol=1e-9 * np.abs(j["alpha"]).max(), err_msg=name)
    np.testing.assert_allclose(Ldiag, j["Ldiag"], rtol=1e-9, err_msg=name)
    np.testing.assert_allclose(Lrow, j["Lrow_last"], rtol=1e-7, atol=1e-9 * np.sqrt(s2f), err_msg=name)
    for acq, sc in scores.items():
        ref = j[f"{acq}_score"]
        assert np.all(np.abs(sc - ref) <= 1e-6 * np.abs(ref) + floor_mu + 1e-13), (name, acq)
        assert int(bests[acq][1]) == int(j[f"{acq}_best_idx"][0]), (name, acq, "arg-max index must be bit-exact")


@pytest.mark.skipif(not HAVE, reason=UNPINNED)
def test_oracle_against_the_real_reference(orc):
    ins, outs = read_cases(INPUTS), read_cases(OUTPUTS)
    assert set(outs) == set(ins)
    for name, c in ins.items():
        j = outs[name]
        X, y, Xs, ll, lsig, lnoise, beta, L, alpha = _oracle_case(orc, c)
        mu, var = orc.predict(X, ll, lsig, beta, L, alpha, Xs, kern=c["kern"], nthreads=8)
        scores, bests = {}, {}
        for k in c:
            if k.endswith("_params"):
                acq = k.split("_")[0]
                p = list(np.atleast_1d(c[k]))
                scores[acq], bv, bi = orc.score(X, ll, lsig, beta, L, alpha, acq, p if p else [0.0], Xs, kern=c["kern"], nthreads=8)
                bests[acq] = (bv, bi)
        _compare(name, c, j, mu, var, alpha, np.diag(L), L[-1], scores, bests)
        # the two UPSTREAM-UNVERIFIED switches of oracle/gp_oracle.c, now answered by the reference itself
        assert abs(float(j["noise_on_diagonal"]) - (np.exp(2 * lnoise) + EPS)) <= 4 * EPS * max(1.0, float(np.exp(2 * float(c["logsig"])))), \
            (name, "ORACLE_NOISE_EPS disagrees with GaussianProcesses.jl", float(j["noise_on_diagonal"]))
        assert float(j["var_min"]) >= 0.0, (name, "predict_f does not clamp: flip ORACLE_CLAMP_VAR")


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE, reason=UNPINNED)
def test_device_against_the_real_reference():
    import bohip

    ins, outs = read_cases(INPUTS), read_cases(OUTPUTS)
    K = {"SEArd": bohip.SEArd, "SEIso": bohip.SEIso, "Mat52Ard": bohip.Mat52Ard}
    for name, c in ins.items():
        j = outs[name]
        X, y, Xs = c["X"], c["y"], c["Xs"]
        mean = bohip.MeanZero() if c["mean"] == "MeanZero" else bohip.MeanConst(float(c["beta"]))
        m = bohip.ElasticGPE(X.shape[1], mean=mean, kernel=K[c["kern"]](np.atleast_1d(c["loglen"]), float(c["logsig"])),
                             logNoise=float(c["lognoise"]), capacity=len(y))
        m.append_(X.T, y)
        mu, var = m.predict_f(Xs.T)
        L = m.factor()
        scores, bests = {}, {}
        for k in c:
            if k.endswith("_params"):
                acq = k.split("_")[0]
                scores[acq], bv, bi = m.score(acq, list(np.atleast_1d(c[k])), Xs.T)
                bests[acq] = (bv, bi)
        _compare(name, c, j, mu, var, m.alpha(), np.diag(L), L[-1], scores, bests)


def test_report_parity_status():
    if not HAVE:
        pytest.skip(UNPINNED)
