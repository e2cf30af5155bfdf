"""BASELINE.json configs[2] and configs[4] at FULL size, and the SURVEY.md 8-D1 stress variant at configs[1] size,
through the C ABI on one MI355X.  The 8-GPU partition is exercised as 8 logical shards behind a one-rank RCCL
communicator (bohip_mgp_*, shards_per_device = 8): same partition, same exchange, same reduction kernel.
Oracle = oracle/gp_oracle.c at full N: on ALL candidates for configs[2] (arg-max asserted against the oracle's own), on a
bounded sample where a full comparison is out of a test's reach (the stress variant's factor, Thompson's S x R draws)."""
import math
import os

import numpy as np
import pytest

from conftest import synth, var_tol
from test_parity_gpu import bohip, check_scores, make_model, mu_floor  # noqa: F401
from test_multigpu_gpu import make_multi

pytestmark = pytest.mark.gpu


def np_thompson_normal(seed, s, j):
    """NumPy twin of bohip_thompson_normal (csrc/kernels_score.hip: splitmix64 keyed on (seed, s, j) + Box-Muller);
    checked against the library's host export below.  s: scalar, j: int64 array."""
    M = np.uint64

    def sm(x):
        x = x + M(0x9E3779B97F4A7C15)
        x = (x ^ (x >> M(30))) * M(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> M(27))) * M(0x94D049BB133111EB)
        return x ^ (x >> M(31))

    with np.errstate(over="ignore"):
        key = np.full(j.shape, s, dtype=np.uint64) * M(0xD1B54A32D192ED03) + j.astype(np.uint64)
        h = sm(np.full(j.shape, seed, dtype=np.uint64) ^ sm(key))
        h2 = sm(h)
    u1 = ((h >> M(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740993.0)
    u2 = (h2 >> M(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)


def test_numpy_twin_of_the_generator(bohip):
    from bohip import _lib

    lib = _lib.load()
    j = np.arange(1000, 1400, dtype=np.int64)
    for seed, s in [(42, 0), (7, 1023), (2 ** 63 + 5, 17)]:
        ref = np.array([lib.bohip_thompson_normal(seed, s, int(v)) for v in j])
        np.testing.assert_allclose(np_thompson_normal(seed, s, j), ref, rtol=0, atol=4e-15)


def test_full_size_c3_sharded_x8(bohip, orc):
    """configs[2]: N=3000, d=8, R=32768 restarts sharded x8.  Bit-identical to the unsharded call; oracle sample incl. the winner."""
    N, d, R = 3000, 8, 32768
    X, y, Xs = synth(N, d, R, seed=0)
    ll = np.full(d, math.log(0.5))
    one = make_model(bohip, X, y, ll, 0.0, -2.0, 0.0)
    tau = float(y.max())
    sc, bv, bi = one.score("EI", [tau], Xs.T)
    assert np.all(np.isfinite(sc)) and bi == int(np.argmax(sc)) and bv == sc[bi]
    mg = make_multi(bohip, X, y, ll, 8)
    scg, bvg, big = mg.score("EI", [tau], Xs.T)
    np.testing.assert_array_equal(scg, sc)
    assert (bvg, big) == (bv, bi)
    mg.set_candidates(Xs.T)
    assert mg.score_resident("EI", [tau]) == (bv, bi)
    # the one-handle shards with the batch hint (what a rank of the one-process-per-GPU form computes) agree too
    from bohip.dist import reduce_best, shard_bounds

    one.set_batch_hint(R)
    recs = []
    for g in range(8):
        lo, hi = shard_bounds(R, 8, g)
        s_g, v_g, i_g = one.score("EI", [tau], Xs[lo:hi].T)
        np.testing.assert_array_equal(s_g, sc[lo:hi])
        recs.append((v_g, i_g + lo))
    one.set_batch_hint(0)
    assert reduce_best(*zip(*recs)) == (bv, bi)
    # the oracle at full N on ALL 32768 candidates (every host core: ~9 s on the GPU box's 128 threads): scores, mu, sigma^2 within
    # tolerance everywhere, and the winner of the device (sharded and unsharded) IS the first arg-max of the oracle's scores
    mu, var = one.predict_f(Xs.T)
    L, alpha = orc.fit(X, y, ll, 0.0, -2.0, 0.0)
    nth = min(128, os.cpu_count() or 8)
    sc_o, bv_o, bi_o = orc.score(X, ll, 0.0, 0.0, L, alpha, "EI", [tau], Xs, nthreads=nth)
    check_scores(sc, sc_o, mu_floor(alpha, 1.0) + 1e-13)
    assert bi == bi_o == int(np.argmax(sc_o)), (bi, bi_o, np.sort(sc_o)[-3:])
    assert abs(bv - bv_o) <= 1e-6 * abs(bv_o)
    mu_o, var_o = orc.predict(X, ll, 0.0, 0.0, L, alpha, Xs, nthreads=nth)
    assert np.all(np.abs(var - var_o) <= var_tol(var_o, N, 1.0))
    assert np.all(np.abs(mu - mu_o) <= 1e-6 * np.abs(mu_o) + mu_floor(alpha, 1.0))


def test_full_size_c5_thompson_1024_draws_x_65536_candidates(bohip, orc):
    """configs[4], with the per-draw reference arg-max built FROM THE DEVICE'S OWN mu / sigma^2 (not from the oracle's posterior: those are
    pinned against the oracle at R = 32768 by the C3 test above and at all 4096 / 640 candidates by the C2 / C4 tests; 65536 candidates at
    N = 3000 would be another minute of oracle).  ThompsonSamplingSimple, S=1024 draws x R=65536 candidates, d=8, N=3000, 8 shards:
    winners of ALL draws against the arg-max over mu_j + sigma_j z_sj with the generator's NumPy twin, a slice of the draws through the C
    oracle's own arg-max rule, shard invariance (8 logical shards through the RCCL exchange) bit for bit."""
    N, d, R, S, seed = 3000, 8, 65536, 1024, 20260928
    X, y, Xs = synth(N, d, R, seed=4)
    ll = np.full(d, math.log(0.5))
    one = make_model(bohip, X, y, ll, 0.0, -2.0, 0.0)
    bv, bi = one.thompson(Xs.T, S, seed=seed)
    assert np.all(bi >= 0) and np.all(bi < R) and np.all(np.isfinite(bv))
    mu, var = one.predict_f(Xs.T)
    sd = np.sqrt(var)
    j = np.arange(R, dtype=np.int64)
    bi_o = np.empty(S, dtype=np.int64)
    bv_o = np.empty(S)
    for s in range(S):
        f = mu + sd * np_thompson_normal(seed, s, j)
        bi_o[s] = int(np.argmax(f))                                      # first maximum (src/acquisition.jl:62)
        bv_o[s] = f[bi_o[s]]
    np.testing.assert_array_equal(bi, bi_o)
    np.testing.assert_allclose(bv, bv_o, rtol=1e-12)
    # the C oracle's own arg-max rule on explicit z for a slice of the draws (oracle/gp_oracle.c oracle_thompson)
    z = np.stack([np_thompson_normal(seed, s, j) for s in range(8)])
    bv_c, bi_c = orc.thompson(mu, var, z)
    np.testing.assert_array_equal(bi_c, bi[:8])
    # 8 shards: bit-identical values and GLOBAL indices
    mg = make_multi(bohip, X, y, ll, 8)
    bvg, big = mg.thompson(Xs.T, S, seed=seed)
    np.testing.assert_array_equal(big, bi)
    np.testing.assert_array_equal(bvg, bv)
    # draws spread over the candidate set (a stuck generator or a broken shard offset would collapse them)
    assert len(np.unique(bi)) >= 8 and len(np.unique(bi // (R // 8))) >= 4             # winners come from several shards


def test_stress_variant_readme_kernel_at_c2_scale(bohip, orc):
    """SURVEY.md 8-D1 stress variant: l_sigma = 5 (sigma_f^2 = e^10), logNoise = 0, every position observed 5 times
    (repetitions = 5, README.md:38-39) at N = 3000, R = 4096: kappa(cK) ~ 5 e^10 -- the case that can break the explicit
    inverse W = L^-1.  Against the oracle (substitution on L) within the documented floors."""
    d, P, reps, R = 8, 600, 5, 4096
    rng = np.random.default_rng(77)
    Xp = rng.random((P, d))
    X = np.repeat(Xp, reps, axis=0)                                      # duplicated columns, adjacent like the BO loop appends them
    N = len(X)
    y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    Xs = rng.random((R, d))
    Xs[:64] = Xp[:64] + 1e-3 * rng.standard_normal((64, d))             # candidates next to observations: sigma^2 cancels hardest
    ll = np.full(d, math.log(0.5))
    lsig, lnoise, beta = 5.0, 0.0, 0.0
    s2f = math.exp(2 * lsig)
    m = make_model(bohip, X, y, ll, lsig, lnoise, beta)
    L, alpha = orc.fit(X, y, ll, lsig, lnoise, beta)
    Lg = m.factor()
    np.testing.assert_allclose(Lg, L, rtol=1e-9, atol=1e-9 * math.sqrt(s2f))
    np.testing.assert_allclose(m.alpha(), alpha, rtol=1e-6, atol=1e-9 * np.abs(alpha).max())
    tau = float(y.max())
    for acq, p in [("EI", [tau]), ("UCB", [10.152008469453344])]:
        sc, bv, bi = m.score(acq, p, Xs.T)
        sel = np.unique(np.concatenate([[bi], np.arange(64), rng.choice(R, 63, replace=False)]))
        sc_o, _, _ = orc.score(X, ll, lsig, beta, L, alpha, acq, p, Xs[sel], nthreads=8)
        mu_o, var_o = orc.predict(X, ll, lsig, beta, L, alpha, Xs[sel], nthreads=8)
        mu, var = m.predict_f(Xs[sel].T)
        assert np.all(var >= 0) and np.all(var <= s2f * (1 + 1e-12))
        assert np.all(np.abs(var - var_o) <= var_tol(var_o, N, s2f)), np.abs(var - var_o).max()
        assert np.all(np.abs(mu - mu_o) <= 1e-6 * np.abs(mu_o) + mu_floor(alpha, s2f))
        # scores: mu's floor, plus sigma^2's floor carried through d(score)/d(sigma^2) <= (1 + beta_t) / (2 sigma)
        floor = mu_floor(alpha, s2f) + (1 + abs(p[0]) if acq == "UCB" else 1.0) * var_tol(var_o, N, s2f, rel=0.0) / (
            2 * np.sqrt(np.maximum(var_o, var_tol(var_o, N, s2f, rel=0.0))))
        assert np.all(np.abs(sc[sel] - sc_o) <= 1e-6 * np.abs(sc_o) + floor), (acq, np.abs(sc[sel] - sc_o).max())
        assert sel[int(np.argmax(sc_o))] == bi
