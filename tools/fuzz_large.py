"""Fewer, larger random cases: N up to 4000 (appends across tile boundaries), d up to 64, R across the candidate-chunk
boundary (8192) and the row-wise/MFMA threshold, checked against the oracle on a random subset of the candidates."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, bohip
from oracle.oracle import COracle
from conftest import synth, var_tol
orc = COracle(); EPS = np.finfo(float).eps
rng = np.random.default_rng(int(os.environ.get("SEED", 0)))
fails = 0; t0 = time.time()
for case in range(int(os.environ.get("CASES", 6))):
    N = int(rng.choice([900, 1537, 2049, 3000, 4000])); d = int(rng.choice([2, 8, 33, 64]))
    R = int(rng.choice([97, 1000, 8191, 8193, 20000]))
    kern = str(rng.choice(["SEArd", "Mat52Ard"]))
    ll = rng.normal(0.3 if d > 16 else -0.6, 0.2, d); lsig = 0.1; lnoise = -1.5; beta = 0.1
    X, y, _ = synth(N, d, 4, seed=int(rng.integers(1 << 30)))
    K = {"SEArd": bohip.SEArd, "Mat52Ard": bohip.Mat52Ard}[kern]
    m = bohip.ElasticGPE(d, mean=bohip.MeanConst(beta), kernel=K(ll, lsig), logNoise=lnoise, capacity=1000)
    pos = 0
    while pos < N:
        p = int(min(N - pos, rng.choice([1, 31, 32, 500, 1500]))); m.append_(X[pos:pos + p].T, y[pos:pos + p]); pos += p
    L, alpha = orc.fit(X, y, ll, lsig, lnoise, beta, kern=kern)
    Xs = rng.random((R, d)); sub = rng.choice(R, size=min(R, 300), replace=False)
    s2f = math.exp(2 * lsig); fl = 64 * EPS * s2f * np.abs(alpha).sum()
    tau = float(y.max())
    sc, bv, bi = m.score("EI", [tau], Xs.T)
    scg, g = m.score_grad("EI", [tau], Xs.T)
    mu, var = m.predict_f(Xs.T)
    sc_o, g_o = orc.score_grad(X, ll, lsig, beta, L, alpha, "EI", [tau], Xs[sub], kern=kern)
    mu_o, var_o = orc.predict(X, ll, lsig, beta, L, alpha, Xs[sub], kern=kern, nthreads=8)
    vt = var_tol(var_o, N, s2f)
    ok = (np.all(np.abs(mu[sub] - mu_o) <= 1e-6 * np.abs(mu_o) + fl) and np.all(np.abs(var[sub] - var_o) <= vt)
          and np.all(np.abs(sc[sub] - sc_o) <= 1e-6 * np.abs(sc_o) + fl + var_tol(var_o, N, s2f, rel=0) + 1e-15)
          and np.array_equal(sc, scg) and bi == int(np.argmax(sc)) and bv == sc[bi])
    good = var_o > 1e3 * vt
    if good.any():
        ok = ok and np.allclose(g.T[sub][good], g_o[good], rtol=1e-5, atol=1e-7 * (np.abs(g_o[good]).max() + 1e-300))
    Lg = m.factor()
    ok = ok and np.allclose(Lg, L, rtol=1e-8, atol=1e-10)
    print(dict(N=N, d=d, R=R, kern=kern), "ok" if ok else "MISMATCH", f"{time.time() - t0:.0f}s"); sys.stdout.flush()
    fails += not ok
print("failures:", fails); sys.exit(1 if fails else 0)
