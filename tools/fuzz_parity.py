"""Randomised parity sweep against the oracle around every tile / chunk / path boundary (run on the GPU box).
Not part of the test suite (minutes); failures are turned into regression tests."""
import sys, os, math, itertools, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, bohip
from oracle.oracle import COracle
from conftest import synth, var_tol
orc = COracle()
EPS = np.finfo(float).eps
Ns = [1, 2, 63, 127, 128, 129, 255, 256, 257, 383, 384, 385, 640, 1000, 1153, 1600]
Rs = [1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 200, 257, 300, 513, 700]
rng = np.random.default_rng(int(os.environ.get("SEED", 0)))
fails = 0; cases = 0
t0 = time.time()
for N in Ns:
    for trial in range(3):
        d = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17]))
        kern = str(rng.choice(["SEArd", "SEIso", "Mat52Ard"]))
        nl = 1 if kern == "SEIso" else d
        ll = rng.normal(-0.6, 0.3, nl); lsig = float(rng.normal(0.2, 0.3)); lnoise = float(rng.uniform(-2.5, -0.5)); beta = float(rng.normal(0, 0.3))
        X, y, _ = synth(N, d, 4, seed=int(rng.integers(1 << 30)))
        llp = ll if nl > 1 else float(ll[0])
        K = {"SEArd": bohip.SEArd, "SEIso": bohip.SEIso, "Mat52Ard": bohip.Mat52Ard}[kern]
        m = bohip.ElasticGPE(d, mean=bohip.MeanConst(beta), kernel=K(llp, lsig), logNoise=lnoise, capacity=max(N // 2, 1))
        # build in random pieces: exercises capacity growth, incremental append (p <= 32) and full refits
        pos = 0
        while pos < N:
            p = int(min(N - pos, rng.choice([1, 3, 32, 33, 200])))
            m.append_(X[pos:pos + p].T, y[pos:pos + p]); pos += p
        L, alpha = orc.fit(X, y, llp, lsig, lnoise, beta, kern=kern)
        s2f = math.exp(2 * lsig)
        Lg = m.factor()
        if not np.allclose(Lg, L, rtol=1e-8, atol=1e-10 * math.sqrt(s2f)):
            print("FACTOR MISMATCH", N, d, kern, np.abs(Lg - L).max()); fails += 1
        for R in rng.choice(Rs, size=4, replace=False):
            R = int(R)
            Xs = rng.random((R, d))
            acq, p = [("EI", [float(y.max())]), ("UCB", [2.0]), ("PI", [float(y.max())]), ("MI", [1.0, 0.3]), ("MaxMean", [])][int(rng.integers(5))]
            sc_o, g_o = orc.score_grad(X, llp, lsig, beta, L, alpha, acq, p, Xs, kern=kern)
            mu_o, var_o = orc.predict(X, llp, lsig, beta, L, alpha, Xs, kern=kern)
            sc, g = m.score_grad(acq, p, Xs.T)
            sc2, bv, bi = m.score(acq, p, Xs.T)
            mu, var = m.predict_f(Xs.T)
            fl = 64 * EPS * s2f * np.abs(alpha).sum()
            vt = var_tol(var_o, N, s2f)
            amp = max(1.0, abs(p[0])) if acq in ("UCB", "MI") else 1.0
            sfl = fl + amp * np.sqrt(var_tol(var_o, N, s2f, rel=0)) if acq in ("UCB", "MI") else fl + var_tol(var_o, N, s2f, rel=0) + 1e-15
            ok = (np.all(np.abs(mu - mu_o) <= 1e-6 * np.abs(mu_o) + fl) and np.all(np.abs(var - var_o) <= vt)
                  and np.all(np.abs(sc - sc_o) <= 1e-6 * np.abs(sc_o) + sfl) and np.array_equal(sc, sc2)
                  and (bi < 0 or sc2[bi] == bv) and (bi == int(np.argmax(np.where(np.isnan(sc2), -np.inf, sc2))) if np.isfinite(sc2).any() else True))
            # gradients: where the variance is not in its cancellation floor
            good = var_o > 1e3 * vt
            if good.any():
                gs = np.abs(g_o[good]).max() + 1e-300
                ok = ok and np.allclose(g.T[good], g_o[good], rtol=1e-5, atol=1e-7 * gs)
            cases += 1
            if not ok:
                fails += 1
                print("MISMATCH", dict(N=N, d=d, kern=kern, R=R, acq=acq), "mu", np.abs(mu - mu_o).max(), "var", np.abs(var - var_o).max(),
                      "sc", np.abs(sc - sc_o).max(), "eq", np.array_equal(sc, sc2), "bi", bi)
print(f"{cases} cases, {fails} failures, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
