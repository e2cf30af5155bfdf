"""Quick GPU-vs-oracle check used while bringing kernels up (the real tests live in tests/)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bohip
from oracle.oracle import COracle

def synth(N, d, R, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    Xs = rng.random((R, d))
    return X, y, Xs

def run(N, d, R, lsig=0.0, lnoise=-2.0, beta=0.0, acq="EI"):
    X, y, Xs = synth(N, d, R)
    ll = np.full(d, np.log(0.5))
    orc = COracle()
    t0 = time.time(); L, alpha = orc.fit(X, y, ll, lsig, lnoise, beta); t_fit = time.time() - t0
    tau = y.max()
    params = [tau] if acq in ("EI", "PI") else [orc.brochu_beta(d, N)]
    t0 = time.time(); sc_o, bv_o, bi_o = orc.score(X, ll, lsig, beta, L, alpha, acq, params, Xs, nthreads=8); t_sc = time.time() - t0
    mu_o, var_o = orc.predict(X, ll, lsig, beta, L, alpha, Xs, nthreads=8)
    m = bohip.ElasticGPE(d, mean=bohip.MeanConst(beta), kernel=bohip.SEArd(ll, lsig), logNoise=lnoise, capacity=N)
    m.enable_timing()
    t0 = time.time(); m.append_(X.T, y); t_gfit = time.time() - t0
    print("  fit stages:", m.timing())
    Lg = m.factor(); ag = m.alpha()
    print(f"N={N} d={d} R={R}: oracle fit {t_fit:.2f}s score {t_sc:.2f}s | gpu fit {t_gfit*1e3:.1f} ms")
    print("  L rel err", np.abs(Lg - L).max() / np.abs(L).max(), " alpha rel err", np.abs(ag - alpha).max() / np.abs(alpha).max())
    mu_g, var_g = m.predict_f(Xs.T)
    print("  mu  max rel", (np.abs(mu_g - mu_o) / np.maximum(np.abs(mu_o), 1e-300)).max(), " abs", np.abs(mu_g - mu_o).max())
    print("  var max rel", (np.abs(var_g - var_o) / np.maximum(np.abs(var_o), 1e-300)).max(), " abs", np.abs(var_g - var_o).max(), "min var", var_o.min())
    for _ in range(2):
        t0 = time.time(); sc_g, bv_g, bi_g = m.score(acq, params, Xs.T); t_g = time.time() - t0
    print("  score stages:", m.timing())
    rel = np.abs(sc_g - sc_o) / np.maximum(np.abs(sc_o), 1e-300)
    print(f"  {acq} max rel {rel.max():.3e} abs {np.abs(sc_g - sc_o).max():.3e}; argmax gpu {bi_g} oracle {bi_o} val {bv_g} {bv_o}; host-call {t_g*1e3:.2f} ms")
    return bi_g == bi_o

if __name__ == "__main__":
    ok = True
    ok &= run(100, 2, 64)
    ok &= run(300, 3, 500, acq="UCB")
    ok &= run(1000, 8, 1024)
    ok &= run(3000, 8, 4096)
    print("ALL ARGMAX MATCH" if ok else "ARGMAX MISMATCH")
