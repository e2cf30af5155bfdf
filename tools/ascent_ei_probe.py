"""EI through the reference's default search on the headline model (N = 3000, d = 8, tau = max y, ten LHS starts, :LD_LBFGS, maxeval 2000):
(a) SciPy's L-BFGS-B on the ORACLE's value + gradient, evaluations per start; (c) the device search: passes, ms, end values.
usage: python tools/ascent_ei_probe.py [seed ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth, lhs, DIM
from oracle.oracle import COracle
from scipy.optimize import minimize

orc = COracle()
X, y = synth(0)
ll = np.full(DIM, np.log(0.5))
L, alpha = orc.fit(X, y, ll, 0.0, -2.0, 0.0)
tau = float(y.max())
lb, ub = np.zeros(DIM), np.ones(DIM)
m = None
try:
    import bohip
    m = bohip.ElasticGPE(DIM, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(ll, 0.0), logNoise=-2.0, capacity=len(y))
    m.append_(X.T, y)
except Exception as e:      # noqa: BLE001
    print("device not available:", e)
for seed in [int(a) for a in sys.argv[1:]] or [7]:
    starts = np.asfortranarray(lhs(10, seed=seed).T)
    sc0, _ = orc.score_grad(X, ll, 0.0, 0.0, L, alpha, "EI", [tau], np.ascontiguousarray(starts.T))
    nf, fs = [], []
    t0 = time.perf_counter()
    for r in range(10):
        cnt = [0]
        def negfg(x):
            cnt[0] += 1
            sc, g = orc.score_grad(X, ll, 0.0, 0.0, L, alpha, "EI", [tau], x[None, :].copy())
            return -float(sc[0]), -g[0]
        res = minimize(negfg, starts[:, r], jac=True, method="L-BFGS-B", bounds=[(0, 1)] * DIM, options=dict(maxiter=2000, maxfun=2000, ftol=1e-10, gtol=1e-10))
        nf.append(cnt[0]); fs.append(-res.fun)
    print(f"seed {seed}: start values {np.array2string(sc0, precision=3)}")
    print(f"  (a) SciPy L-BFGS-B on the oracle: evaluations per start {nf} max {max(nf)} sum {sum(nf)} ({time.perf_counter() - t0:.1f} s)")
    print("      end values", np.array2string(np.array(fs), precision=5))
    if m is not None:
        m.ascend("EI", [tau], lb, ub, starts, 2000)
        t0 = time.perf_counter()
        fd, Xd, bf, bi, bx, ev = m.ascend("EI", [tau], lb, ub, starts, 2000)
        dt = time.perf_counter() - t0
        print(f"  (c) device: passes {ev}, {dt * 1e3:.2f} ms per acquire_max, {dt / max(ev, 1) * 1e6:.1f} us per pass")
        print("      end values", np.array2string(fd, precision=5), " best", bf, " scipy best", max(fs))
