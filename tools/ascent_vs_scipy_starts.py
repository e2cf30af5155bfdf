"""Per start: the device search's end value against SciPy's L-BFGS-B on the oracle from the same start (headline model, UCB at beta_t, ten LHS
starts per seed).  usage: python tools/ascent_vs_scipy_starts.py [seed ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth, lhs, DIM, BETA_T
from oracle.oracle import COracle
from scipy.optimize import minimize
import bohip

orc = COracle()
X, y = synth(0)
ll = np.full(DIM, np.log(0.5))
L, alpha = orc.fit(X, y, ll, 0.0, -2.0, 0.0)
m = bohip.ElasticGPE(DIM, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(ll, 0.0), logNoise=-2.0, capacity=len(y))
m.append_(X.T, y)
lb, ub = np.zeros(DIM), np.ones(DIM)
tot = below = 0
for seed in [int(a) for a in sys.argv[1:]] or [7]:
    starts = np.asfortranarray(lhs(10, seed=seed).T)
    fd, Xd, bf, bi, bx, ev = m.ascend("UCB", [BETA_T], lb, ub, starts, 2000)
    fs, nf = [], []
    for r in range(10):
        cnt = [0]
        def negfg(x):
            cnt[0] += 1
            sc, g = orc.score_grad(X, ll, 0.0, 0.0, L, alpha, "UCB", [BETA_T], x[None, :].copy())
            return -float(sc[0]), -g[0]
        res = minimize(negfg, starts[:, r], jac=True, method="L-BFGS-B", bounds=[(0, 1)] * DIM, options=dict(maxiter=2000, ftol=1e-10, gtol=1e-10))
        fs.append(-res.fun); nf.append(cnt[0])
    fs = np.array(fs)
    sc_o, g_o = orc.score_grad(X, ll, 0.0, 0.0, L, alpha, "UCB", [BETA_T], np.ascontiguousarray(Xd.T))
    pg = np.where(((Xd.T <= 0) & (g_o < 0)) | ((Xd.T >= 1) & (g_o > 0)), 0.0, g_o)
    ge = fd >= fs - 1e-6 * np.abs(fs)
    tot += 10; below += int((~ge).sum())
    print(f"seed {seed}: device passes {ev}, scipy max evals {max(nf)}; device best {bf:.6f} scipy best {fs.max():.6f} rel {bf / fs.max() - 1:+.2e}; device >= scipy for {int(ge.sum())} of 10 starts")
    print("   device", np.array2string(fd, precision=4), "\n   scipy ", np.array2string(fs, precision=4), "\n   |proj grad| at device ends", np.array2string(np.abs(pg).max(1), precision=2))
print(f"{below} of {tot} starts end below SciPy's value from the same start")
