"""How often does the device ascent (bohip_gp_acquire_max, free-running driver) give a start point up at a point that is NOT a KKT point of
the oracle's objective?  The N = 600 case of tests/test_parity_gpu.py::test_device_ascent_against_scipy_lbfgsb_on_the_oracle over several
start seeds; per seed the largest |projected gradient| / max |gradient at the starts| per acquisition and the number of start points above
the test's 2e-4.  Run with BOHIP_SMALL_MFMA=0 for round 4's five-kernel pass."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, bohip
from oracle.oracle import COracle
from conftest import synth
orc = COracle()
N = 600
X, y, _ = synth(N, 3, 1, seed=40)
ll = np.array([-0.9, -0.6, -0.75]); lsig, lnoise, beta = 0.1, -2.0, 0.2
L, alpha = orc.fit(X, y, ll, lsig, lnoise, beta)
m = bohip.ElasticGPE(3, mean=bohip.MeanConst(beta), kernel=bohip.SEArd(ll, lsig), logNoise=lnoise, capacity=N); m.append_(X.T, y)
lb, ub = np.zeros(3), np.ones(3)
tau = float(np.median(y))
tot = 0; bad = 0
for seed in range(41, 41 + int(os.environ.get("SEEDS", 12))):
    starts = np.random.default_rng(seed).random((3, 16))
    line = []
    for acq, p in [("EI", [tau]), ("UCB", [2.0]), ("MaxMean", [])]:
        f, Xd, bf, bi, bx, ev = m.ascend(acq, p, lb, ub, starts, maxeval=2000, ftol_rel=1e-13, xtol_abs=1e-13)
        sc_o, g_o = orc.score_grad(X, ll, lsig, beta, L, alpha, acq, p if p else [0.0], np.ascontiguousarray(Xd.T))
        pg = np.where(((Xd.T <= 0) & (g_o < 0)) | ((Xd.T >= 1) & (g_o > 0)), 0.0, g_o)
        g0 = orc.score_grad(X, ll, lsig, beta, L, alpha, acq, p if p else [0.0], np.ascontiguousarray(starts.T))[1]
        r = np.abs(pg).max(1) / np.abs(g0).max()
        tot += 16; bad += int((r > 2e-4).sum())
        line.append(f"{acq} {ev:4d} passes, max {r.max():.1e}, above 2e-4: {int((r > 2e-4).sum())}")
    print(f"seed {seed}: " + " | ".join(line))
print(f"start points given up at a non-KKT point: {bad} of {tot}")
