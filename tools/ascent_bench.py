"""acquire_max local search (10 restarts, :LD_LBFGS, <= 200 evaluations): device ascent vs the NumPy restatement
driving the device score_grad one call per evaluation."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bohip
from bohip.acquisition import _batched_lbfgs_ascent
rng = np.random.default_rng(0)
for N, d in ((50, 2), (200, 2), (500, 2), (3000, 8), (10000, 16)):   # (N <= 256: one launch, one workgroup per start point)
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
    m.append_(X.T, y)
    lb, ub = np.zeros(d), np.ones(d)
    for R in (10, 64):
        starts = np.asfortranarray(rng.random((d, R)))
        p = [2.0]
        m.ascend("UCB", p, lb, ub, starts, 200)
        t0 = time.perf_counter(); f, Xb, bf, bi, bx, ev = m.ascend("UCB", p, lb, ub, starts, 200); t1 = time.perf_counter() - t0
        cnt = [0]
        def fg(Z):
            cnt[0] += 1
            return m.score_grad("UCB", p, Z)
        t0 = time.perf_counter(); fh, Xh = _batched_lbfgs_ascent(fg, starts, lb, ub, 200); t2 = time.perf_counter() - t0
        print(f"N={N} d={d} R={R}: device {t1*1e3:7.2f} ms ({ev} evals, {t1/ev*1e6:5.0f} us/eval)   host-driven {t2*1e3:7.2f} ms ({cnt[0]} evals, {t2/cnt[0]*1e6:5.0f} us/eval)   best {bf:.6g} vs {fh.max():.6g}")
