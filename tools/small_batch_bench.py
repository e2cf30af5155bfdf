"""Latency of the reference's default usage: a handful of restarts scored WITH gradient per call (R = 1..32)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bohip
rng = np.random.default_rng(0)
for N, d in ((3000, 8), (500, 2), (10000, 16)):
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
    m.append_(X.T, y)
    for R in (1, 10, 32):
        Xs = np.asfortranarray(rng.random((d, R)))
        for timing in (False, True):
            m.enable_timing(timing)
            m.score_grad("EI", [y.max()], Xs)
            t0 = time.perf_counter()
            for _ in range(50): m.score_grad("EI", [y.max()], Xs)
            t = (time.perf_counter() - t0) / 50
            if timing:
                print(f"N={N} R={R}   stages(us):", [(k, round(v * 1e3, 1)) for k, v in m.timing()])
            else:
                t0 = time.perf_counter()
                for _ in range(50): m.score("EI", [y.max()], Xs)
                t2 = (time.perf_counter() - t0) / 50
                print(f"N={N} R={R}   score_grad {t*1e6:.0f} us/call   score {t2*1e6:.0f} us/call   (one pass over W at 8 TB/s: {N*N*4/8e12*1e6:.1f} us)")
