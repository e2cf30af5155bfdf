"""Full model update (covariance, Cholesky, inverse, alpha) per size: host time of refit + stage times.  usage: python tools/refit_bench.py [N ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bohip
rng = np.random.default_rng(0)
for N in ([int(a) for a in sys.argv[1:]] or (1000, 2000, 3000, 4000, 6000)):
    d = 8
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
    m.append_(X.T, y)
    best, st = 1e9, None
    for rep in range(8):
        m.enable_timing(rep >= 4)
        m.set_params_(logNoise=-2.0)
        t0 = time.perf_counter(); m.fit_(); dt = (time.perf_counter() - t0) * 1e3
        if rep < 4: best = min(best, dt)
        else: st = dict(m.timing())
    print(f"N={N}: refit {best:.3f} ms (host call, no stage events) | stages " + " ".join(f"{k} {v:.3f}" for k, v in st.items()), flush=True)
    m.close()
