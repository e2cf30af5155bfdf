"""Refit (all stages, by events) and wall time of a full model update at the sizes the reference's own examples use (N <= 500).
usage: python tools/small_refit.py [N ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bohip
rng = np.random.default_rng(0)
for N in ([int(a) for a in sys.argv[1:]] or (20, 50, 100, 127, 200, 300, 383, 400, 500)):
    d = 2
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
    m.append_(X.T, y)
    m.enable_timing(True)
    best, wall = None, 1e9
    for _ in range(9):
        t0 = time.perf_counter()
        m.set_params_(logNoise=-2.0); m.fit_()
        wall = min(wall, time.perf_counter() - t0)
        t = dict(m.timing())
        if best is None or sum(t.values()) < sum(best.values()): best = t
    print(f"N={N:4d}: stages (us) {dict((k, round(v * 1e3, 1)) for k, v in best.items())}  sum {sum(best.values()) * 1e3:.0f} us, wall {wall * 1e6:.0f} us, form {m.info(4)}", flush=True)
    m.close()
