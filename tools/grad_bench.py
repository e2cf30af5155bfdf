import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, bohip
rng = np.random.default_rng(0)
N, d = 3000, 8
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
m.enable_timing(True); m.append_(X.T, y)
for R in (10, 256, 4096):
    Xs = np.asfortranarray(rng.random((d, R)))
    m.score_grad("EI", [y.max()], Xs)
    t0 = time.perf_counter()
    for _ in range(5): m.score_grad("EI", [y.max()], Xs)
    t = (time.perf_counter() - t0) / 5
    print(R, f"{t*1e3:.3f} ms/call", [(k, round(v, 3)) for k, v in m.timing()])
    t0 = time.perf_counter()
    for _ in range(5): m.score("EI", [y.max()], Xs)
    print("   score only", f"{(time.perf_counter()-t0)/5*1e3:.3f} ms/call")
