import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, bohip
rng = np.random.default_rng(0)
N, d = int(sys.argv[1]), int(sys.argv[2])
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
m.append_(X.T, y)
for R in (10,):
    Xs = np.asfortranarray(rng.random((d, R)))
    for _ in range(200): m.score_grad("EI", [y.max()], Xs)
    for _ in range(200): m.score("EI", [y.max()], Xs)
