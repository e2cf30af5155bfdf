"""Child for `rocprofv3 --pmc ...`: a few refits of an N x d model with the launch-chained factorisation (counter collection serialises kernels:
the dataflow forms would time out).  usage: python tools/build_cov_child.py [N d]"""
import os, sys
os.environ["BOHIP_CHOL_DATAFLOW"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bohip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(3)
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
m.append_(X.T, y)
for _ in range(3):
    m.set_params_(logNoise=-2.0); m.fit_()
