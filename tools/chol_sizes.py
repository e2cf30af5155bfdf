"""Cholesky time of the factorisation paths over N (run once per path: BOHIP_CHOL_DATAFLOW=0 / 1 / 2).  usage: python tools/chol_sizes.py [N ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, bohip
from bohip import _lib
if not os.environ.get("BOHIP_KEEP_INV"):   # the factorisation ALONE: without the executor's inverse queues (W = L^-1 then follows as its own stage)
    C.CDLL(_lib.LIB_PATH).bohip_debug_set_chol_inv_g(0)
rng = np.random.default_rng(0)
for N in ([int(a) for a in sys.argv[1:]] or (500, 1000, 2000, 3000, 4000, 5000, 6000, 8000)):
    d = 8
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
    m.append_(X.T, y)
    m.enable_timing(True)
    best = 1e9
    for _ in range(5):
        m.set_params_(logNoise=-2.0); m.fit_()
        t = dict(m.timing())
        best = min(best, t.get("cholesky", t.get("cholesky+inverse")))
    print(f"N={N}: cholesky {best:.3f} ms = {N**3/3/best/1e9:.2f} TF/s", flush=True)
    m.close()
