"""Refit N observations `reps` times with the factorisation alone (form 1 below 32 row tiles) and let the library dump the flag area on a
time-out (BOHIP_CHOL_DF_DUMP=1).  usage: BOHIP_CHOL_DF_DUMP=1 python tools/chol_dump.py N reps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, bohip
from bohip import _lib
C.CDLL(_lib.LIB_PATH).bohip_debug_set_chol_inv_g(0)
N = int(sys.argv[1]); reps = int(sys.argv[2]); d = 8
rng = np.random.default_rng(N)
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
m.append_(X.T, y)
import time
for r in range(reps):
    m.set_params_(logNoise=-2.0); t0 = time.perf_counter(); m.fit_(); print("refit", r, "wall %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    if m.info(5):
        print("fall-back at refit", r); break
