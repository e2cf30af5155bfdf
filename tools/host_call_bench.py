"""The host-pointer entry points (candidates in host memory, results in host memory) at the headline shape and at a handful of
candidates: microseconds per call of bohip_gp_score / bohip_gp_predict / bohip_gp_score_grad, beside the device-resident call.
BOHIP_LIB selects a build to compare with."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bohip
from bohip import _lib
lib = _lib.load()
N, d = 3000, 8
rng = np.random.default_rng(1)
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
m = bohip.ElasticGPE(d, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
m.append_(X.T, y); m.fit_()
params = (C.c_double * 2)(float(y.max()), 0.0)


def med(f, n):
    for _ in range(5):
        f()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return float(np.median(t)) * 1e6


for R in (4096, 512, 32, 10, 1):
    Xs = np.ascontiguousarray(rng.random((R, d)))
    xp = Xs.ctypes.data_as(C.POINTER(C.c_double))
    best = _lib.Best()
    sc = np.empty(R); mu = np.empty(R); var = np.empty(R)
    t_best = med(lambda: _lib.check(lib.bohip_gp_score(m._h, _lib.ACQ["EI"], params, xp, R, None, C.byref(best))), 200)
    t_all = med(lambda: _lib.check(lib.bohip_gp_score(m._h, _lib.ACQ["EI"], params, xp, R, sc.ctypes.data_as(C.POINTER(C.c_double)), C.byref(best))), 200)
    t_pred = med(lambda: _lib.check(lib.bohip_gp_predict(m._h, xp, R, mu.ctypes.data_as(C.POINTER(C.c_double)), var.ctypes.data_as(C.POINTER(C.c_double)))), 200)
    print(f"N={N} d={d} R={R}: score(best only) {t_best:7.1f} us   score(all scores + best) {t_all:7.1f} us   predict {t_pred:7.1f} us", flush=True)
