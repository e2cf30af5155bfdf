"""W = L^-1 under contention: K host threads refit their own models on ONE device at the same time (executor kernels of different
handles then share the chip, workgroups of one refit land on whatever XCD is free), every refit's W must be the inverse of its factor to rounding (contention makes refits time out and fall back to
the launch chain, whose W differs in the last bits: bit-identity between refits is reported, not required).  This is the situation in which the first group form of the inverse queues read
stale tiles (tests/test_exec_tasks.py: "written more than once").   usage: [BOHIP_CHOL_INV_GRP_MIN=0] python tools/w_stress.py N threads refits"""
import ctypes as C, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bohip
from bohip import _lib
N, K, R = (int(a) for a in (sys.argv[1:4] + ["3000", "4", "6"][len(sys.argv) - 1:]))
lib = C.CDLL(_lib.LIB_PATH)
lib.bohip_debug_read_w.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
d = 4
out = [None] * K


def work(t):
    rng = np.random.default_rng(t)
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N + 128 * (t % 3))
    m.append_(X.T, y)
    first, worst, diffs = None, 0.0, 0
    for r in range(R):
        m.set_params_(logNoise=-2.0); m.fit_()
        W = np.zeros((N, N))
        assert lib.bohip_debug_read_w(m._h, 0, W.ctypes.data_as(C.c_void_p)) == 0
        W = np.tril(W)
        L = m.factor()
        worst = max(worst, np.abs(L @ W - np.eye(N)).max())
        if first is None:
            first = W
        elif not np.array_equal(W, first):
            diffs += 1
    out[t] = (worst, diffs, m.info(_lib.INFO_CHOL_FORM), m.info(_lib.INFO_CHOL_FALLBACKS))
    m.close()


ths = [threading.Thread(target=work, args=(t,)) for t in range(K)]
[t.start() for t in ths]; [t.join() for t in ths]
for t, o in enumerate(out):
    print(f"thread {t}: max |L W - I| over {R} refits {o[0]:.2e}; refits whose W differs from the first in some bit: {o[1]} of {R - 1}; form {o[2]}, fall-backs {o[3]}")
print("N", N, "threads", K, "OK" if all(o and o[0] < 1e-9 for o in out) else "MISMATCH")
