"""Repeatability stress of the factorisation on FRESH handles (new allocations, varying capacity -> varying row stride): every
trial must reproduce the first one bit for bit.  usage: [BOHIP_CHOL_DATAFLOW=0|2] [BOHIP_CHOL_DF_STRICT=1] python tools/chol_stress.py N trials"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bohip
N, trials = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(0)
d = 8
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
ref = None
for t in range(trials):
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N + 128 * (t % 3))
    try:
        m.append_(X.T, y)
        L = m.factor()
        if ref is None: ref = L; print(t, "ref ok", flush=True)
        else:
            D = np.abs(L - ref); bad = np.argwhere(D > 1e-9 * np.abs(ref).max())
            print(t, "max diff", D.max(), "bad tiles", sorted({(int(i) // 128, int(j) // 128) for i, j in bad})[:10], flush=True)
    except Exception as e:
        print(t, "FAILED", str(e)[-60:], flush=True)
    m.close()
