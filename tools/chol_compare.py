"""Factor of one path against another: first run saves L to the file, later runs compare with it (set BOHIP_CHOL_* per run).
usage: python tools/chol_compare.py N file.npy [trials]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bohip
N, out = int(sys.argv[1]), sys.argv[2]
trials = int(sys.argv[3]) if len(sys.argv) > 3 else 1
d = 8
rng = np.random.default_rng(0)
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
for t in range(trials):
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N + 128 * (t % 3))
    try:
        m.append_(X.T, y)
        L = m.factor()
        mu, var = m.predict_f(X[:64].T + 0.01)
    except Exception as e:
        print(N, t, "FAILED", str(e)[-90:], flush=True); m.close(); continue
    if not os.path.exists(out):
        np.save(out, L); np.save(out + ".mu.npy", np.stack([mu, var])); print(N, "saved", out, flush=True)
    else:
        R = np.load(out); D = np.abs(L - R); mv = np.load(out + ".mu.npy")
        bad = np.argwhere(D > 1e-9 * np.abs(R).max())
        print(N, t, "max |dL|", D.max(), "bad entries", len(bad), "tiles", sorted({(int(i) // 128, int(j) // 128) for i, j in bad})[:8],
              "max |dmu|", np.abs(mu - mv[0]).max(), "max |dvar|", np.abs(var - mv[1]).max(), flush=True)
    m.close()
