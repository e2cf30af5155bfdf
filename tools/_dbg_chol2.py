import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, bohip
N, d = int(sys.argv[1]), 16
rng = np.random.default_rng(4)
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.7)), 0.0), logNoise=-2.0, capacity=N)
m.append_(X.T, y)
L = m.factor()
out = sys.argv[2]
if os.path.exists(out):
    R = np.load(out)
    D = np.abs(L - R)
    bad = np.argwhere(D > 1e-9 * np.abs(R).max())
    print("max diff", D.max(), "bad entries", len(bad))
    tiles = {}
    for i, j in bad:
        tiles[(i // 128, j // 128)] = tiles.get((i // 128, j // 128), 0) + 1
    for t in sorted(tiles)[:40]:
        print("  tile", t, tiles[t])
    if len(bad):
        i, j = bad[0]
        print("first bad", i, j, L[i, j], R[i, j], "row within tile", i % 128, "col within tile", j % 128)
        rows = sorted(set(bad[:, 0] % 128)); cols = sorted(set(bad[:, 1] % 128))
        print("rows%128", rows[:20], "cols%128", cols[:40])
else:
    np.save(out, L); print("saved", out)
