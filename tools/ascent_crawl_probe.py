"""Which start point keeps a device ascent alive?  End values per start for growing maxeval (EI at tau = median y, headline model, LHS seed 7)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth, lhs, DIM
import bohip
X, y = synth(0)
ll = np.full(DIM, np.log(0.5))
m = bohip.ElasticGPE(DIM, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(ll, 0.0), logNoise=-2.0, capacity=len(y))
m.append_(X.T, y)
lb, ub = np.zeros(DIM), np.ones(DIM)
tau = float(np.median(y))
starts = np.asfortranarray(lhs(10, seed=int(sys.argv[1]) if len(sys.argv) > 1 else 7).T)
prev = None
for me in (2, 20, 100, 150, 200, 400, 800, 1600, 2000):
    f, Xd, bf, bi, bx, ev = m.ascend("EI", [tau], lb, ub, starts, me)
    ch = "" if prev is None else " changed: " + str(np.flatnonzero(f != prev).tolist())
    print(f"maxeval {me:5d}: passes {ev:5d}{ch}\n    f = {np.array2string(f, precision=10)}")
    if prev is not None:
        for j in np.flatnonzero(f != prev):
            sc, g = m.score_grad("EI", [tau], Xd[:, j:j + 1])
            print(f"      start {j}: f {f[j]:.17g} (+{f[j] - prev[j]:.3e}), x {np.array2string(Xd[:, j], precision=6)}, g {np.array2string(np.asarray(g)[:, 0], precision=3)}")
    prev = f
print("--- a call in which no start is active (EI at tau = max y), then the same call as above")
f, Xd, bf, bi, bx, ev = m.ascend("EI", [float(y.max())], lb, ub, starts, 2000); print("flat call: passes", ev)
f, Xd, bf, bi, bx, ev = m.ascend("EI", [tau], lb, ub, starts, 2000); print("median call behind it: passes", ev, "f", np.array2string(f, precision=6))
f, Xd, bf, bi, bx, ev = m.ascend("EI", [tau], lb, ub, starts, 2000); print("median call again: passes", ev)
print("--- UCB call, flat EI call, median EI call (the order of tests/test_bench_shapes_gpu.py)")
from bench import BETA_T
f, Xd, bf, bi, bx, ev = m.ascend("UCB", [BETA_T], lb, ub, starts, 2000); print("UCB call: passes", ev)
f, Xd, bf, bi, bx, ev = m.ascend("EI", [float(y.max())], lb, ub, starts, 2000); print("flat call: passes", ev)
f, Xd, bf, bi, bx, ev = m.ascend("EI", [tau], lb, ub, starts, 2000); print("median call behind it: passes", ev, "f", np.array2string(f, precision=6))
m2 = bohip.ElasticGPE(DIM, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(ll, 0.0), logNoise=-2.0, capacity=1024)
m2.append_(X.T, y)
f, Xd, bf, bi, bx, ev = m2.ascend("EI", [tau], lb, ub, starts, 2000); print("fresh handle with capacity 1024 grown to 3000, median call: passes", ev)
