"""Would two groups of start points on two streams overlap?  Probe with what exists: TWO handles holding the same model (N = 3000, d = 8), each
running acquire_max on 5 of the 10 default starts from its own host thread, against one handle with all 10."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, bohip
from bench import synth, lhs, DIM
X, y = synth(0)
ll = np.full(DIM, np.log(0.5))
def mk():
    m = bohip.ElasticGPE(DIM, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(ll, 0.0), logNoise=-2.0, capacity=len(y)); m.append_(X.T, y); return m
m1, m2 = mk(), mk()
starts = np.asfortranarray(lhs(10, seed=7).T); lb, ub = np.zeros(DIM), np.ones(DIM); bt = 10.152008469453344
def run(m, s, out, i):
    out[i] = m.ascend("UCB", [bt], lb, ub, s, 2000)
for _ in range(3): m1.ascend("UCB", [bt], lb, ub, starts, 2000)
ts = []
for _ in range(9):
    t0 = time.perf_counter(); r = m1.ascend("UCB", [bt], lb, ub, starts, 2000); ts.append(time.perf_counter() - t0)
print(f"one handle, 10 starts: {np.median(ts)*1e3:.3f} ms, {r[5]} passes")
sa, sb = np.asfortranarray(starts[:, :5]), np.asfortranarray(starts[:, 5:])
for _ in range(3): m1.ascend("UCB", [bt], lb, ub, sa, 2000); m2.ascend("UCB", [bt], lb, ub, sb, 2000)
ts = []
for _ in range(9):
    t0 = time.perf_counter(); ra = m1.ascend("UCB", [bt], lb, ub, sa, 2000); ts.append(time.perf_counter() - t0)
print(f"one handle, 5 starts: {np.median(ts)*1e3:.3f} ms, {ra[5]} passes")
ts = []
for _ in range(9):
    out = [None, None]
    th = [threading.Thread(target=run, args=(m1, sa, out, 0)), threading.Thread(target=run, args=(m2, sb, out, 1))]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    ts.append(time.perf_counter() - t0)
print(f"two handles x 5 starts, concurrently: {np.median(ts)*1e3:.3f} ms, passes {out[0][5]} / {out[1][5]}; best {max(out[0][2], out[1][2]):.6f} vs {r[2]:.6f}")
