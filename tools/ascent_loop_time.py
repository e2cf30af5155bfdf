"""Mean time of 40 back-to-back device ascents (N = 3000, d = 8, ten starts, UCB): what a BO loop pays per acquire_max.
Under rocprofv3 --kernel-trace this is the run behind profiles/r04_ascent_kernel_timeline.txt.  usage: python tools/ascent_loop_time.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bohip
N, d, R = 3000, 8, 10
rng = np.random.default_rng(0)
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
m.append_(X.T, y)
lb, ub = np.zeros(d), np.ones(d)
starts = np.asfortranarray(rng.random((d, R)))
for i in range(3): m.ascend("UCB", [2.0], lb, ub, starts, 200)
t0 = time.perf_counter()
for i in range(40): out = m.ascend("UCB", [2.0], lb, ub, starts, 200)
print("ascend ms", (time.perf_counter() - t0) / 40 * 1e3, "evals", out[-1])
