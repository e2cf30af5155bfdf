"""Latency of score / score_grad by batch size on the row-wise small-batch path (BOHIP_SMALL_R=256 forces it up to 256 candidates)
against the default path choice: where is the break-even now that the triangular products stream?  usage: python tools/small_limit_sweep.py [N]"""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV_LIB = os.path.join(ROOT, "bayesianoptimization.jl_amd", "csrc", "abl", "libbohip_dev.so")   # the knob swept here is a constant of the shipped library: make abl/libbohip_dev.so
code = r'''
import sys, time, json
sys.path.insert(0, %r)
import numpy as np, bohip
N = int(sys.argv[1]); d = 8
rng = np.random.default_rng(0)
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
m.append_(X.T, y)
out = {}
for R in (16, 32, 48, 64, 96, 128, 160, 192, 256):
    Xs = np.asfortranarray(rng.random((d, R)))
    for _ in range(5): m.score("UCB", [2.0], Xs); m.score_grad("UCB", [2.0], Xs)
    ts, tg = [], []
    for _ in range(30):
        t0 = time.perf_counter(); m.score("UCB", [2.0], Xs); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); m.score_grad("UCB", [2.0], Xs); tg.append(time.perf_counter() - t0)
    out[R] = (float(np.median(ts)) * 1e6, float(np.median(tg)) * 1e6)
print(json.dumps(out))
''' % ROOT
N = sys.argv[1] if len(sys.argv) > 1 else "3000"
res = {}
for name, env in (("default", {}), ("row-wise forced", {"BOHIP_SMALL_R": "256"}), ("row-wise off", {"BOHIP_SMALL_R": "0"})):
    r = subprocess.run([sys.executable, "-c", code, N], env=dict({"BOHIP_LIB": DEV_LIB}, **os.environ, **env), capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    res[name] = json.loads(line[-1]) if line else {}
print(f"N={N}: us per call, score / score_grad")
for R in ("16", "32", "48", "64", "96", "128", "160", "192", "256"):
    print(f"  R={R:>3}: " + "   ".join(f"{k}: {res[k].get(R, [0, 0])[0]:6.0f} / {res[k].get(R, [0, 0])[1]:6.0f}" for k in res))
