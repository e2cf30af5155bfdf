"""K*' chunk size sweep (BOHIP_CHUNK_ROWS) for a multi-chunk batch: host-call time of one score over R candidates.
usage: python tools/chunk_sweep.py N d R rows [rows ...]   (rows = 0: the library's rule)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV_LIB = os.path.join(ROOT, "bayesianoptimization.jl_amd", "csrc", "abl", "libbohip_dev.so")   # the knob swept here is a constant of the shipped library: make abl/libbohip_dev.so
code = r'''
import sys, time, json
sys.path.insert(0, %r)
import numpy as np, bohip
from bohip import _lib
N, d, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(0)
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.7 if d > 8 else 0.5)), 0.0), logNoise=-2.0, capacity=N)
m.append_(X.T, y)
Xs = np.asfortranarray(rng.random((d, R)))
tau = float(y.max())
for _ in range(3): m.score("EI", [tau], Xs, want_scores=False)
ts = []
for _ in range(12):
    t0 = time.perf_counter(); m.score("EI", [tau], Xs, want_scores=False); ts.append(time.perf_counter() - t0)
print(json.dumps(dict(ms=float(np.median(ts)) * 1e3, chunk=m.info(_lib.INFO_SCORE_CHUNK), launches=m.info(_lib.INFO_SCORE_LAUNCHES))))
''' % ROOT
N, d, R = sys.argv[1:4]
for rows in sys.argv[4:]:
    env = dict(os.environ); env.setdefault("BOHIP_LIB", DEV_LIB)
    if rows != "0": env["BOHIP_CHUNK_ROWS"] = rows
    r = subprocess.run([sys.executable, "-c", code, N, d, R], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print(f"N={N} R={R} rows={rows:>5}:", line[-1] if line else r.stderr[-200:])
