"""Soak: fresh processes refitting a model of N observations `reps` times each; reports time-outs (fall-backs) of the dataflow forms.
usage: python tools/chol_soak.py N processes reps"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, bohip
N = int(sys.argv[1]); reps = int(sys.argv[2]); d = 16 if N >= 8000 else 8
rng = np.random.default_rng(N)
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
m.append_(X.T, y)
m.enable_timing(True)
ts = []
for _ in range(reps):
    m.set_params_(logNoise=-2.0); m.fit_()
    t = dict(m.timing()); ts.append(t.get("cholesky+inverse", -t.get("cholesky", 0.0)))
print("RESULT", m.info(5), m.info(4), "lock-skips=%%d" %% m.info(14), " ".join("%%.2f" %% v for v in ts))
''' % ROOT
N, procs, reps = sys.argv[1], int(sys.argv[2]), sys.argv[3]
bad = 0
for p in range(procs):
    r = subprocess.run([sys.executable, "-c", code, N, reps], capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    err = [l for l in r.stderr.splitlines() if "libbohip" in l]
    fb = int(line[0].split()[1]) if line else -1
    bad += fb != 0
    print(f"process {p}: fall-backs {fb}, form {line[0].split()[2] if line else '?'}, stage ms (negative: factorisation alone after a fall-back) {' '.join(line[0].split()[3:]) if line else r.stderr[-200:]}", flush=True)
    for e in err: print("   ", e, flush=True)
print(f"{bad} of {procs} processes saw a time-out")
