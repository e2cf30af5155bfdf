"""Refit time (factorisation + inverse stage, by events) under different environment settings, one child process per setting and size,
settings visited in alternation (two rounds) so that box drift shows.
usage: python tools/chol_env_sweep.py "BOHIP_CHOL_EXEC_LATE=0" "BOHIP_CHOL_EXEC_LATE=36" [...] -- 3000 10000"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
import bohip
N = int(sys.argv[1]); d = 16 if N >= 8000 else 8
rng = np.random.default_rng(N)
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
m.append_(X.T, y)
m.enable_timing(True)
ts = []
for _ in range(9):
    m.set_params_(logNoise=-2.0); m.fit_()
    t = dict(m.timing())
    ts.append(t.get("cholesky+inverse", t.get("cholesky")))
ts = sorted(ts[2:])
print(json.dumps(dict(min=ts[0], med=ts[len(ts) // 2])))
''' % ROOT
args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
settings, Ns = args[:cut], args[cut + 1:] or ["3000", "10000"]
for N in Ns:
    for rnd in range(2):
        for st in settings:
            env = dict(os.environ)
            for kv in st.split():
                k, v = kv.split("=", 1)
                env[k] = v
            r = subprocess.run([sys.executable, "-c", code, N], env=env, capture_output=True, text=True, timeout=600)
            try:
                o = json.loads(r.stdout.strip().splitlines()[-1])
                print(f"N={N:>6s}  {st:44s} min {o['min']:8.3f} ms   median {o['med']:8.3f} ms", flush=True)
            except Exception:
                print(f"N={N}  {st}: failed rc {r.returncode}: {r.stderr[-300:]}", flush=True)
