"""A/B of two builds of libbohip on the factorisation: time by size AND the factor of build B against build A (bitwise).
usage: python tools/chol_ab.py abl/libbohip_other.so [N ...]   (A = the production library)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, json
sys.path.insert(0, %r)
import numpy as np
from bohip import _lib
if os.environ.get("LIBV"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["LIBV"])
import ctypes as C
import bohip
if os.environ.get("ALONE"):
    C.CDLL(_lib.LIB_PATH).bohip_debug_set_chol_inv_g(0)
out = {}
for N in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(N)
    d = 8
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
    m.append_(X.T, y)
    m.enable_timing(True)
    best = 1e9
    for _ in range(7):
        m.set_params_(logNoise=-2.0); m.fit_()
        t = dict(m.timing())
        best = min(best, t.get("cholesky", t.get("cholesky+inverse")))
    L = m.factor()
    import hashlib
    out[N] = dict(ms=best, sha=hashlib.sha256(np.ascontiguousarray(L).tobytes()).hexdigest()[:16], alpha=hashlib.sha256(m.alpha().tobytes()).hexdigest()[:16])
    m.close()
print(json.dumps(out))
''' % ROOT
other = sys.argv[1]
Ns = sys.argv[2:] or ["1000", "3000", "6000", "10000"]
for alone in ("1", ""):
    res = {}
    for name, libv in (("A production", ""), ("B " + other, other)):
        env = dict(os.environ, LIBV=libv, ALONE=alone)
        r = subprocess.run([sys.executable, "-c", code] + Ns, env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        res[name] = json.loads(line[-1]) if line else {"error": r.stderr[-300:]}
    print("== factorisation alone" if alone else "== factorisation + inverse (as shipped)")
    for N in Ns:
        a, b = res["A production"].get(N), res["B " + other].get(N)
        if a and b:
            print(f"N={N}: A {a['ms']:.3f} ms   B {b['ms']:.3f} ms   factor identical: {a['sha'] == b['sha']}   alpha identical: {a['alpha'] == b['alpha']}")
        else:
            print(N, res)
