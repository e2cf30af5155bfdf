"""Floor under the executor form of the factorisation + inverse: the SAME task records with their counters removed (nothing to wait
for, no chain kernel beside them), per queue and together, and once more with the bulk / wave operands pointed at one L2-resident
panel (same instruction stream, no fabric-side operand traffic).  Says how much of the refit time is scheduling and how much is the
tasks themselves.   python tools/exec_throughput.py [N=10000] [d=16]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bohip
from bohip import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(0)
X = rng.random((n, d)); y = np.sin(X.sum(1))
m = bohip.ElasticGPE(d, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=n)
m.append_(X.T, y)
m.enable_timing(True)
for _ in range(3):
    m.fit_()
print(f"N = {n}: refit stages (ms) {dict((k, round(v, 3)) for k, v in m.timing())}")
lib = _lib.load()
f = lib.bohip_debug_exec_throughput
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
names = ["urgent", "solve+late", "early", "inv rows", "bulk", "inv waves"]
cases = [("all six queues", 0x3f), ("factorisation (0,1,2,4)", 0x17), ("inverse (3,5)", 0x28), ("bulk only", 0x10), ("waves only", 0x20),
         ("solve+late only", 0x02), ("early only", 0x04), ("inv rows only", 0x08)]
for wgs in (0, 256):
    for hot in (0, 1):
        for name, mask in cases:
            ts = []
            gf = C.c_double(0.0)
            for _ in range(3):
                ms = C.c_double(0.0)
                rc = f(m._h, mask, hot, wgs, C.byref(ms), C.byref(gf))
                if rc != 0:
                    print(f"{name}: rc {rc}"); break
                ts.append(ms.value)
            if ts:
                t = min(ts)
                print(f"wgs {wgs or 'default':>7} {'hot operands' if hot else 'real operands':14s} {name:26s} {t:8.3f} ms  {gf.value:9.1f} GF  {gf.value / t:7.1f} GF/ms = TF/s", flush=True)
m.fit_()
print("refit after:", dict((k, round(v, 3)) for k, v in m.timing()))
