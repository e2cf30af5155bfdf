"""W = L^-1 and W' as resident on the device against numpy (tools only).  usage: python tools/w_check.py [N]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bohip
from bohip import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
d = 4
rng = np.random.default_rng(0)
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
m.append_(X.T, y)
L = m.factor()
lib = C.CDLL(_lib.LIB_PATH)
W = np.zeros((N, N)); WT = np.zeros((N, N))
lib.bohip_debug_read_w.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
assert lib.bohip_debug_read_w(m._h, 0, W.ctypes.data_as(C.c_void_p)) == 0
assert lib.bohip_debug_read_w(m._h, 1, WT.ctypes.data_as(C.c_void_p)) == 0
Wref = np.linalg.inv(L)
T = (N + 127) // 128
print("form", m.info(_lib.INFO_CHOL_FORM), "max |W - inv(L)|", np.abs(W - Wref).max(), " max |WT - inv(L)'|", np.abs(WT - Wref.T).max())
for k in range(T):
    s = slice(128 * k, min(N, 128 * k + 128))
    e0, e1 = np.abs(W[s, s] - Wref[s, s]), np.abs(WT[s, s] - Wref.T[s, s])
    if e0.max() > 1e-9 or e1.max() > 1e-9:
        bad0 = np.argwhere(e0 > 1e-9); bad1 = np.argwhere(e1 > 1e-9)
        print(f"diag block {k}: W err {e0.max():.2e} ({len(bad0)} entries, rows {sorted(set(bad0[:,0]//16))} col-panels {sorted(set(bad0[:,1]//16))}); "
              f"WT err {e1.max():.2e} ({len(bad1)} entries, row-panels {sorted(set(bad1[:,0]//16))} col-panels {sorted(set(bad1[:,1]//16))})")
k = 1
s = slice(128 * k, 128 * k + 128)
e1 = np.abs(WT[s, s] - Wref.T[s, s])
bad = np.argwhere(e1 > 1e-9)
print("block 1 wrong WT entries (row c, col):", bad[:60].tolist())
print("by col panel:", np.bincount(bad[:, 1] // 16, minlength=8).tolist(), " by row panel:", np.bincount(bad[:, 0] // 16, minlength=8).tolist())
print("col mod 16 histogram:", np.bincount(bad[:, 1] % 16, minlength=16).tolist())
print("is the wrong value another correct entry? e.g. WT[c][col] == W[col2][c]:")
for (r_, c_) in bad[:8]:
    val = WT[s, s][r_, c_]
    hits = np.argwhere(np.abs(Wref[s, s] - val) < 1e-12)
    print((int(r_), int(c_)), "got", val, "expected", Wref.T[s, s][r_, c_], "value found in inv(L) at", hits[:3].tolist())
