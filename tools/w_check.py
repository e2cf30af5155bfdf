"""W = L^-1 and W' as resident on the device against numpy (tools only).  usage: python tools/w_check.py [N ...]
Only the triangles anybody reads are compared: the strictly upper tiles of W hold scratch (Z' of the executor's inverse queue)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bohip
from bohip import _lib
lib = C.CDLL(_lib.LIB_PATH)
lib.bohip_debug_read_w.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for N in ([int(a) for a in sys.argv[1:]] or (1000,)):
    d = 4
    rng = np.random.default_rng(0)
    X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
    m.append_(X.T, y)
    L = m.factor()
    W = np.zeros((N, N)); WT = np.zeros((N, N))
    assert lib.bohip_debug_read_w(m._h, 0, W.ctypes.data_as(C.c_void_p)) == 0
    assert lib.bohip_debug_read_w(m._h, 1, WT.ctypes.data_as(C.c_void_p)) == 0
    W, WT = np.tril(W), np.triu(WT)
    Wref = np.linalg.inv(L)
    scale = np.abs(Wref).max()
    alpha_ref = Wref.T @ (Wref @ y)
    print(f"N={N} form {m.info(_lib.INFO_CHOL_FORM)} fall-backs {m.info(_lib.INFO_CHOL_FALLBACKS)}: max |W - inv(L)| / max|W| = {np.abs(W - Wref).max() / scale:.2e}  "
          f"W' == W.T: {np.array_equal(WT, W.T)}  |L W - I| = {np.abs(L @ W - np.eye(N)).max():.2e}  "
          f"alpha rel err {np.abs(m.alpha() - alpha_ref).max() / np.abs(alpha_ref).max():.2e}", flush=True)
    T = (N + 127) // 128
    bad = [(i, j) for i in range(T) for j in range(i + 1) if np.abs(W[128*i:128*i+128, 128*j:128*j+128] - Wref[128*i:128*i+128, 128*j:128*j+128]).max() > 1e-9 * scale]
    if bad:
        print("  wrong tiles (first 20):", bad[:20], "of", len(bad))
    m.close()
