"""Timeline of the dataflow Cholesky chain (build: make -C bayesianoptimization.jl_amd/csrc abl/libbohip_choltrace.so).
wall_clock64 ticks are 10 ns.  usage: python tools/chol_trace.py [N]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bohip import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "abl", "libbohip_choltrace.so")
import bohip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
d = 8
rng = np.random.default_rng(0)
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
m.append_(X.T, y)
for _ in range(3):
    m.set_params_(logNoise=-2.0); m.fit_()
lib = _lib.load()
buf = (C.c_ulonglong * 8192)()
lib.bohip_debug_chol_trace_read.argtypes = [C.c_void_p, C.c_int64]
assert lib.bohip_debug_chol_trace_read(buf, 8192) == 0
t = np.array(buf, dtype=np.int64)
T = (N + 1 + 127) // 128
pub, saw, fin, blk = t[:1024], t[1024:2048], t[2048:3072], t[3072:]
t0 = blk[0]
us = lambda v: float(v - t0) / 100.0
for k in range(min(T, 6)):
    print(f"block {k}: pivot start {us(blk[2*k]):8.1f} end {us(blk[2*k+1]):8.1f} us | published {[round(us(pub[8*k+p]),1) for p in range(8)]}")
    if k + 1 < T:
        print(f"          owner of row {k+1}: saw {[round(us(saw[8*k+p]),1) for p in range(8)]}")
        print(f"                           done {[round(us(fin[8*k+p]),1) for p in range(8)]}")
dur = [(blk[2*k+1] - blk[2*k]) / 100.0 for k in range(T)]
gap = [(blk[2*(k+1)] - blk[2*k+1]) / 100.0 for k in range(T - 1)]
print(f"pivot duration per block: mean {np.mean(dur):.1f} us (min {np.min(dur):.1f}, max {np.max(dur):.1f}); gap to next pivot start: mean {np.mean(gap):.1f} us (min {np.min(gap):.1f} max {np.max(gap):.1f}); total {us(blk[2*(T-1)+1]):.1f} us")

for k in range(min(T - 2, 4)):
    g_ = t[3300 + 4 * k: 3300 + 4 * k + 4]
    r = k + 2
    print(f"block {k}: gated row-{r} update: chunks done {us(g_[0]):.1f}, pre ok {us(g_[1]):.1f}, stored {us(g_[2]):.1f} | owner of row {r}: starts waiting {us(g_[3]):.1f}, "
          f"sees crit {us(t[3400 + r]):.1f}, tiles loaded {us(t[3500 + r]):.1f}")
for k in range(min(T - 1, 5)):
    for sl in (0, 1):
        b = 3648 + (k * 2 + sl) * 9
        if t[b + 8] > 0:
            print(f"block {k}: critical follower of row {k + 1 + sl}: tile loaded {us(t[b + 8]):.1f}, panels done {[round(us(t[b + p]), 1) for p in range(8)]}")
ph = t[3584:3584 + 64].reshape(8, 8)
b1 = blk[2]
print("block 1, pivot workgroup, per panel (us since the block's pivot start): row solves done | next pivot block updated | "
      "factor16 done (wave 0) | trailing done (wave 1) | panel end")
for p_ in range(8):
    print("  panel", p_, [round((ph[p_][i] - b1) / 100.0, 2) if ph[p_][i] > 0 else None for i in (0, 1, 3, 4, 2)])

print("block 1, per panel (us since the block's pivot start): W16 done (wave 4) | row solve done (thread 0, before the barrier) | barrier passed")
for jb in range(8):
    print("  panel", jb, [None if ph[jb][c] == 0 else round((ph[jb][c] - t[3072 + 2]) / 100.0, 2) for c in (5, 6, 0)])

print("block 1, the panel flag's way (us since the block's pivot start): wave 5 starts draining its stores | flag +1 by wave 6 | wave 7 | wave 5 | "
      "owner of row 2 starts waiting | sees the flag")
for jb in range(8):
    r = lambda v: None if v == 0 else round((v - t[3072 + 2]) / 100.0, 2)
    print("  panel", jb, [r(t[5904 + jb]), r(t[5888 + jb]), r(t[5896 + jb]), r(t[8 + jb]), r(t[5912 + jb]), r(t[1024 + 8 + jb])])

print("inverse of the diagonal block published (us after the pivot block's end), blocks 0..7:",
      [round((t[4096 + k] - blk[2 * k + 1]) / 100.0, 1) for k in range(min(T, 8))])
print("gap pivot end -> next pivot start, every block (us):", [round(float(g_), 1) for g_ in gap])
print("pivot block duration, every block (us):", [round(float(d_), 1) for d_ in dur])
