"""Where a small-batch pass spends its time INSIDE its two kernels (kernels_small.hip): wall-clock marks per workgroup.
Build: make -C bayesianoptimization.jl_amd/csrc abl/libbohip_smalltrace.so ; run: BOHIP_LIB=.../abl/libbohip_smalltrace.so python tools/small_pass_trace.py N d [R]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BOHIP_LIB", os.path.join(ROOT, "bayesianoptimization.jl_amd", "csrc", "abl", "libbohip_smalltrace.so"))
import numpy as np, bohip
from bohip import _lib
N, d = int(sys.argv[1]), int(sys.argv[2]); R = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = np.random.default_rng(0)
X = rng.random((N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
m = bohip.ElasticGPE(d, kernel=bohip.SEArd(np.full(d, np.log(0.5)), 0.0), logNoise=-2.0, capacity=N)
m.append_(X.T, y)
Xs = np.asfortranarray(rng.random((d, R)))
lib = _lib.load()
names = {0: ["start", "tile known", "first rhs tile in LDS", "contraction done", "tile published (stores acked)", "counted in", "combined (last arriver)",
             "block record out", "pass counted", "posterior final", "arg-max", ""],
         1: ["start", "tile known", "first rhs tile in LDS", "contraction done", "tile's gradient record out", "counted in", "records fetched (last tile)", "posterior + value done (thread 0)", "",
             "last tile of the pass", "posterior final", "gradient out"]}
for what in ("score_grad", "score"):
    for _ in range(20):
        (m.score_grad if what == "score_grad" else m.score)("EI", [y.max()], Xs)
    buf = np.zeros(8192 * 16, dtype=np.uint64)
    assert lib.bohip_debug_small_trace_read(buf.ctypes.data_as(C.c_void_p)) == 0
    buf = buf.reshape(2, 4096, 16).astype(np.int64)
    for kern in ((0, 1) if what == "score_grad" else (0,)):
        t = buf[kern]
        live = t[:, 0] > 0
        if not live.any():
            continue
        t0 = t[live, 0].min()
        print(f"== {what}: kernel {'k_small_u' if kern else 'k_small_v'}, {int(live.sum())} workgroups; marks in us from the first workgroup's start (min / median / max over the workgroups that reached the mark)")
        for i in range(12):
            col = t[live, i]
            col = col[col > 0]
            if len(col):
                rel = (col - t0) / 100.0
                print(f"   {i:2d} {names[kern][i]:34s} n={len(col):4d}  {rel.min():7.2f} {np.median(rel):7.2f} {rel.max():7.2f}")
