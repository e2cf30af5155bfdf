"""Randomised sweep of the small-batch pass (csrc/kernels_small.hip) against the oracle: model sizes where the number of chunks per tile,
the number of tiles and the last tile's fill all vary (N = 300 ... 7000), every dimension class, 1 ... small_limit candidates, all
kernels / acquisitions; factor, mean, variance, value, gradient, value-path == gradient-path scores (bit for bit), arg-max
(tests/test_bench_shapes_gpu.py::_fuzz_case is the checker).
usage: SEED=3 python tools/fuzz_small_pass.py [cases]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, bohip
from oracle.oracle import COracle
from test_bench_shapes_gpu import _fuzz_case
orc = COracle()
rng = np.random.default_rng(int(os.environ.get("SEED", 0)))
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
fails = 0
for c in range(ncases):
    N = int(rng.choice([300, 511, 512, 513, 900, 1279, 1280, 1281, 2000, 2431, 2432, 2433, 3000, 3071, 3072, 3073, 4500, 7000]))
    d = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 16, 17, 20, 33, 64]))
    kern = str(rng.choice(["SEArd", "SEIso", "Mat52Ard"]))
    lim = max(16, (20 + 260000 // N) // 16 * 16)
    Rs = sorted(set(int(r) for r in rng.choice([1, 2, 7, 10, 15, 16, 17, 31, 32, 33, 48, 64, 80, 96], size=4) if r <= lim))
    t0 = time.time()
    try:
        _fuzz_case(bohip, orc, rng, N, d, kern, [1, 31, 32, 500, 1500], Rs)
        print(dict(N=N, d=d, kern=kern, Rs=Rs), f"ok {time.time() - t0:.0f}s", flush=True)
    except AssertionError as e:
        fails += 1
        print(dict(N=N, d=d, kern=kern, Rs=Rs), "FAILED", str(e)[:400], flush=True)
print("failures:", fails)
