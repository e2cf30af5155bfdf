"""Step time of the C2 workload (bench.py's step) under different event-timing modes and streams (tools only)."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch

import bench
import bohip
from bohip import _lib

lib = _lib.load()
X, y = bench.synth(0)
tau = float(y.max())
Xs = bench.lhs(bench.R_PER_GPU, seed=1)
ll = np.full(bench.DIM, np.log(0.5))
m = bohip.ElasticGPE(bench.DIM, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(ll, 0.0), logNoise=-2.0, capacity=bench.N_OBS)
m.append_(X.T, y)
dXs = torch.from_numpy(np.ascontiguousarray(Xs)).to("cuda:0")
h_best = torch.tensor([0, -1], dtype=torch.int64).pin_memory()
params = (C.c_double * 2)(tau, 0.0)
side = torch.cuda.Stream()


def run(mode, stream, steps=200):
    if stream == "own":
        lib.bohip_gp_set_stream(m._h, None)
    elif stream == "torch-default":
        lib.bohip_gp_set_stream(m._h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    else:
        lib.bohip_gp_set_stream(m._h, C.c_void_p(side.cuda_stream))
    m.enable_timing(mode)

    def step():
        lib.bohip_gp_score_dev(m._h, 0, params, C.c_void_p(dXs.data_ptr()), bench.R_PER_GPU, None, C.c_void_p(h_best.data_ptr()))
        lib.bohip_gp_synchronize(m._h)

    for _ in range(10):
        step()
    m.timing(4096)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps * 1e3
    tg = [ms for n, ms in m.timing(4096) if n == "trigemm_sq"]
    print(f"timing mode {mode} stream {stream:14s}: {el:.4f} ms/step" + (f"  trigemm {np.mean(tg):.4f} ms ({len(tg)} records)" if tg else ""))


for rep in range(2):
    for stream in ("own", "torch-default", "torch-side"):
        for mode in (0, 3, 2, 1):
            run(mode, stream)
