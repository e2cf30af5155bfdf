"""Writes tests/golden/julia_inputs.txt: the INPUTS of the six committed golden cases plus a C2-shaped sample
(N=3000, d=8, 64 candidates, BASELINE.md recipe), for julia/gen_golden.jl to run through the real
GaussianProcesses.jl / BayesianOptimization.jl on a machine that has Julia.  Data only; no reference source.

    python tests/golden/export_inputs_for_julia.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gio import write_cases  # noqa: E402

KERN = {"n1_seiso_maxmean": "SEIso"}
MEANZERO = {"n1_seiso_maxmean"}          # GPE([1.0],[2.0],MeanZero(),SEIso(1.0,0.0)), reference test/acquisition.jl:2


def main():
    cases = {}
    for name in ("n1_seiso_maxmean", "n2_seard", "readme_d2_rep5", "branin_shaped", "n256_d8_r512", "ties_n256"):
        g = dict(np.load(os.path.join(HERE, name + ".npz")))
        c = dict(kern=KERN.get(name, "SEArd"), mean="MeanZero" if name in MEANZERO else "MeanConst", X=g["X"], y=g["y"],
                 loglen=g["loglen"][:1] if KERN.get(name) == "SEIso" else g["loglen"], logsig=g["logsig"],
                 lognoise=g["lognoise"], beta=g["beta"], Xs=g["Xs"])
        for k in g:
            if k.endswith("_params"):
                c[k] = g[k]
        cases[name] = c
    # C2-shaped: the full-size model of BASELINE configs[1], a bounded candidate sample
    rng = np.random.default_rng(0)
    N, d = 3000, 8
    X = rng.random((N, d))
    y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    Xs = rng.random((4096, d))[:64]
    cases["c2_shaped_sample"] = dict(kern="SEArd", mean="MeanConst", X=X, y=y, loglen=np.full(d, np.log(0.5)), logsig=0.0,
                                     lognoise=-2.0, beta=0.0, Xs=Xs, EI_params=[y.max()],
                                     UCB_params=[10.152008469453344], MaxMean_params=[])
    out = os.path.join(HERE, "julia_inputs.txt")
    write_cases(out, cases)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB,", len(cases), "cases")


if __name__ == "__main__":
    main()
