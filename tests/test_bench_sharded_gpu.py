"""GPU test of bench.py's N > 1 path on a one-GPU box: two ranks share GPU 0, exchange their 24-byte records through
gloo, and must report the same (value, global index) as one rank scoring the whole candidate set."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_ranks_sharing_one_gpu_match_single_rank():
    env = dict(os.environ, BOHIP_SHARE_GPU="1", BOHIP_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["R_total"] == 8192 and line["scaling"] == "weak"
    for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in line
    assert line["roofline"]["bound"] == "mfma" and 0 < line["roofline"]["frac"] < 1

    sys.path.insert(0, ROOT)
    import bench
    import bohip

    X, y = bench.synth(0)
    ll = np.full(bench.DIM, np.log(0.5))
    m = bohip.ElasticGPE(bench.DIM, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(ll, 0.0), logNoise=-2.0, capacity=bench.N_OBS)
    m.append_(X.T, y)
    _, bv, bi = m.score("EI", [float(y.max())], bench.lhs(8192, seed=1).T)
    # two processes refit on ONE GPU at the same moment here: if one of them timed out on the dataflow factorisation (bounded
    # waits, tests/test_hardening_gpu.py) it fell back to the launch-chained form, whose factor differs in the last bits
    fell_back = "timed out on a dependency" in out.stderr
    assert line["best"]["index"] == bi
    if fell_back:
        assert line["best"]["value"] == pytest.approx(bv, rel=1e-9)
    else:
        assert line["best"] == {"value": bv, "index": bi}
