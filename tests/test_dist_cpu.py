"""world_size-2 gloo test of the N>1 path: candidate sharding + the single 16-byte all_gather arg-max."""
import math
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["BOHIP_ROOT"])
import bohip
from bohip.dist import allgather_best, shard_bounds
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
scores = np.load(os.environ["BOHIP_SCORES"])
lo, hi = shard_bounds(len(scores), world, rank)
loc = scores[lo:hi]
# local (value desc, index asc) arg-max exactly as k_score/k_argmax_final define it
best_v, best_i = -np.inf, -1
for i, v in enumerate(loc):
    if v > best_v: best_v, best_i = v, i
rec = torch.zeros(2, dtype=torch.int64)
rec[0] = int(np.array([best_v]).view(np.int64)[0]); rec[1] = best_i
v, i = allgather_best(rec, lo, world)
rec3 = torch.tensor([int(rec[0]), int(rec[1]), lo], dtype=torch.int64)      # offset carried inside the record
assert allgather_best(rec3, 0, world) == (v, i)
# Thompson form: S draws, every rank holds its shard's per-draw winners
from bohip.dist import allgather_best_many
S = 5
mu = scores; sig = np.abs(np.nan_to_num(np.roll(scores, 7))) + 0.1
zz = np.random.default_rng(99).standard_normal((S, len(scores)))
f = np.where(np.isnan(mu), -np.inf, mu + sig * zz)
floc = f[:, lo:hi]
tv = torch.from_numpy(floc.max(1).copy()); ti = torch.from_numpy(floc.argmax(1).astype(np.int64))
bv, bi = allgather_best_many(tv, ti, lo, world)
assert np.array_equal(bi, f.argmax(1)) and np.array_equal(bv, f.max(1))
open(os.path.join(os.environ["BOHIP_OUT"], f"rank{rank}.json"), "w").write(json.dumps({"rank": rank, "val": v, "idx": i}))
dist.destroy_process_group()
'''


def run_world(scores, world, tmp_path):
    p = tmp_path / "scores.npy"
    np.save(p, scores)
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, BOHIP_ROOT=ROOT, BOHIP_SCORES=str(p), BOHIP_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", "29571", str(w)],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    return [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(world)]


def test_two_rank_argmax_matches_single_rank(tmp_path):
    rng = np.random.default_rng(0)
    scores = rng.standard_normal(1001)
    scores[700] = scores[123] = scores.max() + 1.0        # tie across the two shards: smaller GLOBAL index must win
    scores[5] = np.nan
    recs = run_world(scores, 2, tmp_path)
    assert len(recs) == 2
    for r in recs:
        assert r["idx"] == 123 and r["val"] == scores[123]


def test_shard_bounds_cover_everything():
    from bohip.dist import shard_bounds, reduce_best

    for R in (0, 1, 7, 4096, 32768 + 3):
        for G in (1, 2, 3, 8):
            spans = [shard_bounds(R, G, g) for g in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == R
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert reduce_best([-math.inf, float("nan")], [0, 1]) == (-math.inf, -1)
    assert reduce_best([1.0, 1.0], [9, 4]) == (1.0, 4)
