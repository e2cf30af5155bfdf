#!/bin/bash
# End-to-end check of bench.py's N > 1 path on a ONE-GPU box: 2 and 4 ranks share GPU 0 (gloo exchange) and must
# report the same winner as one rank scoring the whole candidate set.
set -e
for n in 2 4; do
  BOHIP_SHARE_GPU=1 BOHIP_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
     --master-port $((29600+n)) bench.py --gpus $n --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ranks', d['n_gpus'], 'R_total', d['config']['R_total'], 'best', d['best'])"
done
python - <<'PY'
import sys; sys.path.insert(0, '.')
import numpy as np, bench, bohip
X, y = bench.synth(0); ll = np.full(bench.DIM, np.log(0.5))
m = bohip.ElasticGPE(bench.DIM, mean=bohip.MeanConst(0.0), kernel=bohip.SEArd(ll, 0.0), logNoise=-2.0, capacity=bench.N_OBS)
m.append_(X.T, y)
for n in (2, 4):
    Xs = bench.lhs(bench.R_PER_GPU * n, seed=1)
    _, bv, bi = m.score("EI", [float(y.max())], Xs.T)
    print('single rank, R_total', bench.R_PER_GPU * n, 'best', {'value': bv, 'index': bi})
PY
