#!/bin/bash
# kernel-trace statistics of the small-batch pass (tools/small_pass_profile.py: 200 score_grad + 200 score calls of 10 candidates) at the given sizes.
# Usage (GPU box, repo root): bash tools/small_pass_prof.sh "3000 8" "500 2" ...   (always under `timeout`, always --output-format csv)
# BOHIP_SMALL_M (tile depth override) is read by the measurement build only (abl/libbohip_dev.so through BOHIP_LIB)
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for a in "$@"; do
  set -- $a
  rm -rf /tmp/sprof
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sprof -o out -- python $REPO/tools/small_pass_profile.py $1 $2 > /tmp/sprof.log 2>&1 || { echo "rocprofv3 failed"; tail -5 /tmp/sprof.log; }
  echo "== N=$1 d=$2 ${BOHIP_SMALL_M:+m=$BOHIP_SMALL_M}"
  f=$(find /tmp/sprof -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("k_small", "k_kstar", "k_trimv", "k_grad", "k_asc")):
        print(f'  {n[:60]:60s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"])/1e3:7.2f} us  min {float(r["MinNs"])/1e3:7.2f}  max {float(r["MaxNs"])/1e3:7.2f}')
PY
done
