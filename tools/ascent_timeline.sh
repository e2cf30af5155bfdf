#!/bin/bash
# kernel timeline of the device ascent (tools/ascent_loop_time.py under rocprofv3 --kernel-trace): the kernels of the last call in stream order
export TMPDIR=/tmp
REPO=$PWD
cd /tmp && rm -rf /tmp/atl
timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/atl -o out -- python $REPO/tools/ascent_loop_time.py > /tmp/atl.log 2>&1 || tail -3 /tmp/atl.log
grep "ascend ms" /tmp/atl.log
f=$(find /tmp/atl -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: from the last k_asc_start on
i0 = max(i for i, r in enumerate(rows) if "k_asc_start" in r["Kernel_Name"])
prev_end = None
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void bohip::", "").replace("bohip::", "")[:28]
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    print(f"{name:30s} {(e - s) / 1e3:7.1f} us   gap before {gap:6.1f} us")
    prev_end = e
PY
