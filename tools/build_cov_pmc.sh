#!/bin/bash
# PMC counters of k_build_cov (N = 10000, d = 16): one rocprofv3 --pmc pass per counter group, averages per launch.  usage: bash tools/build_cov_pmc.sh outdir
out=${1:-gpurun_out/bc_pmc}; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_EA0_WRREQ_sum" "FETCH_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/$out/$tag -o pmc -- python $R/tools/build_cov_child.py 10000 16 > /dev/null 2>&1
  python - "$R/$out/$tag" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_build_cov" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print(f"{k}: mean per launch {sum(v)/len(v):.4g} over {len(v)} launches")
PY
done
