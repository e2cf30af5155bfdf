#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench.py run + separate PMC passes for the dominant kernel.
# Usage (on the GPU box, from the repo root): bash tools/prof.sh <tag>
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $REPO/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- python $REPO/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $OUT/pmc_$N.log 2>&1
done
cd $REPO
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
