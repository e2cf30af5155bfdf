#!/bin/bash
# sweep of the executor form's knobs.  usage (GPU box): bash tools/chol_exec_sweep.sh OUTFILE
# varies constants of the library: needs the measurement build (make -C bayesianoptimization.jl_amd/csrc abl/libbohip_dev.so)
export BOHIP_LIB=${BOHIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/bayesianoptimization.jl_amd/csrc/abl/libbohip_dev.so}
out=${1:-gpurun_out/sweep.txt}; mkdir -p $(dirname $out); : > $out
export BOHIP_CHOL_DF_STRICT=1
for nsf in 2 3 4 5; do for pairs in 1 2; do
  echo "# BOHIP_CHOL_NSF=$nsf BOHIP_CHOL_EXEC_PAIRS=$pairs" >> $out
  BOHIP_CHOL_NSF=$nsf BOHIP_CHOL_EXEC_PAIRS=$pairs BOHIP_CHOL_DATAFLOW=2 BOHIP_CHOL_EXEC_MIN=4 timeout 300 python tools/chol_sizes.py 3000 4000 6000 8000 10000 2>&1 | grep -v amdgpu >> $out
done; done
