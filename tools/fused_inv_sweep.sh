#!/bin/bash
# knob sweep of the executor form with its inverse queues.  usage (GPU box): bash tools/fused_inv_sweep.sh OUTDIR
# varies constants of the library: needs the measurement build (make -C bayesianoptimization.jl_amd/csrc abl/libbohip_dev.so)
export BOHIP_LIB=${BOHIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/bayesianoptimization.jl_amd/csrc/abl/libbohip_dev.so}
out=${1:-gpurun_out/finvs}; mkdir -p $out
export BOHIP_CHOL_DF_STRICT=1 BOHIP_CHOL_DATAFLOW=2 BOHIP_CHOL_EXEC_MIN=4
run() { echo "# $*" >> $out/sweep.txt; env "$@" timeout 600 python tools/refit_bench.py $SIZES 2>&1 | grep -v amdgpu.ids >> $out/sweep.txt; }
SIZES="500 700 1000 1500 2000"
run BOHIP_CHOL_INV_G=8
run BOHIP_CHOL_EXEC_MIN=999
SIZES="3000 6000 10000"
run BOHIP_CHOL_INV_G=8
run BOHIP_CHOL_INV_G=8 BOHIP_CHOL_EXEC_URGENT=16
run BOHIP_CHOL_INV_G=8 BOHIP_CHOL_EXEC_URGENT=8
run BOHIP_CHOL_INV_G=8 BOHIP_CHOL_EXEC_INV_PAIRS=1
run BOHIP_CHOL_INV_G=6
run BOHIP_CHOL_INV_G=12
run BOHIP_CHOL_INV_G=8 BOHIP_CHOL_EXEC_FILL=1
run BOHIP_CHOL_INV_G=8 BOHIP_CHOL_EXEC_FILL=2
run BOHIP_CHOL_INV_G=8 BOHIP_CHOL_NSF=2
run BOHIP_CHOL_INV_G=8 BOHIP_CHOL_EXEC_WGS=496
