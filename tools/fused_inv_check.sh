#!/bin/bash
# The executor's inverse queues (W = L^-1 grown behind the chain): correctness of W / W' / alpha and the refit time with and without them.
# usage (GPU box): bash tools/fused_inv_check.sh OUTDIR [quick]
out=${1:-gpurun_out/finv}; mkdir -p $out
export BOHIP_CHOL_DF_STRICT=1
echo "# forced executor from 4 row tiles, BOHIP_CHOL_INV_G default" > $out/wcheck.txt
BOHIP_CHOL_DATAFLOW=2 BOHIP_CHOL_EXEC_MIN=4 timeout 600 python tools/w_check.py 600 1000 2100 3000 6000 >> $out/wcheck.txt 2>&1
echo "# BOHIP_CHOL_INV_G=2" >> $out/wcheck.txt
BOHIP_CHOL_INV_G=2 BOHIP_CHOL_DATAFLOW=2 BOHIP_CHOL_EXEC_MIN=4 timeout 600 python tools/w_check.py 1000 3000 >> $out/wcheck.txt 2>&1
for g in ${GS:-0 8 4 16}; do
  echo "# refit, executor forced from 4 row tiles, BOHIP_CHOL_INV_G=$g" >> $out/refit.txt
  BOHIP_CHOL_INV_G=$g BOHIP_CHOL_DATAFLOW=2 BOHIP_CHOL_EXEC_MIN=4 timeout 900 python tools/refit_bench.py ${SIZES:-1000 2000 3000 4000 6000 8000 10000} >> $out/refit.txt 2>&1
done
