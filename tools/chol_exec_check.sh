#!/bin/bash
# executor form of the factorisation (csrc/kernels_exec.hip) against the launch chain: factor, repeatability, time by size.
# usage (GPU box): bash tools/chol_exec_check.sh OUTDIR
# varies constants of the library: needs the measurement build (make -C bayesianoptimization.jl_amd/csrc abl/libbohip_dev.so)
export BOHIP_LIB=${BOHIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/bayesianoptimization.jl_amd/csrc/abl/libbohip_dev.so}
out=${1:-gpurun_out/exec}; mkdir -p $out
export BOHIP_CHOL_DF_STRICT=1
for N in 1000 3000 6000 10000; do
  rm -f /tmp/ref_$N.npy /tmp/ref_$N.npy.mu.npy
  BOHIP_CHOL_DATAFLOW=0 timeout 300 python tools/chol_compare.py $N /tmp/ref_$N.npy >> $out/compare.txt 2>&1
  BOHIP_CHOL_DATAFLOW=2 BOHIP_CHOL_EXEC=1 BOHIP_CHOL_EXEC_MIN=4 timeout 300 python tools/chol_compare.py $N /tmp/ref_$N.npy 3 >> $out/compare.txt 2>&1
done
BOHIP_CHOL_DATAFLOW=2 BOHIP_CHOL_EXEC=1 BOHIP_CHOL_EXEC_MIN=4 timeout 300 python tools/chol_stress.py 5000 6 > $out/stress.txt 2>&1
echo "# executor default (T >= 47)" > $out/sizes.txt
timeout 600 python tools/chol_sizes.py 6000 8000 10000 12000 >> $out/sizes.txt 2>&1
echo "# executor forced from 4 row tiles" >> $out/sizes.txt
BOHIP_CHOL_DATAFLOW=2 BOHIP_CHOL_EXEC_MIN=4 timeout 600 python tools/chol_sizes.py 1000 2000 3000 4000 5000 >> $out/sizes.txt 2>&1
echo "# BOHIP_CHOL_EXEC=0 (stream-based second form)" >> $out/sizes.txt
BOHIP_CHOL_EXEC=0 timeout 600 python tools/chol_sizes.py 8000 10000 >> $out/sizes.txt 2>&1
echo "# executor, BOHIP_CHOL_EXEC_FILL=0" >> $out/sizes.txt
BOHIP_CHOL_EXEC_FILL=0 timeout 600 python tools/chol_sizes.py 6000 10000 >> $out/sizes.txt 2>&1
echo "# executor, BOHIP_CHOL_EXEC_PAIRS=0" >> $out/sizes.txt
BOHIP_CHOL_EXEC_PAIRS=0 timeout 600 python tools/chol_sizes.py 6000 10000 >> $out/sizes.txt 2>&1
