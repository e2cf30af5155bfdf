# varies constants of the library: needs the measurement build (make -C bayesianoptimization.jl_amd/csrc abl/libbohip_dev.so)
export BOHIP_LIB=${BOHIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/bayesianoptimization.jl_amd/csrc/abl/libbohip_dev.so}
out=gpurun_out/fill1.txt; : > $out
export BOHIP_CHOL_DF_STRICT=1
for fill in 0 1 2; do for urg in 16 24 40; do
  echo "# FILL=$fill URGENT=$urg" >> $out
  BOHIP_CHOL_EXEC_FILL=$fill BOHIP_CHOL_EXEC_URGENT=$urg timeout 300 python tools/chol_sizes.py 6000 8000 10000 12000 2>&1 | grep -v amdgpu >> $out
done; done
