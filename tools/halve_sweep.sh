#!/bin/bash
# k_trigemm_sq with different sets of row tiles issued as two 64-row halves (BOHIP_TRIGEMM_HALVE_LO / _HI; "0,0" = whole tiles only,
# the schedule of rounds 1-3).  Prints the dominant kernel's event time per setting (tools/power_probe.py, line (a)).
# varies constants of the library: needs the measurement build (make -C bayesianoptimization.jl_amd/csrc abl/libbohip_dev.so)
export BOHIP_LIB=${BOHIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/bayesianoptimization.jl_amd/csrc/abl/libbohip_dev.so}
for h in "0,0" "0,8" "0,4" "0,6" "0,10" "0,12" "0,16" "2,10" "4,12" "0,24" "0,0" "0,8"; do
  echo -n "HALVE=$h  "
  BOHIP_TRIGEMM_HALVE_LO=${h%,*} BOHIP_TRIGEMM_HALVE_HI=${h#*,} python tools/power_probe.py ${1:-4096} 2>/dev/null | grep "^(a)"
done
