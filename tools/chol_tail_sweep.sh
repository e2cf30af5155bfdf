# tail-of-the-factorisation knobs of the measurement build (abl/libbohip_dev.so), the factorisation ALONE at N = 8000 ... 12000
export BOHIP_LIB=$PWD/bayesianoptimization.jl_amd/csrc/abl/libbohip_dev.so
run() { echo "== $*"; env "$@" python tools/chol_sizes.py ${SIZES:-8000 10000 12000} 2>&1 | grep -v amdgpu; }
for kv in ${KNOBS:-X=0 BOHIP_CHOL_EXEC_EARLY_TAIL=24 BOHIP_CHOL_EXEC_EARLY_TAIL=40 BOHIP_CHOL_EXEC_EARLY_TAIL=56 BOHIP_CHOL_EXEC_NBU=0 BOHIP_CHOL_EXEC_NBU=4 BOHIP_CHOL_EXEC_NBU=6 BOHIP_CHOL_NSF=4 BOHIP_CHOL_NSF=5 BOHIP_CHOL_EXEC_URGENT=64 BOHIP_CHOL_EXEC_PATIENCE_US=100 X=0}; do run $kv; done
