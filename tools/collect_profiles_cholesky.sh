out=gpurun_out/r06b; mkdir -p $out
python bench.py > $out/bench_line_default.json 2> $out/bench.err
{
echo "# the factorisation ALONE (tools/chol_sizes.py switches the executor's inverse queues off), one process"
python tools/chol_sizes.py 1000 2000 3000 4000 6000 8000 10000 12000 2>&1 | grep -v amdgpu
echo "# factorisation + inverse as ONE stage (the default; flops counted = N^3/3)"
BOHIP_KEEP_INV=1 python tools/chol_sizes.py 500 1000 2000 3000 4000 6000 8000 10000 12000 2>&1 | grep -v amdgpu
echo "# full model update as shipped (tools/refit_bench.py)"
python tools/refit_bench.py 500 1000 3000 6000 10000 2>&1 | grep -v amdgpu
} > $out/cholesky_by_size.txt
timeout 200 python tools/exec_trace.py 3000 2>&1 | grep -v amdgpu > $out/exec_trace_N3000.txt
BOHIP_CHOL_INV_G=0 timeout 200 python tools/exec_trace.py 3000 2>&1 | grep -v amdgpu > $out/exec_trace_N3000_alone.txt
timeout 300 python tools/exec_trace.py 10000 2>&1 | grep -v amdgpu > $out/exec_trace_N10000.txt
{
for N in 10000 7000 3000 1000; do echo "== N=$N"; timeout 600 python tools/chol_soak.py $N 12 8 2>&1 | grep -v amdgpu | tail -1; done
echo "== factorisation alone (BOHIP_CHOL_INV_G=0), N=3000 and 6000"
BOHIP_CHOL_INV_G=0 timeout 600 python tools/chol_soak.py 3000 12 8 2>&1 | grep -v amdgpu | tail -1
BOHIP_CHOL_INV_G=0 timeout 600 python tools/chol_soak.py 6000 12 8 2>&1 | grep -v amdgpu | tail -1
echo "== two processes refitting at once on one device, N=3000, 8 refits each, three rounds (tools/two_process_soak.sh)"
bash tools/two_process_soak.sh 2>&1 | grep -v "^.: process" 
} > $out/soak.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --durations=10 2>&1 | grep -v amdgpu > $out/pytest_gpu.txt
