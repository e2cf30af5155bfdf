#!/bin/bash
# everything copied to profiles/<tag>_* in one GPU call (tag = $1, default r06)
TAG=${1:-r06}
out=gpurun_out/$TAG; mkdir -p $out
bash tools/prof.sh $TAG > $out/prof_stdout.txt 2>&1
cp gpurun_out/prof_$TAG/summary.txt $out/rocprof_bench_summary.txt 2>/dev/null
cp gpurun_out/prof_$TAG/traffic_trigemm_sq.json $out/traffic_trigemm_sq.json 2>/dev/null
rm -rf gpurun_out/prof_$TAG/kt gpurun_out/prof_$TAG/pmc_*   # (raw traces: tens of MB; gpurun merges at most 64 MiB back)
python bench.py > $out/bench_line_default.json 2> $out/bench.err
python bench.py --strong --no-cpu-baseline --no-c4 > $out/bench_line_strong_1gpu.json 2>> $out/bench.err
BOHIP_LOGICAL_SHARDS=1 python bench.py --gpus 8 --no-cpu-baseline > $out/bench_line_8_logical_shards_1gpu.json 2>> $out/bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --no-cpu-baseline --no-c4 > $out/bench_line_torchrun_1gpu.json 2>> $out/bench.err
python tools/bench_configs.py > $out/bench_configs.jsonl 2>> $out/bench.err
{
echo "# the factorisation ALONE (tools/chol_sizes.py switches the executor's inverse queues off), one process"
python tools/chol_sizes.py 1000 2000 3000 4000 6000 8000 10000 12000 2>&1 | grep -v amdgpu
echo "# factorisation + inverse as ONE stage (the default; flops counted = N^3/3)"
BOHIP_KEEP_INV=1 python tools/chol_sizes.py 500 1000 2000 3000 4000 6000 8000 10000 12000 2>&1 | grep -v amdgpu
echo "# full model update as shipped (tools/refit_bench.py)"
python tools/refit_bench.py 500 1000 3000 6000 10000 2>&1 | grep -v amdgpu
} > $out/cholesky_by_size.txt
python tools/ascent_bench.py 2>&1 | grep -v amdgpu > $out/ascent_bench.txt
python tools/small_batch_bench.py 2>&1 | grep -v amdgpu > $out/small_batch_bench.txt
python tools/power_probe.py 4096 2>&1 | grep -v amdgpu > $out/power_probe.txt
make -C bayesianoptimization.jl_amd/csrc abl/libbohip_trace.so > /dev/null 2>&1
python tools/trace_trigemm.py 2>&1 | grep -v amdgpu > $out/trigemm_workgroup_timeline.txt
# round 5: the small-batch pass (kernels_small.hip): kernel trace statistics and the in-kernel timeline; the full GPU test suite's log
{
bash tools/small_pass_prof.sh "3000 8" "500 2" "1000 4" "10000 16"
echo "# the same with round 4's five-kernel pass (BOHIP_SMALL_MFMA=0)"
BOHIP_SMALL_MFMA=0 bash tools/small_pass_prof.sh "3000 8" "10000 16"
} > $out/small_pass_kernel_stats.txt 2>&1
make -C bayesianoptimization.jl_amd/csrc abl/libbohip_smalltrace.so > /dev/null 2>&1
{ timeout 100 python tools/small_pass_trace.py 3000 8; timeout 100 python tools/small_pass_trace.py 500 2; timeout 100 python tools/small_pass_trace.py 10000 16; } 2>&1 | grep -v amdgpu > $out/small_pass_trace.txt
timeout 1500 python -m pytest tests -m gpu -q -s --durations=15 2>&1 | grep -v amdgpu > $out/pytest_gpu.txt
# soak of the dataflow forms (12 fresh processes x 8 refits) and W under 8 concurrent refits (the cross-process lock is non-blocking since round 5)
{
echo "tools/chol_soak.py N 12 8 (12 fresh processes x 8 refits each): the last process's stage times and the count of processes that saw a time-out"
for N in 10000 7000 3000 1000; do echo "== N=$N"; timeout 600 python tools/chol_soak.py $N 12 8 2>&1 | grep -v amdgpu | tail -1; timeout 600 python tools/chol_soak.py $N 12 8 2>&1 | grep -c "fall-backs 0" | sed 's/$/ of 12 processes without a time-out/'; done
echo "== two processes refitting at once on one device (the non-blocking file lock), N=3000, 8 refits each"
(timeout 300 python tools/chol_soak.py 3000 4 8 2>&1 | grep -v amdgpu | sed 's/^/A: /' &) ; timeout 300 python tools/chol_soak.py 3000 4 8 2>&1 | grep -v amdgpu | sed 's/^/B: /'; wait
echo "== tools/w_stress.py 3000 8 8"
timeout 600 python tools/w_stress.py 3000 8 8 2>&1 | grep -v amdgpu
} > $out/soak.txt 2>&1
# round 6: the factorisation's timelines (executor form, in-kernel marks), the ascent against SciPy start by start and its KKT margins,
# the ascent's kernel timeline with the step folded into the gradient kernel (default) and as a launch of its own (BOHIP_ASC_LOCKSTEP=2)
make -C bayesianoptimization.jl_amd/csrc abl/libbohip_choltrace.so > /dev/null 2>&1
timeout 200 python tools/exec_trace.py 3000 2>&1 | grep -v amdgpu > $out/exec_trace_N3000.txt
timeout 300 python tools/exec_trace.py 10000 2>&1 | grep -v amdgpu > $out/exec_trace_N10000.txt
timeout 900 python tools/ascent_kkt_margin.py 2>&1 | grep -v amdgpu > $out/ascent_kkt_margin.txt
timeout 900 python tools/ascent_vs_scipy_starts.py 2>&1 | grep -v amdgpu > $out/ascent_vs_scipy_starts.txt
{
echo "# step folded into k_small_u's last workgroup (default)"
bash tools/ascent_timeline.sh
echo "# step as a launch of its own (BOHIP_ASC_LOCKSTEP=2, rounds 4-5)"
BOHIP_ASC_LOCKSTEP=2 bash tools/ascent_timeline.sh
echo "# 40 back-to-back acquire_max, three runs each: folded / own launch"
for i in 1 2 3; do python tools/ascent_loop_time.py; BOHIP_ASC_LOCKSTEP=2 python tools/ascent_loop_time.py; done 2>&1 | grep ascend
} > $out/ascent_kernel_timeline.txt 2>&1
