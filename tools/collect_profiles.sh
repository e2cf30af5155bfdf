#!/bin/bash
out=gpurun_out/r03; mkdir -p $out
bash tools/prof.sh r03 > $out/prof_stdout.txt 2>&1
cp gpurun_out/prof_r03/summary.txt $out/rocprof_bench_summary.txt 2>/dev/null
python bench.py > $out/bench_line_default.json 2> $out/bench.err
python bench.py --strong --no-cpu-baseline --no-c4 > $out/bench_line_strong_1gpu.json 2>> $out/bench.err
BOHIP_LOGICAL_SHARDS=1 python bench.py --gpus 8 --no-cpu-baseline > $out/bench_line_8_logical_shards_1gpu.json 2>> $out/bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --no-cpu-baseline --no-c4 > $out/bench_line_torchrun_1gpu.json 2>> $out/bench.err
python tools/bench_configs.py > $out/bench_configs.jsonl 2>> $out/bench.err
{
echo "# the factorisation ALONE (tools/chol_sizes.py switches the executor's inverse queues off: first dataflow form below 32 row tiles, executor form from 32 on), one process per size"
for N in 1000 2000 3000 4000 5000 6000 8000 10000 12000; do python tools/chol_sizes.py $N 2>&1 | grep -v amdgpu; done
echo "# factorisation + inverse as ONE stage (the default: executor form with its inverse queues from 4 row tiles on; flops counted = N^3/3)"
BOHIP_KEEP_INV=1 python tools/chol_sizes.py 500 1000 2000 3000 4000 5000 6000 8000 10000 12000 2>&1 | grep -v amdgpu
echo "# executor form forced from 4 row tiles, factorisation alone (BOHIP_CHOL_DATAFLOW=2 BOHIP_CHOL_EXEC_MIN=4)"
BOHIP_CHOL_DATAFLOW=2 BOHIP_CHOL_EXEC_MIN=4 python tools/chol_sizes.py 1000 2000 3000 2>&1 | grep -v amdgpu
echo "# round-2 second dataflow form, left-looking (BOHIP_CHOL_EXEC=0)"
BOHIP_CHOL_EXEC=0 python tools/chol_sizes.py 6000 8000 10000 12000 2>&1 | grep -v amdgpu
echo "# first dataflow form forced (BOHIP_CHOL_EXEC=0 BOHIP_CHOL_DATAFLOW=2 BOHIP_CHOL_DF2_MIN=999)"
BOHIP_CHOL_EXEC=0 BOHIP_CHOL_DATAFLOW=2 BOHIP_CHOL_DF2_MIN=999 python tools/chol_sizes.py 3000 4000 5000 6000 2>&1 | grep -v amdgpu
echo "# launch chain (BOHIP_CHOL_DATAFLOW=0)"
BOHIP_CHOL_DATAFLOW=0 python tools/chol_sizes.py 3000 6000 10000 2>&1 | grep -v amdgpu
echo "# full model update as shipped (tools/refit_bench.py)"
python tools/refit_bench.py 500 1000 2000 3000 4000 6000 8000 10000 2>&1 | grep -v amdgpu
echo "# full model update with the inverse queues off (BOHIP_CHOL_INV_G=0: W = L^-1 level by level after the factorisation)"
BOHIP_CHOL_INV_G=0 python tools/refit_bench.py 500 1000 2000 3000 4000 6000 8000 10000 2>&1 | grep -v amdgpu
} > $out/cholesky_by_size.txt
python tools/exec_trace.py 10000 2>/dev/null > $out/exec_trace_N10000.txt
python tools/exec_trace.py 6000 2>/dev/null > $out/exec_trace_N6000.txt
python tools/exec_trace.py 3000 2>/dev/null > $out/exec_trace_N3000.txt
BOHIP_CHOL_INV_G=0 BOHIP_CHOL_DATAFLOW=2 BOHIP_CHOL_EXEC_MIN=4 python tools/exec_trace.py 3000 2>/dev/null > $out/exec_trace_N3000_factorisation_alone.txt
BOHIP_CHOL_INV_G=0 python tools/exec_trace.py 10000 2>/dev/null > $out/exec_trace_N10000_factorisation_alone.txt
python tools/w_check.py 600 1000 3000 6000 10000 2>&1 | grep -v amdgpu > $out/w_check.txt
python tools/chol_trace.py 3000 2>&1 | grep -v amdgpu > $out/chol_form1_chain_trace_N3000.txt
python tools/ascent_bench.py 2>&1 | grep -v amdgpu > $out/ascent_bench.txt
BOHIP_ASC_LOCKSTEP=1 python tools/ascent_bench.py 2>&1 | grep -v amdgpu > $out/ascent_bench_lockstep.txt
python tools/small_batch_bench.py 2>&1 | grep -v amdgpu > $out/small_batch_bench.txt
