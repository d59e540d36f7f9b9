#!/bin/bash
# Round-2 closing run (GPU box): full GPU test suite, the bench lines, rocprofv3 kernel stats of the default bench command.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r02/final; mkdir -p $O
cd $REPO
(time timeout 1500 python -m pytest tests -x -q -m gpu) > $O/tests_gpu.log 2>&1; tail -3 $O/tests_gpu.log
(time timeout 900 python bench.py) > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/bench_line.json; tail -3 $O/bench_default.log | cut -c1-200
(time timeout 600 python bench.py --config cfg4 --steps 10 --warmup 2) > $O/bench_cfg4.log 2>&1; grep '^{' $O/bench_cfg4.log > $O/bench_line_cfg4.json; tail -3 $O/bench_cfg4.log | cut -c1-200
(time AC_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline) > $O/bench_2proc_gloo.log 2>&1; grep '^{' $O/bench_2proc_gloo.log > $O/bench_line_2proc_gloo_one_gpu.json; tail -2 $O/bench_2proc_gloo.log | cut -c1-200
timeout 300 python tools/latency_probe.py > $O/latency.log 2>&1; grep "single predict" $O/latency.log
cd /tmp && export TMPDIR=/tmp; T=/tmp/prof_final; rm -rf $T
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $T -o bench -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.txt 2>&1
cp $(find $T -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; grep '^{' $O/bench_under_rocprof.txt > $O/bench_line_under_rocprof.json
head -6 $O/bench_kernel_stats.csv | cut -c1-140
