#!/bin/bash
# Round-2 bench lines (run on the GPU box via gpurun); each JSON line lands in gpurun_out/r02/lines/.
O=gpurun_out/r02/lines; mkdir -p $O
(time timeout 900 python bench.py) > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/bench_line.json; tail -3 $O/bench_default.log | cut -c1-300
(time timeout 600 python bench.py --config cfg4 --steps 10 --warmup 2) > $O/bench_cfg4.log 2>&1; grep '^{' $O/bench_cfg4.log > $O/bench_line_cfg4.json; tail -3 $O/bench_cfg4.log | cut -c1-300
(time timeout 1200 python bench.py --config add_examples --examples 50000) > $O/bench_add50k.log 2>&1; grep '^{' $O/bench_add50k.log > $O/bench_line_add_examples_50k.json; tail -3 $O/bench_add50k.log | cut -c1-600
(time AC_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline) > $O/bench_2proc_gloo.log 2>&1; grep '^{' $O/bench_2proc_gloo.log > $O/bench_line_2proc_gloo_one_gpu.json; tail -3 $O/bench_2proc_gloo.log | cut -c1-300
