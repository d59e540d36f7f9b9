#!/bin/bash
mkdir -p gpurun_out/r02
(time timeout 600 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "1024-24") > gpurun_out/r02/tests_large.log 2>&1; tail -4 gpurun_out/r02/tests_large.log
(time timeout 600 python bench.py --config cfg4 --steps 5 --warmup 2) > gpurun_out/r02/bench_cfg4.log 2>&1; tail -c 1800 gpurun_out/r02/bench_cfg4.log
(time timeout 900 python bench.py --config add_examples --examples 6000) > gpurun_out/r02/bench_add6000.log 2>&1; tail -c 2500 gpurun_out/r02/bench_add6000.log
(time AC_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --sweep-rows 2000000 --no-cpu-baseline) > gpurun_out/r02/bench_2proc_gloo.log 2>&1; tail -c 1500 gpurun_out/r02/bench_2proc_gloo.log
