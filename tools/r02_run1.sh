#!/bin/bash
# round-2 first GPU pass: new parity tests, sharded tests, twostream probe, bench line
set -x
mkdir -p gpurun_out/r02
(time timeout 900 python -m pytest tests/test_knn_baseline_gpu.py tests/test_sharded_gpu.py tests/test_golden_gpu.py tests/test_knn_gpu.py -x -q -m gpu -s --durations=12) > gpurun_out/r02/tests_new.log 2>&1
tail -30 gpurun_out/r02/tests_new.log
(time timeout 300 python tools/twostream_probe.py) > gpurun_out/r02/twostream.log 2>&1
tail -12 gpurun_out/r02/twostream.log
(time timeout 600 python bench.py --steps 20 --warmup 3) > gpurun_out/r02/bench1.log 2>&1
tail -c 3000 gpurun_out/r02/bench1.log
