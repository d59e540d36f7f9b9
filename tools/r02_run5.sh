#!/bin/bash
mkdir -p gpurun_out/r02
(time timeout 900 python -m pytest tests/test_knn_batch_gpu.py -x -q -m gpu -s) > gpurun_out/r02/tests_batch.log 2>&1; tail -25 gpurun_out/r02/tests_batch.log
(time timeout 600 python tools/knn_probe2.py) > gpurun_out/r02/knn_probe2.log 2>&1; tail -8 gpurun_out/r02/knn_probe2.log
