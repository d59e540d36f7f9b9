#!/bin/bash
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_knn_batch_gpu.py tests/test_knn_baseline_gpu.py -x -q -m gpu > gpurun_out/r03/pytest_run7.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03/pytest_run7.log
{ echo "## ring 4"; timeout 600 python tools/knn_batch_probe.py; echo "## ring 5"; AC_KNN_BATCH_RING=5 timeout 600 python tools/knn_batch_probe.py; } > gpurun_out/r03/knn_batch_probe7.txt 2>&1
tail -3 gpurun_out/r03/pytest_run7.log; cat gpurun_out/r03/knn_batch_probe7.txt
