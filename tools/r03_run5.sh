#!/bin/bash
# round 3, GPU call 5: the one-product fp16 batched kNN sweep: tests, timings, per-kernel split
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_knn_batch_gpu.py tests/test_knn_baseline_gpu.py tests/test_sharded_gpu.py tests/test_golden_gpu.py -x -q -m gpu -s > gpurun_out/r03/pytest_run5.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03/pytest_run5.log
timeout 600 python tools/knn_batch_probe.py > gpurun_out/r03/knn_batch_probe5.txt 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_knn -- python $GRAFT_REPO_ROOT/tools/knn_batch_probe.py 10000000,768,4096,32 > /tmp/prof_knn.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_knn -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r03/knn_batch_kernel_stats5.csv
tail -5 gpurun_out/r03/pytest_run5.log; cat gpurun_out/r03/knn_batch_probe5.txt; head -12 gpurun_out/r03/knn_batch_kernel_stats5.csv | cut -c1-200
