#!/bin/bash
mkdir -p gpurun_out/r03
{ echo "## 8 waves"; timeout 600 python tools/knn_batch_probe.py 2000000,1024,1024,32 10000000,768,4096,32; echo "## 4 waves of 128x128"; AC_KNN_BATCH_WAVES=4 timeout 600 python tools/knn_batch_probe.py 2000000,1024,1024,32 10000000,768,4096,32; echo "## 4 waves ring 5"; AC_KNN_BATCH_WAVES=4 AC_KNN_BATCH_RING=5 timeout 600 python tools/knn_batch_probe.py 10000000,768,4096,32; } > gpurun_out/r03/knn_batch_probe8.txt 2>&1
AC_KNN_BATCH_WAVES=4 timeout 600 python -m pytest tests/test_knn_batch_gpu.py -x -q -m gpu 2>&1 | tail -2 >> gpurun_out/r03/knn_batch_probe8.txt
cat gpurun_out/r03/knn_batch_probe8.txt
