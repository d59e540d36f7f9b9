#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
AC_KNN_BATCH_DEBUG=1 timeout 60 python tools/knn_batch_probe.py 100000,768,256,16 > $O/knn_debug.txt 2>&1
echo "rc=$?" >> $O/knn_debug.txt
tail -12 $O/knn_debug.txt
