#!/bin/bash
O=gpurun_out/r02/full; mkdir -p $O
(time timeout 1500 python -m pytest tests -x -q -m gpu) > $O/tests_gpu.log 2>&1; tail -4 $O/tests_gpu.log
timeout 300 python tools/latency_probe.py > $O/latency.log 2>&1; grep "single predict\|train step" $O/latency.log
AC_BERT_SMALL=0 timeout 300 python tools/latency_probe.py > $O/latency_old.log 2>&1; grep "single predict" $O/latency_old.log
