#!/bin/bash
O=gpurun_out/r02/epoch; mkdir -p $O
AC_HEAD_EPOCH_DEBUG=1 timeout 120 python tools/epoch_probe3.py > $O/new_dbg.log 2>&1; tail -2 $O/new_dbg.log
timeout 120 python tools/epoch_probe3.py ewc > $O/new_ewc.log 2>&1; tail -1 $O/new_ewc.log
timeout 600 python -m pytest tests/test_head_gpu.py tests/test_multilabel_gpu.py tests/test_golden_gpu.py tests/test_classifier_gpu.py -x -q -m gpu 2>&1 | tail -2
