#!/bin/bash
O=gpurun_out/r02/epoch; mkdir -p $O
AC_HEAD_PERSISTENT=1 AC_HEAD_EPOCH_DEBUG=1 timeout 120 python tools/epoch_probe3.py > $O/new_dbg.log 2>&1; tail -2 $O/new_dbg.log
timeout 900 python -m pytest tests/test_head_gpu.py tests/test_multilabel_gpu.py tests/test_golden_gpu.py tests/test_classifier_gpu.py -x -q -m gpu 2>&1 | tail -3
(timeout 900 python bench.py --config add_examples --examples 6000) > $O/bench_add6000.log 2>&1; grep '^{' $O/bench_add6000.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k:(v['examples_per_s'],v['steps_per_s'],v['host_seconds_by_phase']) for k,v in d['modes'].items()})"
