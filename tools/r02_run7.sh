#!/bin/bash
mkdir -p gpurun_out/r02
(time timeout 900 python -m pytest tests/test_classifier_gpu.py tests/test_golden_gpu.py tests/test_multilabel_gpu.py -x -q -m gpu) > gpurun_out/r02/tests_graph.log 2>&1; tail -12 gpurun_out/r02/tests_graph.log
(time timeout 300 python tools/latency_probe2.py) > gpurun_out/r02/latency2.log 2>&1; tail -8 gpurun_out/r02/latency2.log
(time timeout 300 python tools/knn_probe2.py) 2>&1 | head -2
