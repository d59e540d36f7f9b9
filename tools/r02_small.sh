#!/bin/bash
O=gpurun_out/r02/small; mkdir -p $O
timeout 600 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "small_batch or bert-tiny or 128-2-2" 2>&1 | tail -8
timeout 120 python tools/single_query_probe.py > $O/single.log 2>&1; tail -12 $O/single.log
