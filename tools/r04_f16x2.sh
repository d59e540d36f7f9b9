#!/bin/bash
# round 4: the opt-in fp16x2 arithmetic -- its tests, the per-shape sweep / encoder A-B, and the suites that share the changed code
mkdir -p gpurun_out/f16x2
python -m pytest tests/test_gemm_f16x2_gpu.py -q -m gpu -s -x > gpurun_out/f16x2/pytest_f16x2.txt 2>&1; echo "f16x2 tests rc=$?"
tail -5 gpurun_out/f16x2/pytest_f16x2.txt
timeout 600 python tools/f16x2_probe.py > gpurun_out/f16x2/probe_base.txt 2>&1; echo "probe rc=$?"
tail -12 gpurun_out/f16x2/probe_base.txt
python -m pytest tests/test_gemm_split_gpu.py tests/test_encoder_gpu.py tests/test_classifier_gpu.py -q -m gpu -x > gpurun_out/f16x2/pytest_shared.txt 2>&1; echo "shared rc=$?"
tail -4 gpurun_out/f16x2/pytest_shared.txt
