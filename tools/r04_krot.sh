#!/bin/bash
# round 4: the rotated k order as the default -- the suites that share the code, the per-shape sweep under it, the bench line
mkdir -p gpurun_out/krot
python -m pytest tests/test_gemm_split_gpu.py tests/test_gemm_f16x2_gpu.py tests/test_encoder_gpu.py -q -m gpu -s > gpurun_out/krot/pytest.txt 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/krot/pytest.txt; grep -A1 "peaked=" gpurun_out/krot/pytest.txt | cut -c1-330
timeout 600 python tools/f16x2_probe.py > gpurun_out/krot/probe_base_krot.txt 2>&1; echo "probe rc=$?"
cut -c1-420 gpurun_out/krot/probe_base_krot.txt | tail -10
