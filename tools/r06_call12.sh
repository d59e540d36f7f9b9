#!/bin/bash
# round 6, call 12: the few-tile fp32 kernel: its test, the head / CLS-row shapes timed with and without it, the encoder's last layer with
# it (AC_TAIL_FEWTILES=1) against split-K over planes, head + classifier test files
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
timeout 600 python -m pytest tests/test_gemm_split_gpu.py -x -q -m gpu -k few_tile 2>&1 | tail -5 | tee $O/pytest_fewtiles.txt
python tools/r06_head_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/head_probe.txt
{
for rnd in 1 2; do
  python tools/r06_encode_ab.py "tail: split-K over planes" base
  AC_TAIL_FEWTILES=1 python tools/r06_encode_ab.py "tail: few-tile fp32 kernel" base
done
} 2>&1 | grep -v amdgpu.ids | tee $O/encode_ab_tail.txt
python tools/r06_step_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/step_host_phases_2.txt
timeout 1800 python -m pytest tests/test_head_gpu.py tests/test_classifier_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_head_files.txt
