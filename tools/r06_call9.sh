#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
timeout 1800 python -m pytest tests/test_encoder_gpu.py tests/test_e2e_reference_gpu.py tests/test_golden_gpu.py tests/test_classifier_gpu.py tests/test_gemm_f16x2_gpu.py -x -q -m gpu 2>&1 | tail -6
{
for rnd in 1 2 3; do
  for what in base full; do
    AC_LIBACAMD_PATH=$REPO/tools/ab/libacamd_r05.so python tools/r06_encode_ab.py "r05 library" $what
    AC_LIBACAMD_PATH=$REPO/tools/ab/libacamd_r06_fp32x.so python tools/r06_encode_ab.py "r06 before planes-only activations" $what
    python tools/r06_encode_ab.py "r06, planes-only activations" $what
  done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/encode_ab6.txt
python tools/r06_layer_stamps_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/layer_stamps2.txt
