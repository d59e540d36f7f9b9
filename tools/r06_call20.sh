#!/bin/bash
# round 6, call 20: the FFN1 epilogue's GELU on the packed fp32 ALU (two evaluations per issue slot) against the scalar form
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
{
for rnd in 1 2 3; do
  for what in base full; do
    AC_LIBACAMD_PATH=$REPO/tools/ab/libacamd_r06_scalar_gelu.so python tools/r06_encode_ab.py "scalar GELU" $what
    python tools/r06_encode_ab.py "packed-fp32 GELU" $what
  done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/encode_ab_gelu2.txt
timeout 2400 python -m pytest tests/test_encoder_gpu.py tests/test_e2e_reference_gpu.py tests/test_golden_gpu.py tests/test_gemm_split_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest_gelu2.txt
python tools/r06_layer_stamps_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee $O/layer_stamps_gelu2.txt
