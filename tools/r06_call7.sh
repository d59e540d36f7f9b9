#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
python tools/r06_attn_stamps_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/attn_stamps.txt
timeout 1500 python -m pytest tests/test_encoder_gpu.py tests/test_e2e_reference_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -6
{
for rnd in 1 2; do
  for what in base full; do
    AC_LIBACAMD_PATH=$REPO/tools/ab/libacamd_r05.so python tools/r06_encode_ab.py "r05 library" $what
    AC_QKV_ATTN_FUSION=0 python tools/r06_encode_ab.py "r06, two-launch attention" $what
    AC_QKV_ATTN_EXCHANGE=0 python tools/r06_encode_ab.py "r06, fused + boundary launch" $what
    python tools/r06_encode_ab.py "r06, fused + in-launch exchange" $what
  done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/encode_ab5.txt
