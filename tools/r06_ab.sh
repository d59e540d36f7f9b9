#!/bin/bash
# round 6, first GPU call: new tests of the fused attention epilogue, the encoder test files, then the same-box A/B of three builds /
# switches: r05 library, this tree with AC_QKV_ATTN_FUSION=0 (isolates the branch-free GELU), this tree (fused attention).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
timeout 1500 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "attention_fus" 2>&1 | tail -15 > $O/pytest_attn_fusion.txt
cat $O/pytest_attn_fusion.txt
{
for rnd in 1 2; do
  for what in base full large; do
    AC_LIBACAMD_PATH=$REPO/tools/ab/libacamd_r05.so python tools/r06_encode_ab.py "r05 library" $what
    AC_QKV_ATTN_FUSION=0 python tools/r06_encode_ab.py "r06, two-launch attention" $what
    python tools/r06_encode_ab.py "r06, attention in QKV epilogue" $what
  done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/encode_ab.txt
timeout 2400 python -m pytest tests/test_encoder_gpu.py tests/test_e2e_reference_gpu.py tests/test_gemm_split_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_encoder_files.txt
cat $O/pytest_encoder_files.txt
