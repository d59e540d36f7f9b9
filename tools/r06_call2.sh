#!/bin/bash
# round 6, GPU call 2: fused-attention tests (bit-identity after pinning fp contraction), the training differential, A/B, encoder files
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
timeout 1500 python -m pytest tests/test_encoder_gpu.py -q -m gpu -k "attention_fus" 2>&1 | tail -25 > $O/pytest_attn_fusion.txt
cat $O/pytest_attn_fusion.txt
timeout 1500 python -m pytest tests/test_e2e_reference_gpu.py -q -m gpu -k "training_trajectory or dropout_source" -s 2>&1 | tail -40 > $O/pytest_training_differential.txt
cat $O/pytest_training_differential.txt
{
for rnd in 1 2; do
  for what in base full; do
    AC_LIBACAMD_PATH=$REPO/tools/ab/libacamd_r05.so python tools/r06_encode_ab.py "r05 library" $what
    AC_QKV_ATTN_FUSION=0 python tools/r06_encode_ab.py "r06, two-launch attention" $what
    python tools/r06_encode_ab.py "r06, attention in QKV epilogue" $what
  done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/encode_ab2.txt
timeout 2400 python -m pytest tests/test_encoder_gpu.py tests/test_e2e_reference_gpu.py tests/test_gemm_split_gpu.py tests/test_golden_gpu.py tests/test_gemm_f16x2_gpu.py tests/test_classifier_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_encoder_files.txt
cat $O/pytest_encoder_files.txt
