#!/bin/bash
# round 6, GPU call 5: in-launch exchange of straddling sequences (tests + A/B), fixed tests, profiles
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
timeout 1500 python -m pytest tests/test_encoder_gpu.py -q -m gpu -k "attention_fus" 2>&1 | tail -25 > $O/pytest_attn_fusion.txt
cat $O/pytest_attn_fusion.txt
timeout 1500 python -m pytest tests/test_reference_suite_gpu.py tests/test_e2e_reference_gpu.py tests/test_sharded_gpu.py -q -m gpu -k "itself_fails or training_trajectory or two_processes or rccl" -s 2>&1 | tail -30 > $O/pytest_fixed.txt
cat $O/pytest_fixed.txt
timeout 900 python -m pytest tests/test_classifier_gpu.py -q -m gpu -k "residency" -s 2>&1 | tail -12 > $O/pytest_residency.txt
cat $O/pytest_residency.txt
{
for rnd in 1 2; do
  for what in base full; do
    AC_LIBACAMD_PATH=$REPO/tools/ab/libacamd_r05.so python tools/r06_encode_ab.py "r05 library" $what
    AC_QKV_ATTN_FUSION=0 python tools/r06_encode_ab.py "r06, two-launch attention" $what
    AC_QKV_ATTN_EXCHANGE=0 python tools/r06_encode_ab.py "r06, fused + boundary launch" $what
    python tools/r06_encode_ab.py "r06, fused + in-launch exchange" $what
  done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/encode_ab3.txt
bash tools/r06_profiles.sh 2>&1 | tail -60
