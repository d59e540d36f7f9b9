#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
timeout 1500 python -m pytest tests/test_encoder_gpu.py -q -m gpu -k "attention_fus or modernbert_gemm_arith" 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_reference_suite_gpu.py -q -m gpu -k "itself_fails" -s 2>&1 | tail -12
{
for rnd in 1 2; do
  for what in base full; do
    AC_QKV_ATTN_FUSION=0 python tools/r06_encode_ab.py "r06, two-launch attention" $what
    AC_QKV_ATTN_EXCHANGE=0 python tools/r06_encode_ab.py "r06, fused + boundary launch" $what
    python tools/r06_encode_ab.py "r06, fused + in-launch exchange" $what
  done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/encode_ab4.txt
