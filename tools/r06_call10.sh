#!/bin/bash
# round 6, call 10: the fence-free in-launch exchanges (LnFuse::fences / AttnFuse::fences = 0, the default) against the release /
# acquire form (AC_EXCHANGE_FENCES=1) of the same build, interleaved, same box; then the test files that cover the exchanges.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
{
for rnd in 1 2 3; do
  for what in base full large; do
    AC_EXCHANGE_FENCES=1 python tools/r06_encode_ab.py "release/acquire exchanges" $what
    python tools/r06_encode_ab.py "fence-free exchanges" $what
  done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/encode_ab_fences.txt
timeout 2400 python -m pytest tests/test_encoder_gpu.py tests/test_e2e_reference_gpu.py tests/test_gemm_split_gpu.py tests/test_golden_gpu.py tests/test_classifier_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_fence_free.txt
cat $O/pytest_fence_free.txt
