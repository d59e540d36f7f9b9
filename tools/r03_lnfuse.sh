#!/bin/bash
# fused-LayerNorm GEMM epilogue: parity tests first (short timeout: the exchange spins), then the encode A/B
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r03; mkdir -p $O
cd $REPO
timeout 300 python -m pytest tests/test_encoder_gpu.py -x -q -k "layernorm_fused" 2>&1 | tail -15 | tee $O/lnfuse_tests.txt
grep -q "passed" $O/lnfuse_tests.txt && ! grep -q "failed" $O/lnfuse_tests.txt || exit 1
ROUNDS=5 timeout 200 python tools/encode_ab.py "fused=@builtin" "separate=@builtin-nofuse" 2>&1 | tail -4 | tee $O/lnfuse_ab.txt
