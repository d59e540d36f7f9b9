#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
AC_LIBACAMD_PATH=$REPO/tools/ab/libacamd_pkstamps.so python tools/r06_prologue_stamps_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/prologue_stamps.txt
timeout 900 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "unpad_one_call or padding_free" 2>&1 | tail -3
