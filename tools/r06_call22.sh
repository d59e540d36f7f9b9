#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
AC_HEAD_EPOCH_DEBUG=1 python tools/epoch_probe3.py 2>&1 | grep -v amdgpu.ids | tee $O/epoch_stamps.txt
python tools/epoch_probe3.py 2>&1 | grep -v amdgpu.ids | tail -2
