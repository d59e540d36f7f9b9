#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
python tools/r06_gap_breakdown.py 2>&1 | grep -v amdgpu.ids | tee $O/gap_breakdown.txt
