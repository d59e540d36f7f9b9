#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r03; mkdir -p $O
cd $REPO
timeout 200 python tools/step_host_probe.py 2>&1 | tail -6 | tee $O/step_host.txt
