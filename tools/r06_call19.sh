#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06/final; mkdir -p $O
cd $REPO
s=$(date +%s); timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc=$? wall $(( $(date +%s) - s )) s" >> $O/pytest_gpu_full.txt
tail -4 $O/pytest_gpu_full.txt
