#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
timeout 1200 python -m pytest tests/test_knn_batch_gpu.py -x -q -m gpu -k "thresholds_from_the_sweep" 2>&1 | tail -40 | tee $O/pytest_two_phase.txt
