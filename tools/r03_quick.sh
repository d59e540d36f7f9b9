#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r03; mkdir -p $O
cd $REPO
for v in 0 1 6; do echo "== AC_KNN_RING=$v"; AC_KNN_RING=$v timeout 100 python tools/knn_sweep_probe.py 2>&1 | grep "rows:"; done | tee $O/knn_ring_ab.txt
timeout 300 python -m pytest tests/test_knn_gpu.py tests/test_knn_baseline_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest_knn_ring.txt
