#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r03; mkdir -p $O
cd $REPO
timeout 500 python -m pytest tests/test_classifier_gpu.py tests/test_golden_gpu.py tests/test_multilabel_gpu.py -x -q -m gpu > $O/pytest_clf.log 2>&1
echo "pytest rc=$?" >> $O/pytest_clf.log; tail -3 $O/pytest_clf.log
