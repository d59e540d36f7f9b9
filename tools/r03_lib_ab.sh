#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r03; mkdir -p $O
cd $REPO
for v in head new head new; do
  [ $v = head ] && export AC_LIBACAMD_PATH=$REPO/tools/ab/libacamd_head.so || unset AC_LIBACAMD_PATH
  echo "== $v"; timeout 150 python tools/knn_batch_probe.py 2>&1 | grep "N=" | cut -c1-100
done | tee $O/knn_ab_all.txt
