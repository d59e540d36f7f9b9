#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r03; mkdir -p $O
cd $REPO
for v in head new head new; do
  [ $v = head ] && export AC_LIBACAMD_PATH=$REPO/tools/ab/libacamd_head.so || unset AC_LIBACAMD_PATH
  echo "== $v"; ROUNDS=3 timeout 200 python tools/encode_ab.py "builtin=@builtin" 2>&1 | tail -1
done | tee $O/splitk_ab.txt
unset AC_LIBACAMD_PATH
timeout 500 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu > $O/pytest_enc.log 2>&1
echo "pytest rc=$?" >> $O/pytest_enc.log; tail -3 $O/pytest_enc.log
