#!/bin/bash
# Same-box A/B of the encoder between an older commit's tree and this one (how round 4 found the ring-drain regression,
# DESIGN 2.3g).  Build the old tree HERE first (it travels to the GPU box with the snapshot; tools/ab/ is git-ignored):
#   rm -rf tools/ab/old_tree && mkdir -p tools/ab/old_tree && git archive <commit> adaptive-classifier_amd include | tar -x -C tools/ab/old_tree \
#     && make -C tools/ab/old_tree/adaptive-classifier_amd/csrc -j8
# then on the GPU:  bash tools/r04_tree_ab.sh            (timings)   |   bash tools/r04_tree_ab.sh prof   (+ rocprofv3 kernel stats)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/tree_ab; mkdir -p $O
(python $REPO/tools/time_encode_tree.py $REPO/tools/ab/old_tree "old tree"; python $REPO/tools/time_encode_tree.py $REPO "this tree"; python $REPO/tools/time_encode_tree.py $REPO/tools/ab/old_tree "old tree again") 2>&1 | grep encode | tee $O/encode_old_vs_new.txt
[ "$1" = prof ] || exit 0
cd /tmp && export TMPDIR=/tmp
for t in old new; do
  root=$REPO; [ $t = old ] && root=$REPO/tools/ab/old_tree
  rm -rf /tmp/p_$t
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$t -o b -- python $REPO/tools/time_encode_tree.py $root $t > $O/prof_$t.txt 2>&1
  cp $(find /tmp/p_$t -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$t.csv
  echo "== $t"; head -8 $O/kernel_stats_$t.csv | cut -c1-200
done
