#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02/prof_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o train -- python $REPO/bench.py --config add_examples --examples 3000 > $OUT/stdout.txt 2>&1
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -30 $f | cut -c1-200; done
tail -c 600 $OUT/stdout.txt
