#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp; T=/tmp/prof_ks; rm -rf $T
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T -o ks -- python $REPO/tools/knn_small_batch_probe.py > /dev/null 2>&1
cut -c1-150 $(find $T -name "*kernel_stats.csv" | head -1) | head -14
