#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r02/latprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; T=/tmp/prof_lat; rm -rf $T
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T -o lat -- python $REPO/bench.py --config latency --no-cpu-baseline > $O/out.txt 2>&1
cp $(find $T -name "*kernel_stats.csv" | head -1) $O/latency_kernel_stats.csv
cut -c1-130 $O/latency_kernel_stats.csv | head -24
