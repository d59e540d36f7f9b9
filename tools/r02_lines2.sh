#!/bin/bash
# Round-2 refresh after the persistent kernels: add_examples at 50k, its rocprof kernel stats (6000-example run), default line.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r02/lines2; mkdir -p $O
cd $REPO
(time timeout 1200 python bench.py --config add_examples --examples 50000) > $O/bench_add50k.log 2>&1; grep '^{' $O/bench_add50k.log > $O/bench_line_add_examples_50k.json; tail -4 $O/bench_add50k.log | cut -c1-400
(time timeout 900 python bench.py) > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/bench_line.json; tail -4 $O/bench_default.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
T=/tmp/prof_add; rm -rf $T
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $T -o add -- python $REPO/bench.py --config add_examples --examples 6000 > $O/add6000_under_rocprof.txt 2>&1
cp $(find $T -name "*kernel_stats.csv" | head -1) $O/add_examples_kernel_stats.csv
head -8 $O/add_examples_kernel_stats.csv | cut -c1-160
