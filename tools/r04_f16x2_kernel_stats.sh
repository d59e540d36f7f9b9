#!/bin/bash
# rocprofv3 kernel stats of the bench command under the opt-in fp16x2 arithmetic (AC_GEMM_ARITH=f16x2)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r04/last; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; T=/tmp/prof_f16; rm -rf $T
AC_GEMM_ARITH=f16x2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T -o b -- python $REPO/bench.py --no-extras --no-cpu-baseline --no-sweep > $O/bench_prof_line_f16x2.json 2> /dev/null
cp $(find $T -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_f16x2.csv; head -12 $O/bench_kernel_stats_f16x2.csv | cut -c1-190
