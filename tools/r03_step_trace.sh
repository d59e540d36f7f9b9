#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; T=/tmp/prof11; rm -rf $T
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $T/a -o s -- python $REPO/tools/step_trace_probe.py > $O/step_trace_a.txt 2>&1
python $REPO/tools/trace_agg.py $(find $T/a -name "*kernel_trace.csv" | head -1) 45 >> $O/step_trace_a.txt
AC_STEP_BATCHED=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $T/b -o s -- python $REPO/tools/step_trace_probe.py > $O/step_trace_b.txt 2>&1
python $REPO/tools/trace_agg.py $(find $T/b -name "*kernel_trace.csv" | head -1) 45 >> $O/step_trace_b.txt
grep "ms per step" $O/step_trace_a.txt $O/step_trace_b.txt
