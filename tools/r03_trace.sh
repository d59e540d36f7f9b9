#!/bin/bash
# kernel trace of the timed predict step (per kernel: avg, calls, share, min .. max)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; T=/tmp/prof_tr; rm -rf $T
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $T -o s -- python $REPO/tools/step_trace_probe.py > $O/step_trace.txt 2>&1
python $REPO/tools/trace_agg.py $(find $T -name "*kernel_trace.csv" | head -1) 40 >> $O/step_trace.txt
grep "ms per step\| us x" $O/step_trace.txt | cut -c1-200
