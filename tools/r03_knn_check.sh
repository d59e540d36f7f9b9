#!/bin/bash
# batched kNN: tests + timings + the bench step's kernel trace
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r03; mkdir -p $O
timeout 400 python -m pytest tests/test_knn_batch_gpu.py tests/test_knn_baseline_gpu.py tests/test_sharded_gpu.py tests/test_knn_gpu.py -x -q -m gpu > $O/pytest_knn.log 2>&1
echo "pytest rc=$?" >> $O/pytest_knn.log
timeout 150 python tools/knn_batch_probe.py > $O/knn_batch_probe_staged.txt 2>&1
cd /tmp && export TMPDIR=/tmp; T=/tmp/prof_kc; rm -rf $T
timeout 150 rocprofv3 --kernel-trace --output-format csv -d $T/b -o s -- python $REPO/tools/step_trace_probe.py > $O/step_trace_c.txt 2>&1
python $REPO/tools/trace_agg.py $(find $T/b -name "*kernel_trace.csv" | head -1) 45 >> $O/step_trace_c.txt
cd $REPO; tail -3 $O/pytest_knn.log; cat $O/knn_batch_probe_staged.txt; grep "ms per step" $O/step_trace_c.txt; grep "knn_\|fill" $O/step_trace_c.txt | cut -c1-150
