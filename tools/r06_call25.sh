#!/bin/bash
# round 6, call 25: thresholds of the batch kNN from the main sweep's first tile round (two grid barriers) against the sample stage
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
timeout 1200 python -m pytest tests/test_knn_batch_gpu.py tests/test_knn_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest_two_phase.txt
cat > /tmp/step_only.py <<'PY'
import os, sys, time
R = os.environ["GRAFT_REPO_ROOT"] if "GRAFT_REPO_ROOT" in os.environ else "/root/repo"
sys.path[:0] = [R, os.path.join(R, "adaptive-classifier_amd")]
import torch, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
clf, hf = bench.make_classifier(dev, 0, 1)
ids, types, mask = bench.synthetic_tokens(dev, 0)
for _ in range(30): bench.predict_step(clf, ids, types, mask)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): bench.predict_step(clf, ids, types, mask)
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / 100 * 1e3, bench.time_stages(clf, ids, types, mask)["knn_ms"])
PY
for r in 1 2; do
AC_KNN_TWO_PHASE=0 python /tmp/step_only.py 2>&1 | grep "ms per step" | sed 's/^/sample stage: /'
python /tmp/step_only.py 2>&1 | grep "ms per step" | sed 's/^/two phase:    /'
done
cd /tmp && export TMPDIR=/tmp
T=/tmp/prof_step6; rm -rf $T
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $T -o t -- python /tmp/step_only.py > /dev/null 2>&1
python $REPO/tools/r06_step_seq.py $(find $T -name "*kernel_trace.csv" | head -1) pack_prologue_kernel > $O/step_launch_sequence_7.txt; tail -12 $O/step_launch_sequence_7.txt
cd $REPO
timeout 1800 python -m pytest tests/test_knn_baseline_gpu.py tests/test_sharded_gpu.py tests/test_poisoned_workspace_gpu.py tests/test_classifier_gpu.py tests/test_golden_gpu.py tests/test_e2e_reference_gpu.py -x -q -m gpu 2>&1 | tail -3
