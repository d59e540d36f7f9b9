#!/bin/bash
# round 6, call 14: ac_predict_post (one launch, host-mapped result, completion flag), the C unpack, the faster prologue kernel:
# tests, then the step's host phases and launch sequence
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
timeout 900 python -m pytest tests/test_classifier_gpu.py tests/test_encoder_gpu.py -x -q -m gpu -k "predict_post or post_kernel or unpad_one_call" 2>&1 | tail -25 | tee $O/pytest_post.txt
for i in 1; do timeout 300 python -m pytest tests/test_classifier_gpu.py -x -q -m gpu -k "predict_post" 2>&1 | grep -a "passed\|failed\|^E  " | head -5; done | tee $O/pytest_post_repeat.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/step_ms_4.txt
import os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "adaptive-classifier_amd")]
import torch, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
clf, hf = bench.make_classifier(dev, 0, 1)
ids, types, mask = bench.synthetic_tokens(dev, 0)
for post in ("1", "0", "1", "0", "1", "0"):
    os.environ["AC_HEAD_SIDE_STREAM"] = post
    for _ in range(5): bench.predict_step(clf, ids, types, mask)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60): bench.predict_step(clf, ids, types, mask)
    torch.cuda.synchronize()
    print("AC_HEAD_SIDE_STREAM=%s ms per step %.4f" % (post, (time.perf_counter() - t0) / 60 * 1e3), flush=True)
PY
cd /tmp && export TMPDIR=/tmp
T=/tmp/prof_step6; rm -rf $T
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
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $T -o t -- python /tmp/step_only.py > /dev/null 2>&1
python $REPO/tools/r06_step_seq.py $(find $T -name "*kernel_trace.csv" | head -1) pack_prologue_kernel > $O/step_launch_sequence_4.txt; head -4 $O/step_launch_sequence_4.txt; tail -24 $O/step_launch_sequence_4.txt
cd $REPO
timeout 2400 python -m pytest tests/test_classifier_gpu.py tests/test_multilabel_gpu.py tests/test_e2e_reference_gpu.py tests/test_golden_gpu.py tests/test_encoder_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_classifier_files_4.txt
