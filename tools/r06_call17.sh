#!/bin/bash
# round 6, call 17: the prologue kernel after its LDS-atomic diet; the last layer's tail in one launch + 32-key CLS attention: tests, A/B
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
AC_LIBACAMD_PATH=$REPO/tools/ab/libacamd_pkstamps.so python tools/r06_prologue_stamps_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/prologue_stamps.txt
timeout 900 python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "tail_in_one_launch or unpad_one_call or padding_free" 2>&1 | tail -5 | tee $O/pytest_tail.txt
{
for rnd in 1 2; do
  AC_BERT_TAIL_FUSED=0 python tools/r06_encode_ab.py "tail: reduce, LayerNorm, normalize" base
  python tools/r06_encode_ab.py "tail: one launch" base
done
} 2>&1 | grep -v amdgpu.ids | tee $O/encode_ab_tail2.txt
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
python $REPO/tools/r06_step_seq.py $(find $T -name "*kernel_trace.csv" | head -1) pack_prologue_kernel > $O/step_launch_sequence_5.txt; head -3 $O/step_launch_sequence_5.txt; tail -24 $O/step_launch_sequence_5.txt
cd $REPO
timeout 2400 python -m pytest tests/test_encoder_gpu.py tests/test_e2e_reference_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest_encoder_files_5.txt
