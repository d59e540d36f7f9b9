#!/bin/bash
# closing run of round 6: the whole GPU suite, smoke, the default bench line (what the driver runs), bench --gpus 2 self-spawned over
# gloo on the one GPU, the round's profiles (tools/r06_profiles.sh), the in-network stamps, the alternate code paths.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06/final; mkdir -p $O
cd $REPO
s=$(date +%s); timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc=$? wall $(( $(date +%s) - s )) s" >> $O/pytest_gpu_full.txt
tail -4 $O/pytest_gpu_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
s=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$? wall $(( $(date +%s) - s )) s"
python - <<PY
import json
d = json.load(open("$O/bench_line.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["stages_ms"])
print("roofline", d["roofline"]["frac"], d["roofline"].get("traffic"), "fp16 plane", d["roofline_fp16_plane"]["frac"], "encoder", d["roofline_encoder"]["frac"])
print("latency", d.get("latency_ms_b1", {}).get("value"), "cfg4", d.get("cfg4", {}).get("value"), "add_examples", d.get("add_examples", {}).get("value"),
      d.get("add_examples", {}).get("steps_per_s"), "with encoder", d.get("add_examples_with_encoder", {}).get("value"))
print("parity", d.get("parity")); print("sustained", d["config"]["value_sustained"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["cores"], "| add_examples cpu", (d["add_examples"].get("cpu_baseline") or {}).get("value"))
print("full length", d["config"]["value_full_length"], "f32", d["config"]["value_f32_mfma"], "unfrozen", d["config"]["value_gc_unfrozen"])
PY
AC_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --sweep-rows 2000000 --no-cpu-baseline > $O/bench_line_gpus2_selfspawn_gloo_one_gpu.json 2> /dev/null
python -c "
import json; d=json.load(open('$O/bench_line_gpus2_selfspawn_gloo_one_gpu.json')); print('gpus2', d['value'], d['config']['self_check'])"
python tools/r06_layer_stamps_probe.py 2>&1 | grep -v amdgpu.ids > $O/layer_stamps.txt; cat $O/layer_stamps.txt
python tools/r06_attn_stamps_probe.py 2>&1 | grep -v amdgpu.ids > $O/attn_stamps.txt
( AC_QKV_ATTN_FUSION=0 timeout 600 python -m pytest tests/test_encoder_gpu.py tests/test_e2e_reference_gpu.py tests/test_classifier_gpu.py -q -m gpu -k "not attention_fus and not residency" 2>&1 | tail -1
  AC_QKV_ATTN_EXCHANGE=0 timeout 600 python -m pytest tests/test_encoder_gpu.py tests/test_e2e_reference_gpu.py -q -m gpu 2>&1 | tail -1
  AC_GEMM_ARITH=f32 timeout 600 python -m pytest tests/test_encoder_gpu.py tests/test_e2e_reference_gpu.py -q -m gpu -k "not fused_into and not attention_fus and not starved and not sticky and not per_call and not modernbert_gemm_arith" 2>&1 | tail -1
  AC_GEMM_ARITH=f16x2 timeout 600 python -m pytest tests/test_encoder_gpu.py tests/test_classifier_gpu.py tests/test_golden_gpu.py tests/test_e2e_reference_gpu.py -q -m gpu -k "not per_call and not residency and not modernbert_gemm_arith" 2>&1 | tail -1
  AC_LN_FUSION=0 timeout 600 python -m pytest tests/test_encoder_gpu.py tests/test_classifier_gpu.py tests/test_e2e_reference_gpu.py -q -m gpu -k "not starved and not sticky and not fused_into and not gave_up and not per_call and not residency" 2>&1 | tail -1
  AC_KNN_RING=0 timeout 300 python -m pytest tests/test_knn_gpu.py -q -m gpu -k "not lds_ring" 2>&1 | tail -1
  AC_KNN_TWO_PHASE=0 AC_KNN_THR_EXACT=1 timeout 600 python -m pytest tests/test_knn_batch_gpu.py tests/test_knn_gpu.py tests/test_classifier_gpu.py tests/test_golden_gpu.py -q -m gpu 2>&1 | tail -1
  AC_EXCHANGE_FENCES=1 timeout 600 python -m pytest tests/test_encoder_gpu.py tests/test_e2e_reference_gpu.py -q -m gpu 2>&1 | tail -1
  AC_BERT_UNPAD_ONE_CALL=0 timeout 600 python -m pytest tests/test_encoder_gpu.py tests/test_e2e_reference_gpu.py tests/test_classifier_gpu.py -q -m gpu 2>&1 | tail -1
  AC_PREDICT_POST=0 timeout 600 python -m pytest tests/test_classifier_gpu.py tests/test_multilabel_gpu.py tests/test_e2e_reference_gpu.py tests/test_golden_gpu.py -q -m gpu 2>&1 | tail -1
  AC_BERT_TAIL_FUSED=0 AC_GEMM_FEWTILES=0 timeout 600 python -m pytest tests/test_encoder_gpu.py tests/test_head_gpu.py tests/test_e2e_reference_gpu.py tests/test_golden_gpu.py -q -m gpu -k "not few_tile" 2>&1 | tail -1 ) > $O/alternate_paths.txt 2>&1
cat $O/alternate_paths.txt
bash tools/r06_profiles.sh 2>&1 | tail -45

# the round's same-box A/B: the round-5 library, this tree with the round's switches off one at a time, this tree
{
for rnd in 1 2; do
  AC_LIBACAMD_PATH=$REPO/tools/ab/libacamd_r05.so python tools/r06_encode_ab.py "r05 library" base
  AC_EXCHANGE_FENCES=1 AC_BERT_UNPAD_ONE_CALL=0 AC_BERT_TAIL_FUSED=0 AC_GEMM_FEWTILES=0 python tools/r06_encode_ab.py "r06 at its first closing run" base
  AC_EXCHANGE_FENCES=1 python tools/r06_encode_ab.py "r06, release/acquire exchanges" base
  AC_BERT_UNPAD_ONE_CALL=0 python tools/r06_encode_ab.py "r06, pack -> read-back -> encode" base
  python tools/r06_encode_ab.py "r06" base
done
for what in full large; do
  AC_LIBACAMD_PATH=$REPO/tools/ab/libacamd_r05.so python tools/r06_encode_ab.py "r05 library" $what
  python tools/r06_encode_ab.py "r06" $what
done
} 2>&1 | grep -v amdgpu.ids | tee $O/encode_ab_closing.txt
python tools/r06_gap_probe.py 2>&1 | grep -v amdgpu.ids | head -3 | tee $O/gap_probe.txt
python tools/r06_head_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/head_probe.txt
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
python $REPO/tools/r06_step_seq.py $(find $T -name "*kernel_trace.csv" | head -1) pack_prologue_kernel > $O/step_launch_sequence.txt; head -3 $O/step_launch_sequence.txt; tail -24 $O/step_launch_sequence.txt
