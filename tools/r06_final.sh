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
  AC_KNN_RING=0 timeout 300 python -m pytest tests/test_knn_gpu.py -q -m gpu -k "not lds_ring" 2>&1 | tail -1 ) > $O/alternate_paths.txt 2>&1
cat $O/alternate_paths.txt
bash tools/r06_profiles.sh 2>&1 | tail -45
