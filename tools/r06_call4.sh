#!/bin/bash
# round 6, GPU call 4: the whole GPU suite, smoke, the default bench line, bench --gpus 2 self-spawned over gloo on the one GPU
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
timeout 3300 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/pytest_gpu_full_mid.txt
cat $O/pytest_gpu_full_mid.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py > $O/bench_line_mid.json 2> $O/bench_mid_stderr.txt
tail -c 1500 $O/bench_mid_stderr.txt
python - <<PY
import json
d = json.load(open("$O/bench_line_mid.json"))
print({k: d[k] for k in ("value", "ms_per_step", "stages_ms")})
print("roofline", d.get("roofline", {}).get("frac"), "encoder", d.get("roofline_encoder", {}).get("frac"))
print("add_examples", {k: d["add_examples"].get(k) for k in ("value", "examples", "steps_per_s")})
print("cfg4", d["cfg4"]["value"], "full_length", d["config"]["value_full_length"], "unfrozen", d["config"].get("value_gc_unfrozen"))
PY
AC_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --sweep-rows 2000000 --no-cpu-baseline > $O/bench_line_gpus2_selfspawn_gloo_one_gpu.json 2> $O/bench_gpus2_stderr.txt
tail -c 1200 $O/bench_gpus2_stderr.txt
python - <<PY
import json
d = json.load(open("$O/bench_line_gpus2_selfspawn_gloo_one_gpu.json"))
print("gpus2:", d["value"], d["config"].get("self_check"), d["config"]["one_gpu_same_workload"])
PY
