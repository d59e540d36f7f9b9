#!/bin/bash
# round 6, GPU call 3: residency / CU-mask test, the whole GPU suite, the default bench line
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
timeout 900 python -m pytest tests/test_classifier_gpu.py -q -m gpu -k "residency" -s 2>&1 | tail -30 > $O/pytest_residency.txt
cat $O/pytest_residency.txt
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $O/pytest_gpu_full_mid.txt
cat $O/pytest_gpu_full_mid.txt
timeout 900 python bench.py > $O/bench_line_mid.json 2> $O/bench_mid_stderr.txt
tail -c 3000 $O/bench_mid_stderr.txt
python - <<PY
import json
d = json.load(open("$O/bench_line_mid.json"))
print({k: d[k] for k in ("value", "ms_per_step", "stages_ms")})
print("roofline", d.get("roofline", {}).get("frac"), "encoder", d.get("roofline_encoder", {}).get("frac"))
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "kind", "cores")})
print("add_examples", {k: d["add_examples"].get(k) for k in ("value", "examples", "steps_per_s")}, d["add_examples"].get("cpu_baseline"))
print("cfg4", d["cfg4"]["value"], "full_length", d["config"]["value_full_length"], "unfrozen", d["config"].get("value_gc_unfrozen"))
PY
