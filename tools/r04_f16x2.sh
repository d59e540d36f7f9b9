#!/bin/bash
# round 4: the opt-in fp16x2 arithmetic -- its tests, the bench line with its fp16x2 leg, the bert-large encoder A/B
mkdir -p gpurun_out/f16x2
python -m pytest tests/test_gemm_f16x2_gpu.py -q -m gpu -s > gpurun_out/f16x2/pytest_f16x2.txt 2>&1; echo "f16x2 tests rc=$?"
tail -5 gpurun_out/f16x2/pytest_f16x2.txt
python bench.py > gpurun_out/f16x2/bench_line.json 2> gpurun_out/f16x2/bench_err.txt; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/f16x2/bench_line.json"))
print("value", d["value"], "ms", d["ms_per_step"], "stages", d["stages_ms"])
print("f16x2", json.dumps(d["config"]["value_f16x2_opt_in"])[:900])
print("cfg4", d.get("cfg4", {}).get("value"), json.dumps(d.get("cfg4", {}).get("value_f16x2_opt_in")))
PY
timeout 600 python tools/f16x2_probe.py --large --no-sweep > gpurun_out/f16x2/probe_large.txt 2>&1; echo "probe rc=$?"
tail -4 gpurun_out/f16x2/probe_large.txt
