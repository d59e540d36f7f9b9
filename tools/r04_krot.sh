#!/bin/bash
# round 4: the rotated k order (branch-free form) -- GEMM / encoder suites, the A/B in the encoder, the bench line
mkdir -p gpurun_out/krot
python -m pytest tests/test_gemm_split_gpu.py tests/test_gemm_f16x2_gpu.py tests/test_encoder_gpu.py -q -m gpu > gpurun_out/krot/pytest2.txt 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/krot/pytest2.txt
python tools/encode_krot_ab.py > gpurun_out/krot/krot_ab_base.txt 2>&1; tail -4 gpurun_out/krot/krot_ab_base.txt
python bench.py > gpurun_out/krot/bench_line.json 2> gpurun_out/krot/bench_err.txt; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/krot/bench_line.json"))
print("value", d["value"], "ms", d["ms_per_step"], "stages", d["stages_ms"])
f = d["config"]["value_f16x2_opt_in"]; print("f16x2", {k: f[k] for k in f if "note" not in k})
print("f32", d["config"]["value_f32_mfma"], "full", d["config"]["value_full_length"])
print("cfg4", d.get("cfg4", {}).get("value"), d.get("cfg4", {}).get("value_f16x2_opt_in"))
PY
