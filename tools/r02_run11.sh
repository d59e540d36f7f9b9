#!/bin/bash
mkdir -p gpurun_out/r02
(time timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_gemm_split_gpu.py tests/test_classifier_gpu.py tests/test_tokenizer_gpu.py -x -q -m gpu) > gpurun_out/r02/tests_pack.log 2>&1; tail -8 gpurun_out/r02/tests_pack.log
(time timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sweep) > gpurun_out/r02/bench4.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r02/bench4.log"):
    if l.startswith("{"):
        j = json.loads(l); print({k: j[k] for k in ("value", "ms_per_step", "stages_ms")}); print(j["roofline_encoder"]); print(j["parity"]); print(j["config"]["value_f32_mfma"])
PY
tail -3 gpurun_out/r02/bench4.log | cut -c1-400
