#!/bin/bash
mkdir -p gpurun_out/r02
(time timeout 900 python -m pytest tests/test_gemm_split_gpu.py tests/test_encoder_gpu.py -x -q -m gpu) > gpurun_out/r02/tests_gemm.log 2>&1
tail -8 gpurun_out/r02/tests_gemm.log
(time timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sweep) > gpurun_out/r02/bench2.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r02/bench2.log"):
    if l.startswith("{"):
        j = json.loads(l); print({k: j[k] for k in ("value", "ms_per_step", "stages_ms")}, j["roofline_encoder"]["frac"], j.get("parity"))
PY
tail -3 gpurun_out/r02/bench2.log | cut -c1-300
