#!/bin/bash
mkdir -p gpurun_out/r02
(time timeout 900 python -m pytest tests/test_knn_batch_gpu.py tests/test_knn_baseline_gpu.py tests/test_sharded_gpu.py -x -q -m gpu -s) > gpurun_out/r02/tests_batch2.log 2>&1; tail -12 gpurun_out/r02/tests_batch2.log
(time timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline) > gpurun_out/r02/bench3.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r02/bench3.log"):
    if l.startswith("{"):
        j = json.loads(l); print({k: j[k] for k in ("value", "ms_per_step", "stages_ms")}); print(j["roofline"]["batch4096"]); print(j["roofline"]["frac"], j["roofline"]["parity"])
PY
(time timeout 600 python bench.py --config cfg4 --steps 5 --warmup 2) > gpurun_out/r02/bench_cfg4b.log 2>&1; tail -c 900 gpurun_out/r02/bench_cfg4b.log
