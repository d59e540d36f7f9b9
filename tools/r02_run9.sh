#!/bin/bash
mkdir -p gpurun_out/r02
(time timeout 900 python -m pytest tests/test_classifier_gpu.py tests/test_multilabel_gpu.py tests/test_head_gpu.py -x -q -m gpu) > gpurun_out/r02/tests_mem2.log 2>&1; tail -5 gpurun_out/r02/tests_mem2.log
(time timeout 900 python bench.py --config add_examples --examples 6000) > gpurun_out/r02/bench_add6000c.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r02/bench_add6000c.log"):
    if l.startswith("{"):
        j = json.loads(l)
        for m, v in j["modes"].items(): print(m, {k: v[k] for k in ("examples_per_s", "seconds", "train_steps", "steps_per_s", "host_seconds_by_phase", "new_class_seconds", "accuracy_5way")})
PY
