#!/bin/bash
# the bench line the way the driver takes it: a fresh box, smoke, then `python bench.py`; plus the one alternate-path line whose
# filter changed
mkdir -p gpurun_out/r04/closing
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/closing/smoke.txt 2>&1; tail -1 gpurun_out/r04/closing/smoke.txt
python bench.py > gpurun_out/r04/closing/bench_line.json 2> gpurun_out/r04/closing/bench_err.txt; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04/closing/bench_line.json"))
print("value", d["value"], "ms", d["ms_per_step"], "stages", d["stages_ms"], "roofline", d["roofline"]["frac"])
f = d["config"]["value_f16x2_opt_in"]; print("f16x2", f["value"], f["encode_ms"])
PY
AC_KNN_PLANE=0 timeout 300 python -m pytest tests/test_knn_batch_gpu.py -q -m gpu -k "not plane and not load_rows and not second_search and not push_pressure and not incrementally" 2>&1 | tail -1
