#!/bin/bash
# the final tree once more: whole GPU suite, smoke, the default bench line (a second sample next to the closing run's)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06/final2; mkdir -p $O
cd $REPO
s=$(date +%s); timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc=$? wall $(( $(date +%s) - s )) s" >> $O/pytest_gpu_full.txt
tail -3 $O/pytest_gpu_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
s=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$? wall $(( $(date +%s) - s )) s"
python - <<PY
import json
d = json.load(open("$O/bench_line.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["stages_ms"])
print("roofline", d["roofline"]["frac"], "encoder", d["roofline_encoder"]["frac"], "cfg4", d.get("cfg4", {}).get("value"), "add_examples", d.get("add_examples", {}).get("value"))
PY
AC_LIBACAMD_PATH=$REPO/tools/ab/libacamd_r05.so python tools/r06_encode_ab.py "r05 library" base 2>&1 | grep -v amdgpu.ids
python tools/r06_encode_ab.py "r06" base 2>&1 | grep -v amdgpu.ids
