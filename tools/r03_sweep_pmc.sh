#!/bin/bash
# HBM traffic of the fp32 kNN sweep (knn_sweep_ring at <= 16 queries) at 10M x 768: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
# separate passes, no tracing (MI355X_MICROARCH.md, HBM section)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T=/tmp/prof_sw; rm -rf $T; mkdir -p $T
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $T/$c -o p -- python $REPO/tools/knn_probe.py 10000000,768,16,32 > $OUT/sweep_pmc_$c.txt 2>&1
done
python - <<PY
import csv, glob, collections, json, re
agg = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$T/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "knn_sweep" in r["Kernel_Name"]:
                agg[(re.sub(r"\(anonymous namespace\)::|\(.*$|^void ", "", r["Kernel_Name"])[:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
out = {"%s | %s" % k: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in sorted(agg.items())}
json.dump(out, open("$OUT/sweep_pmc_raw.json", "w"), indent=1)
for k, v in out.items(): print(k, v)
PY
