#!/bin/bash
# round 5: fabric reads of knn_batch_sweep at BASELINE configs[2] (4096 x 10M x 768) with 4 (default) and 8 query tiles per XCD
# (AC_KNN_BATCH_B).  One FETCH_SIZE pass each (rocprofv3 --pmc only, no tracing) -> gpurun_out/r05/knn_batch_b_pmc_raw.json
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 4 8; do
  T=/tmp/prof_b$b; rm -rf $T; mkdir -p $T
  AC_KNN_BATCH_B=$b timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $T -o p -- python $REPO/tools/knn_batch_pmc_probe.py 10000000,768,4096,32 > $O/batch_b${b}_pass.txt 2>&1
done
python - <<PY
import csv, glob, collections, json
out = {}
for b in (4, 8):
    vals = []
    for f in glob.glob("/tmp/prof_b%d/**/*counter_collection.csv" % b, recursive=True):
        for r in csv.DictReader(open(f)):
            if "knn_batch_sweep" in r["Kernel_Name"] and "false" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                vals.append(float(r["Counter_Value"]))
    if vals:
        m = sum(vals) / len(vals)
        out["query_tiles_per_xcd_%d" % b] = {"launches": len(vals), "FETCH_SIZE_mean": m, "fabric_read_bytes_corrected": m * 1024 * 2,
                                             "store_passes": m * 1024 * 2 / (10000000 * 768 * 2)}
json.dump(out, open("$O/knn_batch_b_pmc_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
