#!/bin/bash
# round 6: knn_batch_sweep at BASELINE configs[2] on one GPU (4096 x 10M x 768) with the TILE-MAJOR query plane: time and fabric reads
# (one FETCH_SIZE pass each, counters only) for 4 and 8 query tiles per XCD
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd $REPO
timeout 1200 python -m pytest tests/test_knn_batch_gpu.py tests/test_knn_baseline_gpu.py -x -q -m gpu 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
: > $O/knn_batch_qplane_times.txt
for b in 4 8; do
  AC_KNN_BATCH_B=$b REPS=4 timeout 400 python $REPO/tools/batch_b_ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/knn_batch_qplane_times.txt
  T=/tmp/prof_qb${b}; rm -rf $T; mkdir -p $T
  AC_KNN_BATCH_B=$b timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $T -o p -- python $REPO/tools/knn_batch_pmc_probe.py 10000000,768,4096,32 > /dev/null 2>&1
done
python - <<PY
import csv, glob, json
out = {}
for b in (4, 8):
    vals = []
    for f in glob.glob("/tmp/prof_qb%d/**/*counter_collection.csv" % b, recursive=True):
        for r in csv.DictReader(open(f)):
            if "knn_batch_sweep" in r["Kernel_Name"] and "false" in r["Kernel_Name"].split("knn_batch_sweep")[1][:16] and r["Counter_Name"] == "FETCH_SIZE":
                vals.append(float(r["Counter_Value"]))
    if vals:
        m = sum(vals) / len(vals)
        out["query_tiles_per_xcd_%d" % b] = {"launches": len(vals), "FETCH_SIZE_mean": m, "fabric_read_GB_corrected": m * 1024 * 2 / 1e9,
                                             "store_passes": m * 1024 * 2 / (10000000 * 768 * 2)}
json.dump(out, open("$O/knn_batch_qplane_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
