#!/bin/bash
# round 6: knn_batch_sweep at BASELINE configs[2] on one GPU (4096 x 10M x 768): query tiles per XCD (AC_KNN_BATCH_B = 4 | 8) x
# non-temporal store-plane DMA (AC_KNN_BATCH_NT = 0 | 1): time (un-profiled) and fabric reads (one FETCH_SIZE pass each, counters only)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/knn_batch_nt_times.txt
for b in 4 8; do for nt in 0 1; do
  AC_KNN_BATCH_B=$b AC_KNN_BATCH_NT=$nt REPS=4 timeout 400 python $REPO/tools/batch_b_ab.py 2>&1 | grep -v amdgpu.ids | sed "s/^/NT=$nt /" | tee -a $O/knn_batch_nt_times.txt
  T=/tmp/prof_b${b}_$nt; rm -rf $T; mkdir -p $T
  AC_KNN_BATCH_B=$b AC_KNN_BATCH_NT=$nt timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $T -o p -- python $REPO/tools/knn_batch_pmc_probe.py 10000000,768,4096,32 > /dev/null 2>&1
done; done
python - <<PY
import csv, glob, json
out = {}
for b in (4, 8):
    for nt in (0, 1):
        vals = []
        for f in glob.glob("/tmp/prof_b%d_%d/**/*counter_collection.csv" % (b, nt), recursive=True):
            for r in csv.DictReader(open(f)):
                if "knn_batch_sweep" in r["Kernel_Name"] and "false" in r["Kernel_Name"].split("knn_batch_sweep")[1][:16] and r["Counter_Name"] == "FETCH_SIZE":
                    vals.append(float(r["Counter_Value"]))
        if vals:
            m = sum(vals) / len(vals)
            out["query_tiles_per_xcd_%d_nt_%d" % (b, nt)] = {"launches": len(vals), "FETCH_SIZE_mean": m, "fabric_read_GB_corrected": m * 1024 * 2 / 1e9,
                                                              "store_passes": m * 1024 * 2 / (10000000 * 768 * 2)}
json.dump(out, open("$O/knn_batch_nt_pmc_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
