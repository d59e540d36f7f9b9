#!/bin/bash
# round 3, GPU call 6: rocprofv3 evidence -- kernel split of the batched kNN call, its HBM traffic, MFMA-busy counters of
# the encoder GEMMs under the built-in dispatch
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r03/prof6; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; T=/tmp/prof6; rm -rf $T
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $T/knn -o knn -- python $REPO/tools/knn_batch_probe.py 10000000,768,4096,32 > $O/knn_probe_under_rocprof.txt 2>&1
cp $(find $T/knn -name "*kernel_stats.csv" | head -1) $O/knn_batch_kernel_stats.csv
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $T/k_$n -o p -- python $REPO/tools/knn_batch_pmc_probe.py 10000000,768,4096,32 > $O/k_$n.txt 2>&1
done
for shp in 5141,2304,768,0,0,0 5141,768,768,0,1,0 5141,3072,768,2,0,1 5141,768,3072,0,1,0; do
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $T/g_$shp -o p -- python $REPO/tools/gemm_pmc_probe.py $shp > $O/g_$shp.txt 2>&1
done
python - <<PY
import csv, glob, collections, json, os
out = {}
for d in sorted(glob.glob("$T/*")):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "knn_batch_sweep" in kn or "gemm_pipe_nt" in kn or "gemm_planes_nt" in kn:
                agg[(kn.split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    if agg:
        out[os.path.basename(d)] = {"%s | %s" % k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in sorted(agg.items())}
json.dump(out, open("$O/pmc_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
head -14 $O/knn_batch_kernel_stats.csv | cut -c1-160
