#!/bin/bash
# rocprofv3 evidence for the two persistent kernels: kernel stats of the single-query predict probe (how many launches a
# query costs now) and SQ counters of head_epoch_kernel.  Summaries only (gpurun_out/r02/persist/).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r02/persist; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; T=/tmp/prof_persist; rm -rf $T
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T/lat -o lat -- python $REPO/tools/latency_probe.py > $O/latency_under_rocprof.txt 2>&1
cp $(find $T/lat -name "*kernel_stats.csv" | head -1) $O/single_query_kernel_stats.csv
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $T/$n -o p -- python $REPO/tools/epoch_probe3.py > $O/pmc_$n.txt 2>&1
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("$T/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "head_epoch_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in sorted(agg.items())}
json.dump(out, open("$O/head_epoch_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
head -12 $O/single_query_kernel_stats.csv | cut -c1-150
du -sh $O
