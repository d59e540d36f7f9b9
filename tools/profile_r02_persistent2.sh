#!/bin/bash
# SQ counters of the two persistent kernels after the LDS-conflict fix (gpurun_out/r02/persist2/).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r02/persist2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; T=/tmp/prof_persist2; rm -rf $T
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $T/e_$n -o p -- python $REPO/tools/epoch_probe3.py > $O/e_$n.txt 2>&1
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $T/s_$n -o p -- python $REPO/tools/single_query_probe.py > $O/s_$n.txt 2>&1
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("$T/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("head_epoch_kernel", "bert_small_kernel"):
            if k in r["Kernel_Name"]: agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
out = {"%s | %s" % k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in sorted(agg.items())}
json.dump(out, open("$O/persistent_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
