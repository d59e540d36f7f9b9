#!/bin/bash
# round 4: is knn_batch_sweep (4096 x 10M x 768) LDS-bound as DESIGN 2.2b says?  rocprofv3 --pmc passes (no tracing) with the
# LDS counters of MI355X_MICROARCH.md (SQ_LDS_IDX_ACTIVE = LDS-array cycles, SQ_LDS_BANK_CONFLICT = extra cycles), the MFMA-busy
# counter and the fabric reads again (the store plane is tile-major since round 4) -> gpurun_out/r04/knn_batch_lds_pmc_raw.json
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; T=/tmp/prof_lds; rm -rf $T; mkdir -p $T
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*" | sort -u > $O/avail_lds_counters.txt
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $T/p$i -o p -- python $REPO/tools/knn_batch_pmc_probe.py 10000000,768,4096,32 > $O/batch_lds_pass$i.txt 2>&1
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("$T/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "knn_batch_sweep" in r["Kernel_Name"] and "false" in r["Kernel_Name"]:          # the long (whole-store) form
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in sorted(agg.items())}
json.dump(out, open("$O/knn_batch_lds_pmc_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
