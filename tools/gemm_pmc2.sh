#!/bin/bash
# rocprofv3 memory-path counters of the planes GEMM (run on the GPU box via gpurun)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_gemm2
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA|TCP|TD|TCC)_[A-Z0-9_]+(_sum|_avr)?\b" | sort -u > $OUT/counters.txt
wc -l $OUT/counters.txt
i=0
for set in "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o g -- python $REPO/tools/gemm_pmc_probe.py $1 > $OUT/p$i.txt 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_planes" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, "n=%d mean=%.4g" % (len(v), sum(v) / len(v)))
PY
grep -il "error\|invalid\|not found" $OUT/p*.txt | head
