#!/bin/bash
# round 4: what paces the fp16x2 GEMM loop?  Matrix-pipe activity and the texture-data return path (global_load_lds staging) of
# the FFN2 and QKV shapes at the bench's 5141 rows, bf16x3 and fp16x2 kernels side by side.  Counter passes only (no tracing).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/f16x2_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; T=/tmp/f16pmc; rm -rf $T
for shp in 5141,768,3072,0,1,0 5141,2304,768,0,0,0; do
  for ar in bf16x3 f16x2; do
    i=0
    for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_TC_STALL_sum TA_TA_BUSY_sum" "TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum"; do
      i=$((i+1))
      timeout 200 rocprofv3 --pmc $set --output-format csv -d $T/${shp}_${ar}_$i -o p -- python $REPO/tools/gemm_pmc_probe.py $shp $ar > $O/${shp}_${ar}_$i.txt 2>&1
    done
  done
done
python - <<PY
import csv, glob, collections, json, os
out = {}
for d in sorted(glob.glob("$T/*")):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "gemm_pipe_nt" in kn:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    key = os.path.basename(d).rsplit("_", 1)[0]
    out.setdefault(key, {}).update({k: sum(v) / len(v) for k, v in sorted(agg.items())})
for k, c in out.items():
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if cyc:
        c["shader_cycles_per_launch"] = cyc
        c["mfma_busy_fraction_of_simd_cycles"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024)
        if "TD_TD_BUSY_sum" in c:
            c["td_busy_fraction_of_cu_cycles"] = c["TD_TD_BUSY_sum"] / (cyc * 256)
            c["td_stalled_on_cache_fraction_of_cu_cycles"] = c["TD_TC_STALL_sum"] / (cyc * 256)
            c["ta_busy_fraction_of_cu_cycles"] = c.get("TA_TA_BUSY_sum", 0) / (cyc * 256)
        if "TCC_HIT_sum" in c:
            c["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
json.dump(out, open("$O/f16x2_gemm_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:5000])
PY
