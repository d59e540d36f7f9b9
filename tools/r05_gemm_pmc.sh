#!/bin/bash
# round 5: launch-level matrix-pipe activity, texture-data return path and L2 hit rate of the encoder's four GEMM shapes at the
# bench's 5141 rows (default bf16x3 dispatch; the N = 768 shapes as bias + residual, i.e. without the fused LayerNorm epilogue).
# Counter passes only (no tracing) -> gpurun_out/r05/gemm_planes_pmc.json
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; T=/tmp/gpmc5; rm -rf $T
for shp in 5141,2304,768,0,0,0 5141,768,768,0,1,0 5141,3072,768,2,0,1 5141,768,3072,0,1,0; do
  i=0
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_TC_STALL_sum TA_TA_BUSY_sum" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --output-format csv -d $T/${shp}_$i -o p -- python $REPO/tools/gemm_pmc_probe.py $shp > /dev/null 2>&1
  done
done
python - <<PY
import csv, glob, collections, json, os
names = {"5141,2304,768,0,0,0": "QKV 5141 x 2304 x 768", "5141,768,768,0,1,0": "attention output 5141 x 768 x 768 (+ residual)",
         "5141,3072,768,2,0,1": "FFN1 5141 x 3072 x 768 (GELU, planes out)", "5141,768,3072,0,1,0": "FFN2 5141 x 768 x 3072 (+ residual)"}
out = {}
for d in sorted(glob.glob("$T/*")):
    agg = collections.defaultdict(list); kn = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_pipe_nt" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"])); kn.add(r["Kernel_Name"].split("(")[0][-60:])
    key = names[os.path.basename(d).rsplit("_", 1)[0]]
    e = out.setdefault(key, {})
    e.update({k: sum(v) / len(v) for k, v in sorted(agg.items())}); e["kernel"] = sorted(kn)
for k, c in out.items():
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if cyc:
        c["shader_cycles_per_launch"] = cyc
        c["mfma_busy_fraction_of_simd_cycles"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024)
        if "TD_TD_BUSY_sum" in c:
            c["td_busy_fraction_of_cu_cycles"] = c["TD_TD_BUSY_sum"] / (cyc * 256)
            c["td_stalled_on_cache_fraction_of_cu_cycles"] = c["TD_TC_STALL_sum"] / (cyc * 256)
        if "TCC_HIT_sum" in c:
            c["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
json.dump(out, open("$O/gemm_planes_pmc.json", "w"), indent=1)
for k, c in out.items():
    print(k, "| mfma busy %.3f  td busy %.3f  td stalled %.3f  L2 hit %.3f  cycles %.0f" % (c.get("mfma_busy_fraction_of_simd_cycles", 0),
          c.get("td_busy_fraction_of_cu_cycles", 0), c.get("td_stalled_on_cache_fraction_of_cu_cycles", 0), c.get("l2_hit_rate", 0), c.get("shader_cycles_per_launch", 0)))
PY
