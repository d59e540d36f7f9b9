#!/bin/bash
# round 6 profiles: (1) kernel trace of the encoder alone on the bench batch (per-kernel time per forward, gaps); (2) counter passes of the
# SAME encoder forwards aggregated per kernel (matrix-pipe busy cycles of every GEMM as it runs IN the network, incl. the QKV GEMM with the
# attention epilogue; texture-return path; L2 hits); (3) rocprofv3 --kernel-trace --stats of the bench command; (4) FETCH / WRITE passes
# of the roofline kernel.  Counter passes carry no tracing flags.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T=/tmp/prof_trace6; rm -rf $T
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $T -o t -- python $REPO/tools/encode_trace_probe.py > $O/encode_trace_stdout.txt 2>&1
tail -1 $O/encode_trace_stdout.txt
python - <<PY
import csv, glob, re, json, collections
f = glob.glob("$T/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0][:64] for r in rows]
starts = [i for i, n in enumerate(names) if n.startswith("pack_prologue_kernel")]
fw = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)][-9:]
out = {"forwards_analysed": len(fw)}
tot_k = tot_gap = tot_wall = 0.0
per = collections.defaultdict(lambda: [0, 0.0])
gaps = []
for a, b in fw:
    end = max(i for i in range(a, b) if names[i].startswith(("splitk_ln_normalize_kernel", "cls_normalize_kernel")))
    seg = rows[a:end + 1]
    k = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    wall = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    tot_k += k; tot_wall += wall; tot_gap += wall - k
    for i in range(a, end + 1):
        per[names[i]][0] += 1; per[names[i]][1] += int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])
        if i > a: gaps.append(int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]))
n = len(fw)
out.update({"kernels_per_forward": sum(v[0] for v in per.values()) / n, "kernel_time_us_per_forward": tot_k / n / 1e3,
            "wall_us_per_forward": tot_wall / n / 1e3, "gap_us_per_forward": tot_gap / n / 1e3,
            "gap_us_median": sorted(gaps)[len(gaps) // 2] / 1e3, "gap_us_max": max(gaps) / 1e3,
            "per_kernel_us": {k: {"calls_per_forward": v[0] / n, "avg_us": v[1] / v[0] / 1e3, "us_per_forward": v[1] / n / 1e3}
                              for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])}})
json.dump(out, open("$O/encode_trace.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
PY
T=/tmp/prof_pmc6; rm -rf $T
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $T/$i -o p -- python $REPO/tools/encode_trace_probe.py > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$T/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0][:64]
        if n.startswith(("gemm_", "attention", "embed", "ln_kernel", "pack_prologue", "splitk_ln")):
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for n, cs in agg.items():
    c = {k: sum(v) / len(v) for k, v in cs.items()}
    c["launches_counted"] = len(cs.get("GRBM_GUI_ACTIVE", []))
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if cyc:
        c["shader_cycles_per_launch"] = cyc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c: c["mfma_busy_fraction_of_simd_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
        if "TD_TD_BUSY_sum" in c:
            c["td_busy_fraction_of_cu_cycles"] = c["TD_TD_BUSY_sum"] / (cyc * 256)
            c["td_stalled_on_cache_fraction_of_cu_cycles"] = c["TD_TC_STALL_sum"] / (cyc * 256)
        if "TCC_HIT_sum" in c: c["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    out[n] = c
json.dump(out, open("$O/encode_pmc_per_kernel.json", "w"), indent=1)
for n, c in sorted(out.items(), key=lambda kv: -kv[1].get("shader_cycles_per_launch", 0)):
    print("%-64s mfma busy %.3f td busy %.3f stalled %.3f L2 hit %.3f cycles %.0f" % (n, c.get("mfma_busy_fraction_of_simd_cycles", 0),
          c.get("td_busy_fraction_of_cu_cycles", 0), c.get("td_stalled_on_cache_fraction_of_cu_cycles", 0), c.get("l2_hit_rate", 0), c.get("shader_cycles_per_launch", 0)))
PY
T=/tmp/prof_fin6; rm -rf $T
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $T -o b -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_line_under_rocprof.json 2> /dev/null
cp $(find $T -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv 2>/dev/null; head -8 $O/bench_kernel_stats.csv | cut -c1-170
T=/tmp/prof_sw6; rm -rf $T; mkdir -p $T
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $T/$c -o p -- python $REPO/tools/knn_probe.py 10000000,768,16,32 > $O/sweep_pmc_$c.txt 2>&1
done
python - <<PY
import csv, glob, collections, json, re
agg = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$T/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "knn_sweep" in r["Kernel_Name"]:
                agg[(re.sub(r"\(anonymous namespace\)::|\(.*$|^void ", "", r["Kernel_Name"])[:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
out = {"%s | %s" % k: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in sorted(agg.items())}
json.dump(out, open("$O/sweep_pmc_raw.json", "w"), indent=1)
for k, v in out.items(): print(k, v)
PY
