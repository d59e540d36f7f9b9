#!/bin/bash
# round 5: where do the encoder's 5 ms go -- kernel time or the gaps between kernels?  Kernel trace of 18 forwards; the last 10 are analysed.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; T=/tmp/prof_trace5; rm -rf $T
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $T -o t -- python $REPO/tools/encode_trace_probe.py > $O/encode_trace_stdout.txt 2>&1
tail -1 $O/encode_trace_stdout.txt
python - <<PY
import csv, glob, re, json, collections
f = glob.glob("$T/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0][:60] for r in rows]
# one forward = from one embed_ln_kernel to the next
starts = [i for i, n in enumerate(names) if n.startswith("embed_ln_kernel")]
fw = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)][-9:]
out = {"forwards_analysed": len(fw)}
tot_k = tot_gap = tot_wall = 0.0
per = collections.defaultdict(lambda: [0, 0.0])
gaps = []
for a, b in fw:
    seg = rows[a:b]
    # drop the pack kernels / memsets that belong to the NEXT call's preamble: the forward ends at cls_normalize_kernel
    end = max(i for i in range(a, b) if names[i].startswith("cls_normalize_kernel"))
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
            "gap_us_median": sorted(gaps)[len(gaps) // 2] / 1e3, "gap_us_mean": sum(gaps) / len(gaps) / 1e3, "gap_us_max": max(gaps) / 1e3,
            "per_kernel_us": {k: {"calls_per_forward": v[0] / n, "avg_us": v[1] / v[0] / 1e3, "us_per_forward": v[1] / n / 1e3}
                              for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])}})
json.dump(out, open("$O/encode_trace.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3500])
PY
