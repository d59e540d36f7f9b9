#!/bin/bash
# HBM traffic of the two bandwidth-bound kNN sweeps at 10M x 768, 16 resident queries: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
# separate passes, no tracing (MI355X_MICROARCH.md, HBM section) -> gpurun_out/r04/knn_plane_sweep_pmc.json, knn_sweep_pmc.json
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T=/tmp/prof_sw4; rm -rf $T; mkdir -p $T
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $T/$c -o p -- python $REPO/tools/plane_pmc_probe.py 16 > $OUT/sweep_pmc_$c.txt 2>&1
done
python - <<PY
import csv, glob, collections, json, re
agg = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$T/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"knn_plane_sweep<\d+>|knn_sweep_ring<[\d, ]+>", r["Kernel_Name"])
            if m:
                agg[(m.group(0), r["Counter_Name"])].append(float(r["Counter_Value"]))
raw = {"%s | %s" % k: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in sorted(agg.items())}
json.dump(raw, open("$OUT/sweep_pmc_raw.json", "w"), indent=1)
for k, v in raw.items(): print(k, v)
N, D = 10_000_000, 768
note = ("MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read (16 B per lane, "
        "global_load and LDS-DMA alike) -> x2; WRITE_SIZE used as reported")
for pat, name, alg, fn in (("knn_plane_sweep", "knn_plane_sweep<32>  (16 resident queries: fp16 plane by non-temporal whole-line loads)", N * D * 2, "knn_plane_sweep_pmc.json"),
                           ("knn_sweep_ring", "knn_sweep_ring<4, 24>  (16 resident queries: fp32 rows by non-temporal LDS-DMA)", N * D * 4, "knn_sweep_pmc.json")):
    f = [v for (kn, c), v in agg.items() if pat in kn and c == "FETCH_SIZE"]
    w = [v for (kn, c), v in agg.items() if pat in kn and c == "WRITE_SIZE"]
    if not f or not w:
        continue
    f, w = f[0], w[0]
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    rd, wr = fk * 1024 * 2, wk * 1024
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes, no tracing) -- python tools/plane_pmc_probe.py 16   (tools/r04_sweep_pmc.sh)",
               "kernel": name, "rows": N, "dim": D, "algorithmic_bytes": alg, "FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk,
               "launches": len(f), "gfx950_correction": note, "hbm_read_bytes_per_launch_corrected": rd, "hbm_write_bytes_per_launch": wr,
               "traffic_over_algorithmic": (rd + wr) / alg}, open("$OUT/" + fn, "w"), indent=1)
    print(fn, "traffic / algorithmic =", (rd + wr) / alg)
PY
