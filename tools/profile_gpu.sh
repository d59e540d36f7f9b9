#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py + PMC passes for the kNN sweep.
# Summaries land in gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/.
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel time of the default bench command (fewer steps; no CPU baseline leg)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- \
    python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_stdout.txt 2>&1
# 2. HBM traffic of the sweep kernel: separate PMC passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o sweep -- \
    python $REPO/tools/knn_probe.py 10000000,768,16,32 > $OUT/pmc_fetch_stdout.txt 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o sweep -- \
    python $REPO/tools/knn_probe.py 10000000,768,16,32 > $OUT/pmc_write_stdout.txt 2>&1
find $OUT -name "*.csv" | head -50
for f in $(find $OUT/bench -name "*kernel_stats.csv"); do echo "== $f"; head -25 $f; done
python - <<PY
import csv, glob, collections
for tag in ("pmc_fetch", "pmc_write"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print(tag, k, c, "launches=%d mean=%.1f max=%.1f" % (len(v), sum(v) / len(v), max(v)))
PY
