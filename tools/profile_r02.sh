#!/bin/bash
# Round-2 evidence (run on the GPU box via gpurun): rocprofv3 kernel stats of the default bench command, HBM-traffic PMC
# passes of the kNN sweep and of the batched sweep, matrix-pipe PMC of the planes GEMM and of the batched sweep.
# Only small summaries are kept (gpurun_out/r02/prof/); copy what should be judged into profiles/r02/.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T=/tmp/prof_r02; rm -rf $T; mkdir -p $T
# 1. per-kernel time of the default bench command (no CPU-baseline leg)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $T/bench -o bench -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_stdout.txt 2>&1
cp $(find $T/bench -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
grep '^{' $OUT/bench_stdout.txt > $OUT/bench_line_under_rocprof.json
# 2. PMC passes (one counter set per run, no tracing)
pmc() { # name, counters, command...
  name=$1; shift; set_="$1"; shift
  timeout 300 rocprofv3 --pmc $set_ --output-format csv -d $T/$name -o p -- "$@" > $OUT/$name.txt 2>&1
}
pmc sweep_fetch "FETCH_SIZE" python $REPO/tools/knn_probe.py 10000000,768,16,32
pmc sweep_write "WRITE_SIZE" python $REPO/tools/knn_probe.py 10000000,768,16,32
pmc batch_fetch "FETCH_SIZE" python $REPO/tools/knn_batch_pmc_probe.py 10000000,768,4096,32
pmc batch_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" python $REPO/tools/knn_batch_pmc_probe.py 10000000,768,4096,32
pmc gemm_mfma_8192 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" python $REPO/tools/gemm_pmc_probe.py 8192,2304,768
pmc gemm_mfma_5141 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" python $REPO/tools/gemm_pmc_probe.py 5141,2304,768
python - <<PY
import csv, glob, collections, json, re
out = {}
for name in ("sweep_fetch", "sweep_write", "batch_fetch", "batch_mfma", "gemm_mfma_8192", "gemm_mfma_5141"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$T/%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[(re.sub(r"\(anonymous namespace\)::|\(.*$|^void ", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    out[name] = {"%s | %s" % k: {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)} for k, v in sorted(agg.items())}
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
for name, d in out.items():
    for k, v in d.items():
        if any(s in k for s in ("knn_sweep", "knn_batch_sweep", "gemm_planes")): print(name, k, v)
PY
head -12 $OUT/bench_kernel_stats.csv | cut -c1-180
du -sh $OUT
