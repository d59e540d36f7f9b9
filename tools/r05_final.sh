#!/bin/bash
# closing run of round 5: the whole GPU suite, smoke, the default bench line (what the driver runs), the kernel stats of that
# command (rocprofv3 --kernel-trace --stats), the FETCH / WRITE PMC passes of the roofline kernel, configs[3] at full size with and
# without the encoder, and the alternate code paths.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r05/final; mkdir -p $O
cd $REPO
s=$(date +%s); timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc=$? wall $(( $(date +%s) - s )) s" >> $O/pytest_gpu_full.txt
tail -4 $O/pytest_gpu_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
s=$(date +%s); timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$? wall $(( $(date +%s) - s )) s"
python -c "
import json;d=json.load(open('$O/bench_line.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline_fp16_plane']['frac'], d.get('latency_ms_b1',{}).get('value'), d.get('cfg4',{}).get('value'), d.get('add_examples',{}).get('value'), d.get('add_examples_with_encoder',{}).get('value'))
print(d['stages_ms'], d.get('parity')); print(d['config']['value_sustained']); print(d['cpu_baseline']['value'], d['cpu_baseline']['kind'], d['cpu_baseline']['cores'])"
cd /tmp && export TMPDIR=/tmp; T=/tmp/prof_fin5; rm -rf $T
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $T -o b -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_prof_line.json 2> /dev/null
cp $(find $T -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv 2>/dev/null; head -8 $O/bench_kernel_stats.csv | cut -c1-170
T=/tmp/prof_sw5; rm -rf $T; mkdir -p $T
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
cd $REPO
timeout 600 python bench.py --config add_examples > $O/bench_line_add_examples_50k.json 2> /dev/null; echo "add_examples 50k rc=$?"
timeout 600 python bench.py --config add_examples --with-encoder > $O/bench_line_add_examples_50k_with_encoder.json 2> /dev/null; echo "add_examples 50k with encoder rc=$?"
python -c "
import json
for f in ('bench_line_add_examples_50k.json','bench_line_add_examples_50k_with_encoder.json'):
    d=json.load(open('$O/'+f)); print(f, d['value'], d.get('host_seconds_by_phase') or d['modes']['as_wired']['host_seconds_by_phase'])"
( AC_KNN_RING=0 timeout 300 python -m pytest tests/test_knn_gpu.py -q -m gpu -k "not lds_ring" 2>&1 | tail -1
  AC_KNN_PLANE=0 timeout 300 python -m pytest tests/test_knn_batch_gpu.py -q -m gpu -k "not plane and not load_rows and not second_search and not push_pressure and not incrementally" 2>&1 | tail -1
  AC_GEMM_ARITH=f32 timeout 400 python -m pytest tests/test_encoder_gpu.py tests/test_e2e_reference_gpu.py -q -m gpu -k "not fused_into and not starved and not sticky and not per_call" 2>&1 | tail -1
  AC_GEMM_ARITH=f16x2 timeout 400 python -m pytest tests/test_encoder_gpu.py tests/test_classifier_gpu.py tests/test_golden_gpu.py tests/test_e2e_reference_gpu.py -q -m gpu -k "not per_call" 2>&1 | tail -1
  AC_LN_FUSION=0 timeout 400 python -m pytest tests/test_encoder_gpu.py tests/test_classifier_gpu.py tests/test_e2e_reference_gpu.py -q -m gpu -k "not starved and not sticky and not fused_into and not gave_up and not per_call" 2>&1 | tail -1 ) > $O/alternate_paths.txt 2>&1
cat $O/alternate_paths.txt
