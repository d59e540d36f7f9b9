#!/bin/bash
# closing run of the round: the whole GPU suite, the default bench line, the kernel stats of the bench command
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r03; mkdir -p $O
cd $REPO
date +%s > $O/t0
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_full.log
tail -3 $O/pytest_gpu_full.log
s=$(date +%s); timeout 400 python bench.py > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$? wall $(( $(date +%s) - s )) s"
python -c "
import json;d=json.load(open('$O/bench_line.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d.get('latency_ms_b1',{}).get('value'), d.get('cfg4',{}).get('value'), d.get('add_examples',{}).get('value'), (d.get('predict_from_text') or {}).get('device_tokenizer_texts_per_s'))
print(d['config'].get('stage_ms'), d.get('parity'))"
cd /tmp && export TMPDIR=/tmp; T=/tmp/prof_fin; rm -rf $T
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T -o b -- python $REPO/bench.py --no-extras --no-cpu-baseline > $O/bench_prof_line.json 2> /dev/null
cp $(find $T -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv 2>/dev/null; head -8 $O/bench_kernel_stats.csv | cut -c1-160
