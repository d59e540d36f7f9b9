#!/bin/bash
# closing run of round 4: the whole GPU suite, smoke, the default bench line, bench.py --gpus 2 spawning its own ranks (gloo, one
# GPU), the kernel stats of the bench command, and the GPU suite's kNN / encoder files again under the alternate code paths
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/r04/final; mkdir -p $O
cd $REPO
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_full.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_full.txt
tail -3 $O/pytest_gpu_full.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
s=$(date +%s); timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$? wall $(( $(date +%s) - s )) s"
python -c "
import json;d=json.load(open('$O/bench_line.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline_fp16_plane']['frac'], d.get('latency_ms_b1',{}).get('value'), d.get('cfg4',{}).get('value'), d.get('add_examples',{}).get('value'))
print(d['stages_ms'], d.get('parity'))"
AC_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 5 --warmup 1 > $O/bench_gpus2_gloo.json 2> $O/bench_gpus2_gloo.err; echo "gpus2 rc=$? lines=$(grep -c '^{' $O/bench_gpus2_gloo.json)"
# (tests that assert WHICH form ran are left out of the run that switches that form off)
( AC_KNN_RING=0 timeout 300 python -m pytest tests/test_knn_gpu.py -q -m gpu -k "not lds_ring" 2>&1 | tail -1
  AC_KNN_PLANE=0 timeout 300 python -m pytest tests/test_knn_batch_gpu.py -q -m gpu -k "not plane and not load_rows and not second_search and not push_pressure and not incrementally" 2>&1 | tail -1
  AC_GEMM_ARITH=f32 timeout 400 python -m pytest tests/test_encoder_gpu.py -q -m gpu -k "not fused_into and not starved and not sticky" 2>&1 | tail -1
  AC_GEMM_ARITH=f16x2 timeout 400 python -m pytest tests/test_encoder_gpu.py tests/test_classifier_gpu.py tests/test_golden_gpu.py -q -m gpu 2>&1 | tail -1
  AC_GEMM_KROT=1 timeout 400 python -m pytest tests/test_encoder_gpu.py tests/test_golden_gpu.py -q -m gpu 2>&1 | tail -1
  AC_LN_FUSION=0 timeout 400 python -m pytest tests/test_encoder_gpu.py tests/test_classifier_gpu.py -x -q -m gpu -k "not starved and not sticky and not fused_into and not gave_up" 2>&1 | tail -1 ) > $O/alternate_paths.txt 2>&1
cat $O/alternate_paths.txt
cd /tmp && export TMPDIR=/tmp; T=/tmp/prof_fin4; rm -rf $T
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T -o b -- python $REPO/bench.py --no-extras --no-cpu-baseline > $O/bench_prof_line.json 2> /dev/null
cp $(find $T -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv 2>/dev/null; head -12 $O/bench_kernel_stats.csv | cut -c1-170
