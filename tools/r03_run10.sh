#!/bin/bash
mkdir -p gpurun_out/r03
s=$(date +%s)
timeout 900 python bench.py > gpurun_out/r03/bench_run10.json 2> gpurun_out/r03/bench_run10.err
echo "bench rc=$? wall=$(( $(date +%s) - s )) s" | tee -a gpurun_out/r03/bench_run10.err
tail -5 gpurun_out/r03/bench_run10.err; tail -c 3000 gpurun_out/r03/bench_run10.json
