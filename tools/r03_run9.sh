#!/bin/bash
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03/pytest_run9.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03/pytest_run9.log
/usr/bin/time -v timeout 600 python bench.py > gpurun_out/r03/bench_run9.json 2> gpurun_out/r03/bench_run9.err
tail -4 gpurun_out/r03/pytest_run9.log; grep -E "Elapsed|Maximum resident" gpurun_out/r03/bench_run9.err; tail -c 1500 gpurun_out/r03/bench_run9.json
