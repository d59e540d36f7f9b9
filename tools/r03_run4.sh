#!/bin/bash
# round 3, GPU call 4: built-in per-shape choice vs explicit tables inside both encoders; full GPU suite; bench line
mkdir -p gpurun_out/r03
N768="768x768=124261;768x3072=124262"
{
echo "### bert-base encoder (ragged 256 x 32 batch)"
timeout 600 python tools/encode_ab.py "two-buffer=" "builtin=@builtin" "t2=$N768;2304x768=234232" "t4=$N768;2304x768=234232;3072x768=244232"
echo "### bert-large (BASELINE configs[4] encoder: 1024 texts x 32, ragged)"
timeout 600 python tools/encode_ab.py --large "two-buffer=" "builtin=@builtin" "b=1024x1024=124262;1024x4096=224242;3072x1024=244232;4096x1024=244232"
} > gpurun_out/r03/encode_ab4.txt 2>&1
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r03/pytest_run4.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03/pytest_run4.log
timeout 300 python bench.py > gpurun_out/r03/bench_run4.json 2> gpurun_out/r03/bench_run4.err
tail -4 gpurun_out/r03/pytest_run4.log; cat gpurun_out/r03/encode_ab4.txt
