#!/bin/bash
# round 3, GPU call 1: ring-staged GEMM sweep (gemm_pipe.hip) + the tests touched so far + a baseline bench line
mkdir -p gpurun_out/r03
CFGS=1,2220,2230,2231,2232,2241,2261,2262,1220,1240,1241,1281,1430,1431,1461,2420,2431,2441,2442
{
echo "### ring-staged planes GEMM sweep; variant 1 = two-buffer tile kernels (round-2 dispatch); cfg = tm*1000+wmw*100+ns*10+pipe"
timeout 300 tools/ab/gemm_bench $CFGS 20 3 5141,2304,768,0,0,0 5141,768,768,0,1,0 5141,3072,768,2,0,1 5141,768,3072,0,1,0
timeout 300 tools/ab/gemm_bench $CFGS 10 3 20564,3072,1024,0,0,0 20564,1024,1024,0,1,0 20564,4096,1024,2,0,1 20564,1024,4096,0,1,0 8192,8192,8192,0,0,0
} > gpurun_out/r03/gemm_sweep1.txt 2>&1
timeout 900 python -m pytest tests/test_gemm_split_gpu.py tests/test_encoder_gpu.py tests/test_sharded_gpu.py tests/test_classifier_gpu.py -x -q -m gpu -s > gpurun_out/r03/pytest_run1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03/pytest_run1.log
timeout 300 python bench.py > gpurun_out/r03/bench_run1.json 2> gpurun_out/r03/bench_run1.err
tail -5 gpurun_out/r03/pytest_run1.log
tail -c 600 gpurun_out/r03/gemm_sweep1.txt
