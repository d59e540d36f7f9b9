#!/bin/bash
# round 3, GPU call 3: one-round tiles (256 x 192, 256 x 256) for the QKV / FFN1 shapes; tables inside the encoder
mkdir -p gpurun_out/r03
CFGS=1,222232,124261,124262,224242,234231,234232,234230,322432,244230,244231,244232,422430,224430,222442
{
echo "### planes GEMM sweep 3: cfg = tm tn wmw wnw ns pipe; variant 1 = two-buffer tile kernels"
GEMM_BENCH_STAMPS=1 timeout 300 tools/ab/gemm_bench $CFGS 20 3 5141,2304,768,0,0,0 5141,768,768,0,1,0 5141,3072,768,2,0,1 5141,768,3072,0,1,0
timeout 300 tools/ab/gemm_bench $CFGS 10 3 20564,3072,1024,0,0,0 20564,1024,1024,0,1,0 20564,4096,1024,2,0,1 20564,1024,4096,0,1,0 8192,8192,8192,0,0,0
} > gpurun_out/r03/gemm_sweep.txt 2>&1
N768="768x768=124261;768x3072=124262"
{
echo "### per-shape tables inside the bert-base encoder (ragged 256 x 32 batch, 5141 token rows)"
timeout 600 python tools/encode_ab.py "base=" "t1=$N768;2304x768=222232" "t2=$N768;2304x768=234232" "t3=$N768;2304x768=234232;3072x768=244231" \
   "t4=$N768;2304x768=234232;3072x768=244232" "t5=$N768;2304x768=234232;3072x768=244230" "t6=$N768;2304x768=234232;3072x768=224430" \
   "t7=$N768;2304x768=234231;3072x768=244231" "t8=$N768;2304x768=322432;3072x768=422430" "t9=$N768;2304x768=234230;3072x768=244232"
echo "### bert-large (BASELINE configs[4] encoder: 1024 texts x 32, ragged)"
L1="1024x1024=124262"
timeout 600 python tools/encode_ab.py --large "base=" "a=$L1;1024x4096=224242;3072x1024=224242;4096x1024=224242" \
   "b=$L1;1024x4096=224242;3072x1024=244232;4096x1024=244232" "c=$L1;1024x4096=222232;3072x1024=244231;4096x1024=244231" \
   "d=1024x1024=222232;1024x4096=224242;3072x1024=234232;4096x1024=244232"
} > gpurun_out/r03/encode_ab.txt 2>&1
tail -22 gpurun_out/r03/encode_ab.txt
