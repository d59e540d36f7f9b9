#!/bin/bash
SH="5141,2304,768,0,0,0 5141,768,768,0,1,0 5141,3072,768,2,0,1 5141,768,3072,0,1,0"
echo "== default"; timeout 60 tools/ab/gemm_bench 0 20 $SH | grep -v "^M="
echo "== TILE256=1"; AC_GEMM_TILE256=1 timeout 60 tools/ab/gemm_bench 0 20 $SH | grep -v "^M="
echo "== TILE256=0 TM=1"; AC_GEMM_TILE256=0 AC_GEMM_TM=1 timeout 60 tools/ab/gemm_bench 0 20 $SH | grep -v "^M="
echo "== TILE256=0 TM=2"; AC_GEMM_TILE256=0 AC_GEMM_TM=2 timeout 60 tools/ab/gemm_bench 0 20 $SH | grep -v "^M="
echo "== ring stream-K (variant 2)"; timeout 60 tools/ab/gemm_bench 2 20 $SH | grep -v "^M="
