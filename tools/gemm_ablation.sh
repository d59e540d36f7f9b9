#!/bin/bash
# Evidence for DESIGN.md 2.3c: tile kernels (variant 0) vs the ring kernel (variant 2) on the encoder shapes, and the
# ablation ladder of the ring kernel at 8192^3 (which resource binds).  Output -> gpurun_out/r02/gemm_ablation.txt
mkdir -p gpurun_out/r02
{
echo "### tile kernels (variant 0) vs ring kernel (variant 2), uniform random operands, 20 reps"
timeout 120 tools/ab/gemm_bench 0,2 20
echo
echo "### ring kernel ablations at 8192^3 (software-pipelined PIPE=1 unless stated); MFMAs always see real data"
for cfg in "1 0:full kernel, compiler schedule" "1 16:full kernel, pinned 2-MFMA/1-read interleave" "0 0:full kernel, two staggered wave groups (PIPE=0)" \
           "1 6:MFMA only (no DMA, no fragment reads)" "1 4:MFMA + DMA (no fragment reads)" "1 2:MFMA + fragment reads (no DMA)" "1 9:DMA + fragment reads (no MFMA)" "1 3:barriers only"; do
  set -- $cfg; pipe=$1; rest="${cfg#* }"; abl="${rest%%:*}"; label="${rest#*:}"
  echo "-- $label   [AC_RING_PIPE=$pipe AC_RING_ABLATE=$abl]"
  AC_RING_PIPE=$pipe AC_RING_ABLATE=$abl timeout 60 tools/ab/gemm_bench 2 10 8192,8192,8192,0,0,0 | grep variant
done
} 2>&1 | tee gpurun_out/r02/gemm_ablation.txt
