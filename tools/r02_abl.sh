#!/bin/bash
for pipe in 1 0; do for a in 6 4; do echo "== PIPE=$pipe ABL=$a (MFMA on real data, no fragment reads)"; AC_RING_PIPE=$pipe AC_RING_ABLATE=$a timeout 60 tools/ab/gemm_bench 0 10 8192,8192,8192,0,0,0 | grep -v "^M="; done; done
