#!/bin/bash
# round 4: the opt-in fp16x2 arithmetic -- its tests (with their printed error tables), the per-shape sweep + encoder A/B for
# bert-base and bert-large, the rotated-k A/B, old tree vs this tree on the same box
mkdir -p gpurun_out/f16x2
python -m pytest tests/test_gemm_f16x2_gpu.py -q -m gpu -s > gpurun_out/f16x2/pytest_f16x2.txt 2>&1; echo "f16x2 tests rc=$?"
tail -2 gpurun_out/f16x2/pytest_f16x2.txt
timeout 600 python tools/f16x2_probe.py > gpurun_out/f16x2/probe_base.txt 2>&1; echo "probe rc=$?"
cut -c1-400 gpurun_out/f16x2/probe_base.txt | tail -10
timeout 600 python tools/f16x2_probe.py --large --no-sweep > gpurun_out/f16x2/probe_large.txt 2>&1; tail -3 gpurun_out/f16x2/probe_large.txt
python tools/encode_krot_ab.py --large > gpurun_out/f16x2/krot_ab_large.txt 2>&1; tail -4 gpurun_out/f16x2/krot_ab_large.txt
if [ -d tools/ab/old_tree ]; then
  (python tools/time_encode_tree.py tools/ab/old_tree "tree before the fp16x2 work (e5dd58b)"; python tools/time_encode_tree.py . "this tree") 2>&1 | grep encode | tee gpurun_out/f16x2/old_vs_new_same_box.txt
fi
