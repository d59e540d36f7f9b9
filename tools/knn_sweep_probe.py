"""Scratch probe: the fp32 kNN sweep kernel alone (HIP events around it, as bench.py's `roofline`) at 10M x 768 and at
100k x 768; AC_KNN_NT=0/1 in the environment forces plain / non-temporal row loads."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
for rows in (10_000_000, 100_000):
    r = bench.sweep_roofline(dev, rows, full=False, parity=False)
    print(rows, "rows:", "%.0f GB/s  frac %.3f  kernel %.4f ms (min %.4f)  whole call %.4f ms" %
          (r["achieved"], r["frac"], r["avg_kernel_ms"], r["min_kernel_ms"], r["whole_call_ms"]))
    torch.cuda.empty_cache()
