import sys, ctypes
sys.path[:0] = ["/root/repo", "/root/repo/adaptive-classifier_amd"]
from adaptive_classifier import _native as nv
import torch
torch.zeros(1, device="cuda")
for k in range(3):
    for tm in (1, 2):
        n = ctypes.c_int(0)
        nv.check(nv.lib().ac_gemm_occupancy(k, tm, ctypes.byref(n)), "occ")
        print("kernel", k, "tm", tm, "blocks/CU", n.value)
