"""Scratch probe for rocprofv3 --pmc: a few launches of the planes GEMM (both operands pre-split) on one shape."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import _native as nv
dev = torch.device("cuda:0"); lib = nv.lib()
M, N, K = (int(x) for x in sys.argv[1].split(","))
lib.ac_gemm_set_arith(1)
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** .5; b = torch.randn(N, device=dev)
C = torch.empty(M, N, device=dev)
Wp = torch.empty(3 * N * K, dtype=torch.int16, device=dev); Ap = torch.empty(3 * M * K, dtype=torch.int16, device=dev)
st = nv.stream_ptr(dev)
lib.ac_split_bf16x3(nv.ptr(W), K, N, K, nv.ptr(Wp), st); lib.ac_split_bf16x3(nv.ptr(A), K, M, K, nv.ptr(Ap), st)
for _ in range(4):
    nv.check(lib.ac_linear_bf16x3(nv.ptr(A), K, nv.ptr(Ap), nv.ptr(W), K, nv.ptr(Wp), nv.ptr(b), None, N, nv.ptr(C), N, None,
                                  M, N, K, 0, st), "lin")
torch.cuda.synchronize()
