"""Scratch probe for rocprofv3 --pmc: a few launches of the planes GEMM (both operands pre-split) on one shape
(M,N,K[,act,res,cplanes]) through the default dispatch."""
import os as _os; _os.environ.setdefault("AC_TEST_HOOKS", "1")  # (the process-wide switches used below are test hooks)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import _native as nv
dev = torch.device("cuda:0"); lib = nv.lib()
v = [int(x) for x in sys.argv[1].split(",")] + [0, 0, 0]
M, N, K, act, res, cpl = v[:6]
F16 = len(sys.argv) > 2 and sys.argv[2] == "f16x2"       # the opt-in fp16x2 kernels (ac_linear_f16x2) instead of bf16x3
lib.ac_gemm_set_arith(1)
A = torch.rand(M, K, device=dev) * 2 - 1; W = (torch.rand(N, K, device=dev) * 2 - 1) * 0.05; b = torch.randn(N, device=dev)
R = torch.randn(M, N, device=dev) if res else None
C = torch.empty(M, N, device=dev)
Cp = torch.empty(3 * M * N, dtype=torch.int16, device=dev) if cpl else None
Wp = torch.empty(3 * N * K, dtype=torch.int16, device=dev); Ap = torch.empty(3 * M * K, dtype=torch.int16, device=dev)
st = nv.stream_ptr(dev)
lib.ac_split_bf16x3(nv.ptr(W), K, N, K, nv.ptr(Wp), st); lib.ac_split_bf16x3(nv.ptr(A), K, M, K, nv.ptr(Ap), st)
if F16:
    Wh = torch.empty(2 * N * K, dtype=torch.int16, device=dev); Ah = torch.empty(2 * M * K, dtype=torch.int16, device=dev)
    lib.ac_split_f16x2(nv.ptr(W), K, N, K, 10, nv.ptr(Wh), st); lib.ac_split_f16x2(nv.ptr(A), K, M, K, 6, nv.ptr(Ah), st)
    for _ in range(6):
        nv.check(lib.ac_linear_f16x2(nv.ptr(Ah), nv.ptr(Wh), nv.ptr(b), nv.ptr(R) if res else None, N, None if cpl else nv.ptr(C), N,
                                     nv.ptr(Cp) if cpl else None, M, N, K, act, st), "lin16")
    torch.cuda.synchronize()
    sys.exit(0)
for _ in range(6):
    nv.check(lib.ac_linear_bf16x3(nv.ptr(A), K, nv.ptr(Ap), nv.ptr(W), K, nv.ptr(Wp), nv.ptr(b), nv.ptr(R) if res else None, N,
                                  None if cpl else nv.ptr(C), N, nv.ptr(Cp) if cpl else None, M, N, K, act, st), "lin")
torch.cuda.synchronize()
