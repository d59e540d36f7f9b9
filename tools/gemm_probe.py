"""Scratch perf probe for ac_linear_f32 on the encoder's GEMM shapes."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import _native as nv
dev = torch.device("cuda:0")
shapes = [(8192, 2304, 768), (8192, 768, 768), (8192, 3072, 768), (8192, 768, 3072), (1024, 768, 768), (32768, 3072, 768)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev); C = torch.empty(M, N, device=dev)
    def run():
        nv.check(nv.lib().ac_linear_f32(nv.ptr(A), K, nv.ptr(W), K, nv.ptr(b), None, 0, nv.ptr(C), N, M, N, K, 0,
                                        nv.stream_ptr(dev)), "linear")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ref = (A[:64].double() @ W.double().T + b.double())
    err = (C[:64].double() - ref).abs().max().item()
    print(f"M={M} N={N} K={K}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s  err={err:.2e}", flush=True)
