"""Round 6: the head's forward over a predict batch (256 x 768 -> 768 -> 384 -> 4) and the four CLS-row GEMM shapes of the encoder's
last layer, by HIP events (back to back, 200 launches): the few-tile fp32 kernel (default) against AC_GEMM_FEWTILES=0 (run in two
processes; the switch is read at every call)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import _native as nv
from adaptive_classifier.models import AdaptiveHead
dev = torch.device("cuda:0")
torch.manual_seed(0)
head = AdaptiveHead(768, 4, [768, 384]).to(dev).eval()
x = torch.nn.functional.normalize(torch.randn(256, 768, device=dev), dim=1)
ref = head.model(x)
def t(f, n=200):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for sw in ("1", "0"):
    os.environ["AC_GEMM_FEWTILES"] = sw
    out = head.forward_native(x)
    print("AC_GEMM_FEWTILES=%s head forward %.1f us  max |diff| vs torch %.3g" % (sw, t(lambda: head.forward_native(x)), (out - ref).abs().max().item()))
    for M, N, K in ((256, 768, 768), (256, 3072, 768), (256, 768, 3072), (256, 384, 768)):
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev); C = torch.empty(M, N, device=dev)
        f = lambda: nv.check(nv.lib().ac_linear_f32(nv.ptr(A), K, nv.ptr(W), K, nv.ptr(b), None, N, nv.ptr(C), N, M, N, K, 0, nv.stream_ptr(dev)), "lin")
        us = t(f)
        err = (C.double() - (A.double() @ W.double().T + b.double())).abs().max().item()
        print("   linear %4d x %4d x %4d  %.1f us  max err vs fp64 %.3g" % (M, N, K, us, err))
