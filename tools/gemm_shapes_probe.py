"""Scratch probe: GEMM time on encoder shapes for the library selected by AC_LIBACAMD_PATH, arith from argv."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import _native as nv
dev = torch.device("cuda:0"); lib = nv.lib()
lib.ac_gemm_set_arith(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
out = []
for (M, N, K) in ((8192, 2304, 768), (8192, 768, 768), (8192, 3072, 768), (8192, 768, 3072), (8192, 8192, 8192)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** .5; b = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev)
    mode = sys.argv[2] if len(sys.argv) > 2 else ""
    Wp = torch.empty(3 * N * K, dtype=torch.int16, device=dev); Ap = torch.empty(3 * M * K, dtype=torch.int16, device=dev)
    lib.ac_split_bf16x3(nv.ptr(W), K, N, K, nv.ptr(Wp), nv.stream_ptr(dev)); lib.ac_split_bf16x3(nv.ptr(A), K, M, K, nv.ptr(Ap), nv.stream_ptr(dev))
    def go():
        if not mode:
            nv.check(lib.ac_linear_f32(nv.ptr(A), K, nv.ptr(W), K, nv.ptr(b), None, N, nv.ptr(C), N, M, N, K, 0, nv.stream_ptr(dev)), "lin")
        else:
            nv.check(lib.ac_linear_bf16x3(nv.ptr(A), K, nv.ptr(Ap) if "a" in mode else None, nv.ptr(W), K, nv.ptr(Wp), nv.ptr(b), None, N,
                                          nv.ptr(C), N, None, M, N, K, 0, nv.stream_ptr(dev)), "lin")
    for _ in range(5): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    out.append(f"{N}x{K}: {ms*1e3:.0f}us {2.0*M*N*K/ms/1e9:.0f}TF")
print(os.environ.get("AC_LIBACAMD_PATH", "default"), sys.argv[1:], " | ".join(out))
