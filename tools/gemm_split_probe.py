"""Scratch probe: fp32-MFMA vs bf16x3-split GEMM time on the encoder's shapes + encoder end to end."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import _native as nv
dev = torch.device("cuda:0")
lib = nv.lib()
def t(M, N, K, act, res, mode):
    lib.ac_gemm_set_arith(mode)
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** .5; b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if res else None; C = torch.empty(M, N, device=dev)
    def go():
        nv.check(lib.ac_linear_f32(nv.ptr(A), K, nv.ptr(W), K, nv.ptr(b), nv.ptr(R) if res else None, N, nv.ptr(C), N,
                                   M, N, K, act, nv.stream_ptr(dev)), "lin")
    for _ in range(5): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50
for (M, N, K, act, res) in ((8192, 2304, 768, 0, False), (8192, 768, 768, 0, True), (8192, 3072, 768, 2, False),
                            (8192, 768, 3072, 0, True), (8192, 8192, 8192, 0, False)):
    a, b = t(M, N, K, act, res, 0), t(M, N, K, act, res, 1)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: f32 {a*1e3:.0f} us ({fl/a/1e9:.0f} TF)   bf16x3 {b*1e3:.0f} us ({fl/b/1e9:.0f} TF-equiv)  x{a/b:.2f}")
from adaptive_classifier.encoder import HipBertEncoder
from transformers import BertConfig, BertModel
torch.manual_seed(0)
enc = HipBertEncoder(BertModel(BertConfig(), add_pooling_layer=False).eval(), device=dev)
ids = torch.randint(1000, 30000, (256, 32)).to(dev)
for mode in (0, 1, 0, 1):
    lib.ac_gemm_set_arith(mode)
    for _ in range(3): enc.encode_cls(ids)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): enc.encode_cls(ids)
    e1.record(); torch.cuda.synchronize()
    print(f"encoder 256x32 arith={mode}: {e0.elapsed_time(e1)/20:.3f} ms")
