"""A/B of the rotated k order of the ring-staged GEMMs (ac_gemm_set_krot) inside the encoder: bert-base on the bench's ragged
256 x 32 batch (or --large: bert-large arch, 1024 texts), both arithmetics, interleaved rounds."""
import os as _os; _os.environ.setdefault("AC_TEST_HOOKS", "1")  # (the process-wide switches used below are test hooks)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import _native as nv
from adaptive_classifier.encoder import HipBertEncoder
from transformers import BertConfig, BertModel
large = "--large" in sys.argv
dev = torch.device("cuda:0"); lib = nv.lib()
torch.manual_seed(0)
cfg = BertConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096) if large else BertConfig()
enc = HipBertEncoder(BertModel(cfg, add_pooling_layer=False).eval(), device=dev).enable_f16x2()
B, S = (1024, 32) if large else (256, 32)
g = torch.Generator().manual_seed(1234)
ids = torch.randint(1000, 30000, (B, S), generator=g); ids[:, 0] = 101
lens = torch.randint(8, S + 1, (B,), generator=g); lens[0] = S
mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
ids = (ids * mask).to(dev); mask = mask.to(dev); types = torch.zeros_like(ids)
variants = [("bf16x3 in order", 1, 0), ("bf16x3 rotated", 1, 1), ("f16x2 in order", 2, 0), ("f16x2 rotated", 2, 1)]
times = {n: [] for n, _, _ in variants}; outs = {}
for rnd in range(4):
    for name, arith, krot in variants:
        lib.ac_gemm_set_arith(arith); lib.ac_gemm_set_krot(krot)
        for _ in range(2): enc.encode_cls(ids, types, mask, verify=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out = enc.encode_cls(ids, types, mask, verify=False)
        e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 10); outs[name] = out.clone()
lib.ac_gemm_set_arith(1); lib.ac_gemm_set_krot(0)
for name, _, _ in variants:
    t = sorted(times[name])
    print(f"{name:18s} med {t[len(t)//2]:.3f} ms  min {t[0]:.3f}   max |emb - in-order bf16x3| {(outs[name] - outs['bf16x3 in order']).abs().max().item():.2e}   tokens {enc.last_tokens}")
