"""A/B of per-shape GEMM kernel tables INSIDE the encoder (bert-base, the bench's ragged 256 x 32 batch): interleaved rounds,
median encode time per table.  usage: encode_ab.py [--large] "name=NxK=cfg;NxK=cfg" ...   (name=  alone = two-buffer kernels;
name=@builtin = the built-in choice; name=@builtin-nofuse = the same with the LayerNorms as separate launches)"""
import os as _os; _os.environ.setdefault("AC_TEST_HOOKS", "1")  # (the process-wide switches used below are test hooks)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import _native as nv
from adaptive_classifier.encoder import HipBertEncoder
from transformers import BertConfig, BertModel
args = sys.argv[1:]
large = "--large" in args
args = [a for a in args if a != "--large"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = BertConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096) if large else BertConfig()
hf = BertModel(cfg, add_pooling_layer=False).eval()
enc = HipBertEncoder(hf, device=dev)
B, S = (1024, 32) if large else (256, 32)
g = torch.Generator().manual_seed(1234)
ids = torch.randint(1000, 30000, (B, S), generator=g); ids[:, 0] = 101
lens = torch.randint(8, S + 1, (B,), generator=g); lens[0] = S
mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
ids = (ids * mask).to(dev); mask = mask.to(dev); types = torch.zeros_like(ids)
tables = [a.split("=", 1) for a in args] or [["base", ""]]
times = {n: [] for n, _ in tables}
ref = None
for rnd in range(int(os.environ.get("ROUNDS", 3))):
    for name, spec in tables:
        nv.check(nv.lib().ac_gemm_set_pipe_table(None if spec.startswith("@builtin") else spec.encode()), "table")
        nv.check(nv.lib().ac_gemm_set_ln_fusion(0 if spec.endswith("-nofuse") else 1), "ln fusion")
        for _ in range(2): out = enc.encode_cls(ids, types, mask)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n): out = enc.encode_cls(ids, types, mask)
        e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / n)
        if ref is None: ref = out.clone()
        if spec.endswith("-nofuse") or tables[0][1].endswith("-nofuse"): assert (out - ref).abs().max().item() < 5e-6, name
        else: assert torch.equal(out, ref), name       # every table computes the same bits
for name, spec in tables:
    t = sorted(times[name])
    print(f"{name:12s} med {t[len(t)//2]:.3f} ms  min {t[0]:.3f} ms   tokens {enc.last_tokens}   [{spec}]")
