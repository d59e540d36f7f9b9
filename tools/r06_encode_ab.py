"""Round 6 same-box A/B of the encoder forward (bench batch: 256 ragged texts = 5141 rows; all 32 tokens = 8192 rows; bert-large
arch, 1024 ragged texts).  One process per arm (argv: label, then KEY=VALUE environment settings are read by the caller's shell):
prints the median / min encode time by HIP events over 4 rounds of 20 forwards."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import _native as nv
from adaptive_classifier.encoder import HipBertEncoder
from transformers import BertConfig, BertModel
label = sys.argv[1]
what = sys.argv[2] if len(sys.argv) > 2 else "base"
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = BertConfig() if what != "large" else BertConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
enc = HipBertEncoder(BertModel(cfg, add_pooling_layer=False).eval(), device=dev)
B, S = (256, 32) if what != "large" else (1024, 32)
g = torch.Generator().manual_seed(1234)
ids = torch.randint(1000, 30000, (B, S), generator=g); ids[:, 0] = 101
lens = torch.randint(8, S + 1, (B,), generator=g); lens[0] = S
if what == "full":
    lens[:] = S
mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
ids = (ids * mask).to(dev); mask = mask.to(dev); types = torch.zeros_like(ids)
ts = []
out = None
for rnd in range(4):
    for _ in range(3): out = enc.encode_cls(ids, types, mask, verify=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20 if what != "large" else 5
    for _ in range(n): enc.encode_cls(ids, types, mask, verify=False)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / n)
chk = float(out.double().sum()), float(out[0, :4].double().abs().sum())
print(f"{label:34s} {what:5s} rows {enc.last_tokens:6d} encode med {sorted(ts)[len(ts)//2]:.3f} ms  min {min(ts):.3f}  checksum {chk[0]:.9f} {chk[1]:.9f}", flush=True)
