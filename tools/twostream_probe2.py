"""Scratch probe: does splitting the ragged BASELINE batch into two halves on two HIP streams (kernel-level overlap of one
half's GEMMs with the other half's LayerNorm / attention / tile epilogues) beat one 256-text launch sequence?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier.encoder import HipBertEncoder
from transformers import BertConfig, BertModel
dev = torch.device("cuda:0")
torch.manual_seed(0)
hf = BertModel(BertConfig(), add_pooling_layer=False).eval()
encs = [HipBertEncoder(hf, device=dev) for _ in range(2)]
B, S = 256, 32
g = torch.Generator().manual_seed(1234)
ids = torch.randint(1000, 30000, (B, S), generator=g); ids[:, 0] = 101
lens = torch.randint(8, S + 1, (B,), generator=g); lens[0] = S
mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
ids, mask = ids.to(dev), mask.to(dev)
def one():
    return encs[0].encode_cls(ids, None, mask)
streams = [torch.cuda.Stream(dev) for _ in range(2)]
def two():
    outs = []
    cur = torch.cuda.current_stream(dev)
    for h in range(2):
        streams[h].wait_stream(cur)
        with torch.cuda.stream(streams[h]):
            sl = slice(h * B // 2, (h + 1) * B // 2)
            outs.append(encs[h].encode_cls(ids[sl], None, mask[sl]))
    for h in range(2): cur.wait_stream(streams[h])
    return torch.cat(outs)
a = one(); b = two(); torch.cuda.synchronize()
print("max diff", (a - b).abs().max().item())
for name, f in (("one stream, 256 texts", one), ("two streams, 2 x 128 texts", two), ("one stream, 256 texts", one)):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
