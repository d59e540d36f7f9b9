"""Encoder time of one source tree (argv[1] = its root, argv[2] = a label): bert-base, the bench's ragged 256 x 32 batch.  Run once per
tree in separate processes on the same box -- tools/r04_tree_ab.sh."""
import os, sys
root = sys.argv[1]
sys.path[:0] = [os.path.join(root, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import _native as nv
from adaptive_classifier.encoder import HipBertEncoder
from transformers import BertConfig, BertModel
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = HipBertEncoder(BertModel(BertConfig(), add_pooling_layer=False).eval(), device=dev)
B, S = 256, 32
g = torch.Generator().manual_seed(1234)
ids = torch.randint(1000, 30000, (B, S), generator=g); ids[:, 0] = 101
lens = torch.randint(8, S + 1, (B,), generator=g); lens[0] = S
mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
ids = (ids * mask).to(dev); mask = mask.to(dev); types = torch.zeros_like(ids)
ts = []
for rnd in range(4):
    for _ in range(3): enc.encode_cls(ids, types, mask, verify=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): enc.encode_cls(ids, types, mask, verify=False)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20)
print(f"{sys.argv[2]:28s} encode med {sorted(ts)[len(ts)//2]:.3f} ms  min {min(ts):.3f}  (lib {nv.__file__})")
