"""Scratch probe: kernel trace target -- BERT-base encode_cls at b = 1, S = 16 (single predict latency)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier.encoder import HipBertEncoder
from transformers import BertConfig, BertModel
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = HipBertEncoder(BertModel(BertConfig(), add_pooling_layer=False).eval(), device=dev)
ids = torch.randint(1000, 30000, (1, 16)).to(dev)
for _ in range(5): enc.encode_cls(ids)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): enc.encode_cls(ids)
e1.record(); torch.cuda.synchronize()
print(f"b=1 S=16: {e0.elapsed_time(e1)/50*1e3:.0f} us per encode")
for S in (8, 32):
    ids = torch.randint(1000, 30000, (1, S)).to(dev)
    for _ in range(5): enc.encode_cls(ids)
    e0.record()
    for _ in range(50): enc.encode_cls(ids)
    e1.record(); torch.cuda.synchronize()
    print(f"b=1 S={S}: {e0.elapsed_time(e1)/50*1e3:.0f} us per encode")
