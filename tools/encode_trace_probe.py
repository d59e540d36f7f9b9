"""Scratch probe (round 5) for `rocprofv3 --kernel-trace`: a few encoder forwards of the bench batch, nothing else -- to read the
per-kernel durations AND the gaps between consecutive kernels of one forward off the trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
sys.argv = [sys.argv[0]]
import torch
import bench
dev = torch.device("cuda:0")
from adaptive_classifier.encoder import HipBertEncoder
from transformers import BertConfig, BertModel
torch.manual_seed(0)
enc = HipBertEncoder(BertModel(BertConfig(), add_pooling_layer=False).eval(), device=dev)
ids, types, mask = bench.synthetic_tokens(dev, 0)
for _ in range(8):
    enc.encode_cls(ids, types, mask, verify=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    enc.encode_cls(ids, types, mask, verify=False)
e1.record(); torch.cuda.synchronize()
print("encode_ms_by_events", e0.elapsed_time(e1) / 10)
