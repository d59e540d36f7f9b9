"""Scratch probe: BERT-base encode_cls time vs sequence length (same token count), for same-box A/Bs
(AC_LIBACAMD_PATH selects the library variant)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier.encoder import HipBertEncoder
from transformers import BertConfig, BertModel
dev = torch.device("cuda:0")
torch.manual_seed(0)
hf = BertModel(BertConfig(), add_pooling_layer=False).eval()
enc = HipBertEncoder(hf, device=dev)
for b, S in ((256, 32), (64, 128), (16, 512), (1, 16)):
    ids = torch.randint(1000, 30000, (b, S)).to(dev)
    mask = torch.ones((b, S), dtype=torch.int64, device=dev)
    for _ in range(3): enc.encode_cls(ids, None, mask)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): enc.encode_cls(ids, None, mask)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"b={b} S={S}: {ms:.3f} ms  ({enc.flops(b, S, executed=True) / ms / 1e9:.1f} TF executed)")
