"""Scratch probe: ModernBERT-base architecture (random init) encode_cls time."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier.encoder import make_encoder
from oracle import bert_oracle
dev = torch.device("cuda:0")
m = bert_oracle.make_modernbert(768, 22, 12, 1152, vocab=50368, max_pos=8192, local_attention=128, seed=0)
enc = make_encoder(m, device=dev)
for b, S in ((256, 32), (64, 128), (16, 512), (2, 8192), (1, 16)):
    ids = torch.randint(5, 50000, (b, S)).to(dev)
    mask = torch.ones((b, S), dtype=torch.int64, device=dev)
    for _ in range(3): enc.encode_cls(ids, None, mask)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n): enc.encode_cls(ids, None, mask)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"ModernBERT-base b={b} S={S}: {ms:.3f} ms  {b/ms*1e3:.0f} seq/s  ({enc.flops(b, S) / ms / 1e9:.1f} TF)")
