"""Scratch probe: single-query predict latency and head training steps/s (latency-bound paths)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from adaptive_classifier import AdaptiveClassifier, AdaptiveHead, Example
from adaptive_classifier.encoder import HipBertEncoder
from adaptive_classifier.training import HeadTrainer
from transformers import BertConfig, BertModel
from oracle import synth
dev = torch.device("cuda:0")
torch.manual_seed(0)
hf = BertModel(BertConfig(), add_pooling_layer=False).eval()
enc = HipBertEncoder(hf, device=dev)
clf = AdaptiveClassifier("x", device="cuda:0", encoder=enc, tokenizer=None)
labels = [f"c{i}" for i in range(4)]
X = synth.synth_unit_rows(100, 768, 3)
clf.add_embeddings([f"t{i}" for i in range(100)], [torch.from_numpy(x) for x in X], [labels[i % 4] for i in range(100)])
print("train info", clf.last_train_info)
for S in (16, 64):
    ids = torch.randint(1000, 30000, (1, S)); ids[:, 0] = 101
    ids_d = ids.to(dev)
    def one():
        emb = clf.model.encode_cls(ids_d)
        S_, I_, P_ = clf._device_stage(emb, 4)
        return clf._finish(S_, I_, P_, 3, True, b=1)
    for _ in range(5): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n): r = one()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): clf.model.encode_cls(ids_d)
    e1.record(); torch.cuda.synchronize()
    print(f"single predict S={S}: {dt*1e3:.3f} ms end-to-end; encoder alone {e0.elapsed_time(e1)/n:.3f} ms", r[0][:2])
# training steps/s
head = AdaptiveHead(768, 4, [768, 384]).to(dev)
tr = HeadTrainer(head)
Xb = torch.from_numpy(synth.synth_unit_rows(32, 768, 5)).to(dev); yb = (torch.arange(32) % 4).to(dev)
m1, m2 = tr.dropout_masks(32)
for _ in range(20): tr.step(Xb, yb, m1, m2)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 500
for _ in range(n): tr.step(Xb, yb, m1, m2)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"train step (B=32, fixed masks): {dt*1e6:.1f} us/step = {1/dt:.0f} steps/s")
t0 = time.perf_counter()
for _ in range(n):
    a, b = tr.dropout_masks(32); tr.step(Xb, yb, a, b)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"train step incl. mask generation: {dt*1e6:.1f} us/step = {1/dt:.0f} steps/s")
# reference on CPU for comparison (torch)
from oracle import head_oracle
ref = head_oracle.make_head(768, 4).train(); opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=0.01)
Xc, yc = Xb.cpu(), yb.cpu()
torch.set_num_threads(16)
for _ in range(5): head_oracle.train_step(ref, opt, Xc, yc)
t0 = time.perf_counter()
for _ in range(50): head_oracle.train_step(ref, opt, Xc, yc)
print(f"torch CPU reference step: {(time.perf_counter()-t0)/50*1e6:.0f} us/step")
tr.loss_accum.zero_()
Xall = torch.from_numpy(synth.synth_unit_rows(4000, 768, 6)).to(dev); yall = (torch.arange(4000) % 4).to(dev)
order = torch.randperm(4000).to(dev)
for i in range(20): tr.fused_step(Xall, yall, order[(i % 100) * 32:(i % 100) * 32 + 32], 0.1, i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(n): tr.fused_step(Xall, yall, order[(i % 100) * 32:(i % 100) * 32 + 32], 0.1, i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"fused train step (gather + in-kernel dropout, one call): {dt*1e6:.1f} us/step = {1/dt:.0f} steps/s")
