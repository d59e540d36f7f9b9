"""Scratch probe: does the GPU overlap with the enqueue of a long epoch call?  CPU time of ac_head_train_epoch vs
time to completion, against the per-step Python loop."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier.models import AdaptiveHead
from adaptive_classifier.training import HeadTrainer
dev = torch.device("cuda:0")
n, D, C, B = 4000, 768, 4, 32
head = AdaptiveHead(D, C, [D, D // 2]).to(dev); tr = HeadTrainer(head)
X = torch.nn.functional.normalize(torch.randn(n, D), dim=1).to(dev); y = (torch.arange(n) % C).to(dev)
order = torch.randperm(n).to(dev)
for mode in ("epoch", "steps", "epoch", "steps"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == "epoch":
        k = tr.fused_epoch(X, y, order, B, 0.1, 123)
    else:
        k = 0
        for off in range(0, n, B):
            tr.fused_step(X, y, order[off:off + B], 0.1, 123 + k); k += 1
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{mode}: {k} steps, enqueue {(t1-t0)/k*1e6:.0f} us/step, total {(t2-t0)/k*1e6:.0f} us/step")
