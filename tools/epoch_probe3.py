"""Scratch probe: persistent training epoch vs the step-by-step launches (AC_HEAD_PERSISTENT=0/1 in the environment)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import AdaptiveHead
from adaptive_classifier import index as ix
from adaptive_classifier.training import HeadTrainer
dev = torch.device("cuda:0")
torch.manual_seed(0)
head = AdaptiveHead(768, 4, [768, 384]).to(dev).train()
tr = HeadTrainer(head)
n = 4000
X = ix.synth_unit_rows(n, 768, 5, device=dev)[:, :768].contiguous(); y = (torch.arange(n, device=dev) % 4)
order = torch.randperm(n).to(dev)
with_ewc = len(sys.argv) > 1 and sys.argv[1] == "ewc"
F = torch.rand_like(tr.flat) if with_ewc else None
old = tr.flat.clone() if with_ewc else None
for rep in range(4):
    tr.loss_accum.zero_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps = tr.fused_epoch(X, y, order, 32, 0.1, 1234 + rep, fisher=F, old_params=old, lambda_B=100.0)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"epoch of {steps} steps: enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms ({1e6*(t2-t0)/steps:.2f} us/step)  "
          f"mean loss {tr.loss_accum.item()/steps:.6f} out3 {tr.out3.tolist()} |p| {tr.flat.norm().item():.6f}", flush=True)
