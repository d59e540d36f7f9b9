"""Scratch probe: is the fused training epoch host-enqueue bound or GPU bound?  (enqueue time vs total per epoch)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import AdaptiveHead
from adaptive_classifier import index as ix
from adaptive_classifier.training import HeadTrainer
dev = torch.device("cuda:0")
head = AdaptiveHead(768, 4, [768, 384]).to(dev).train()
tr = HeadTrainer(head)
n = 4000
X = ix.synth_unit_rows(n, 768, 5, device=dev)[:, :768].contiguous(); y = (torch.arange(n, device=dev) % 4)
order = torch.randperm(n).to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps = tr.fused_epoch(X, y, order, 32, 0.1, 1234)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"epoch of {steps} steps: enqueue {1e3*(t1-t0):.2f} ms ({1e6*(t1-t0)/steps:.1f} us/step), total {1e3*(t2-t0):.2f} ms ({1e6*(t2-t0)/steps:.1f} us/step)", flush=True)
# the same epoch captured in a HIP graph (arguments baked: timing only)
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    tr.fused_epoch(X, y, order, 32, 0.1, 1234)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"graph replay: enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms ({1e6*(t2-t0)/125:.1f} us/step)", flush=True)
