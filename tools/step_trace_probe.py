"""Scratch probe for rocprofv3 --kernel-trace: N predict() steps of the bench (configs[1]); AC_STEP_BATCHED=1 forces the
prepared-store (batched) kNN path for the 256-query step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
import bench
from adaptive_classifier import index as ix
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
if os.environ.get("AC_STEP_BATCHED"):
    ix.BATCH_MIN_PAIRS = 1.0e7
clf, hf = bench.make_classifier(dev, 0, 1)
ids, types, mask = bench.synthetic_tokens(dev, 0)
for _ in range(3): bench.predict_step(clf, ids, types, mask)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
n = 10
for _ in range(n): bench.predict_step(clf, ids, types, mask)
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / n * 1e3, bench.time_stages(clf, ids, types, mask))
