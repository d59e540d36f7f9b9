"""Scratch A/B (round 5): query tiles per XCD of knn_batch_sweep at BASELINE configs[2] on one GPU (4096 x 10M x 768, k = 32).
Run once per setting of AC_KNN_BATCH_B (the library reads it once)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import index as ix
dev = torch.device("cuda:0")
N, D, nq, k = int(os.environ.get("ROWS", 10_000_000)), 768, 4096, 32
P = ix.synth_unit_rows(N, D, 1, device=dev)
Q = ix.synth_unit_rows(nq, D, 2, device=dev)
prep = ix.prepare_store(P, N, D)
ws = torch.empty(ix.knn_batch_workspace_bytes(N, D, nq, k), dtype=torch.uint8, device=dev)
st = torch.zeros(4, dtype=torch.int32, device=dev)
out = (torch.empty((nq, k), device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev))
for _ in range(2):
    ix.knn_l2_topk(P, N, D, Q, k, out=out, workspace=ws, stats=st, prepared=prep)
torch.cuda.synchronize()
reps = int(os.environ.get("REPS", 5))
t0 = time.perf_counter()
for _ in range(reps):
    ix.knn_l2_topk(P, N, D, Q, k, out=out, workspace=ws, stats=st, prepared=prep)
torch.cuda.synchronize()
print("AC_KNN_BATCH_B=%s  %.2f ms per batch  fallbacks %d  ids checksum %d" % (os.environ.get("AC_KNN_BATCH_B", "-"), (time.perf_counter() - t0) / reps * 1e3,
      int(st[0]), int(out[1].sum())))
