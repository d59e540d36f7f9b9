"""Scratch probe (rocprofv3 kernel stats target): the prepared-store batched path at the timed predict step's shape."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import index as ix
dev = torch.device("cuda:0")
N, D, nq, k = 100_000, 768, 256, 16
P = ix.synth_unit_rows(N, D, 1, device=dev); Q = ix.synth_unit_rows(nq, D, 2, device=dev)
prep = ix.prepare_store(P, N, D)
ws = torch.empty(ix.knn_batch_workspace_bytes(N, D, nq, k), dtype=torch.uint8, device=dev)
for _ in range(20): ix.knn_l2_topk(P, N, D, Q, k, workspace=ws, prepared=prep)
torch.cuda.synchronize()
