"""Scratch probe: prepared-store batched path at configs[2] / configs[4] shapes (env AC_KNN_BATCH_MAP = 0 / 1 A/B)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import index as ix
dev = torch.device("cuda:0")
for (N, D, nq, k) in [(10_000_000, 768, 4096, 32), (2_000_000, 1024, 1024, 32), (10_000_000, 768, 1024, 32)]:
    P = ix.synth_unit_rows(N, D, 1, device=dev); Q = ix.synth_unit_rows(nq, D, 2, device=dev)
    prep = ix.prepare_store(P, N, D)
    ws = torch.empty(ix.knn_batch_workspace_bytes(N, D, nq, k), dtype=torch.uint8, device=dev)
    st = torch.zeros(4, dtype=torch.int32, device=dev)
    out = ix.knn_l2_topk(P, N, D, Q, k, workspace=ws, stats=st, prepared=prep); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): out = ix.knn_l2_topk(P, N, D, Q, k, workspace=ws, stats=st, prepared=prep)
    e1.record(); torch.cuda.synchronize()
    print(f"N={N} D={D} nq={nq}: batch {e0.elapsed_time(e1)/3:.2f} ms  fallbacks {int(st[0])}  checksum {int(out[1].sum())}", flush=True)
    del P, Q, prep, ws, out; torch.cuda.empty_cache()
