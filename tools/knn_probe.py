"""Scratch perf probe for the kNN sweep (not part of the product or tests)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import index as ix

dev = torch.device("cuda:0")
cfgs = [(100_000, 768, 1, 16), (100_000, 768, 16, 16), (100_000, 768, 32, 16), (100_000, 768, 256, 16),
        (1_000_000, 768, 1, 32), (1_000_000, 768, 16, 32), (1_000_000, 768, 32, 32), (1_000_000, 768, 256, 32),
        (10_000_000, 768, 16, 32), (10_000_000, 768, 32, 32)]
if len(sys.argv) > 1:
    cfgs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
P = None
for (N, D, nq, k) in cfgs:
    if P is None or P.shape[0] != N or P.shape[1] != D:
        P = None
        torch.cuda.empty_cache()
        P = ix.synth_unit_rows(N, D, 1, device=dev)
    Q = ix.synth_unit_rows(nq, D, 2, device=dev)
    ws = torch.empty(ix.knn_workspace_bytes(N, D, nq, k), dtype=torch.uint8, device=dev)
    stats = torch.zeros(4, dtype=torch.int32, device=dev)
    out = (torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev))
    for _ in range(2):
        ix.knn_l2_topk(P, N, D, Q, k, out=out, workspace=ws, stats=stats)
    torch.cuda.synchronize()
    reps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ix.knn_l2_topk(P, N, D, Q, k, out=out, workspace=ws, stats=stats)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = N * D * 4 / 1e9
    tf = 2.0 * nq * N * D / 1e12
    print(f"N={N} D={D} nq={nq} k={k}: {ms:.3f} ms  {gb/ms*1e3:.0f} GB/s (algorithmic)  {tf/ms*1e3:.1f} TFLOP/s  fallbacks={int(stats[0])}", flush=True)
