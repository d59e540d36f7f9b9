"""Scratch probe: the batched (prepared-store) kNN call vs the fp32 sweep path at given (N, D, nq, k); ids must agree."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import index as ix
dev = torch.device("cuda:0")
cfgs = [(100_000, 768, 256, 16), (2_000_000, 1024, 1024, 32), (10_000_000, 768, 4096, 32)]
if len(sys.argv) > 1:
    cfgs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for (N, D, nq, k) in cfgs:
    torch.cuda.empty_cache()
    P = ix.synth_unit_rows(N, D, 1, device=dev)
    Q = ix.synth_unit_rows(nq, D, 2, device=dev)
    prep = ix.prepare_store(P, N, D)
    ws = torch.empty(ix.knn_batch_workspace_bytes(N, D, nq, k), dtype=torch.uint8, device=dev)
    st = torch.zeros(4, dtype=torch.int32, device=dev)
    out = (torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev))
    for _ in range(2): ix.knn_l2_topk(P, N, D, Q, k, out=out, workspace=ws, stats=st, prepared=prep)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps): ix.knn_l2_topk(P, N, D, Q, k, out=out, workspace=ws, stats=st, prepared=prep)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nsub = min(nq, 64)
    Ds, Is = ix.knn_l2_topk(P, N, D, Q[:nsub].contiguous(), k)
    same = bool(torch.equal(out[1][:nsub], Is) and torch.equal(out[0][:nsub], Ds))
    e0.record(); ix.knn_l2_topk(P, N, D, Q[:min(nq, 256)].contiguous(), k); e1.record(); torch.cuda.synchronize()
    print(f"N={N} D={D} nq={nq} k={k}: batched {ms:.2f} ms ({nq/ms:.1f} k queries/s), fallbacks={int(st[0])}, "
          f"first {nsub} queries equal the fp32 sweep path: {same}; fp32 sweep of {min(nq,256)} queries {e0.elapsed_time(e1):.2f} ms", flush=True)
    del P, prep, ws
