"""Scratch probe: whole-call time of the fp32 sweep path vs the prepared-store batched path at the BASELINE shapes."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import index as ix
dev = torch.device("cuda:0")
for (N, D, nq, k) in [(100_000, 768, 256, 16), (2_000_000, 1024, 1024, 32), (10_000_000, 768, 4096, 32), (10_000_000, 768, 256, 32)]:
    P = ix.synth_unit_rows(N, D, 1, device=dev); Q = ix.synth_unit_rows(nq, D, 2, device=dev)
    t0 = time.perf_counter(); prep = ix.prepare_store(P, N, D); torch.cuda.synchronize(); tp = time.perf_counter() - t0
    res = {}
    for name, pr in (("sweep", None), ("batch", prep)):
        st = torch.zeros(4, dtype=torch.int32, device=dev)
        need = ix.knn_batch_workspace_bytes(N, D, nq, k) if pr else ix.knn_workspace_bytes(N, D, nq, k)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        out = ix.knn_l2_topk(P, N, D, Q, k, workspace=ws, stats=st, prepared=pr); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps): out = ix.knn_l2_topk(P, N, D, Q, k, workspace=ws, stats=st, prepared=pr)
        e1.record(); torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / reps, out, int(st[0].item()), need)
    same = torch.equal(res["sweep"][1][1], res["batch"][1][1]) and torch.equal(res["sweep"][1][0], res["batch"][1][0])
    fl = 2.0 * nq * N * D
    print(f"N={N} D={D} nq={nq} k={k}: sweep {res['sweep'][0]:.2f} ms ({fl/res['sweep'][0]/1e9:.0f} TF)  batch {res['batch'][0]:.2f} ms "
          f"({fl/res['batch'][0]/1e9:.0f} TF-equiv)  identical={same}  fallbacks sweep/batch={res['sweep'][2]}/{res['batch'][2]}  "
          f"prepare {tp*1e3:.1f} ms  ws {res['batch'][3]/1e6:.0f} MB", flush=True)
    del P, Q, prep, res, ws, out
    torch.cuda.empty_cache()
