"""Probe (round 4): the fp16-plane sweep (knn_plane_sweep, 1 .. 64 queries on a prepared store) against the fp32 sweeps on the
same store: kernel time by HIP events around the sweep launch(es) (ac_knn_set_profile_events), whole-call time, ids equal.
    python tools/plane_sweep_probe.py [rows] [dim]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import numpy as np
import torch
from adaptive_classifier import _native as nv
from adaptive_classifier import index as ix

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
k = 32
P = ix.synth_unit_rows(N, D, 1, device=dev)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); prep = ix.prepare_store(P, N, D); t1.record(); torch.cuda.synchronize()
print(f"{N} x {D}: prepare_store {t0.elapsed_time(t1):.1f} ms")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); e1.record(); torch.cuda.synchronize()


def run(nq, prepared, reps=6):
    Q = ix.synth_unit_rows(nq, D, 2, device=dev)
    st = torch.zeros(4, dtype=torch.int32, device=dev)
    out = (torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev))
    ws = torch.empty(max(ix.knn_batch_workspace_bytes(N, D, nq, k), ix.knn_workspace_bytes(N, D, nq, k)), dtype=torch.uint8, device=dev)
    for _ in range(2):
        ix.knn_l2_topk(P, N, D, Q, k, out=out, workspace=ws, stats=st, prepared=prepared)
    torch.cuda.synchronize()
    ker, call = [], []
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nv.lib().ac_knn_set_profile_events(e0.cuda_event, e1.cuda_event)
    try:
        for _ in range(reps):
            c0.record()
            ix.knn_l2_topk(P, N, D, Q, k, out=out, workspace=ws, stats=st, prepared=prepared)
            c1.record(); torch.cuda.synchronize()
            ker.append(e0.elapsed_time(e1)); call.append(c0.elapsed_time(c1))
    finally:
        nv.lib().ac_knn_set_profile_events(None, None)
    return float(np.mean(ker)), float(np.min(ker)), float(np.mean(call)), out[1].clone(), st.tolist()


for nq in (1, 8, 16, 32, 33, 48, 64):
    kp, kpmin, cp, ip, sp = run(nq, prep)
    if nq <= 32:
        kf, kfmin, cf, if_, sf = run(nq, None)
        same = bool(torch.equal(ip, if_))
        print(f"nq {nq:3d}: fp16 plane kernel {kp:.3f} ms (min {kpmin:.3f}) = {N * D * 2 / kp / 1e6:.0f} GB/s of plane bytes, call {cp:.3f} ms, "
              f"form {sp[1]} fallbacks {sp[0]} | fp32 sweep kernel {kf:.3f} ms = {N * D * 4 / kf / 1e6:.0f} GB/s, call {cf:.3f} ms, form {sf[1]} | ids equal {same}")
    else:
        print(f"nq {nq:3d}: fp16 plane kernel {kp:.3f} ms (min {kpmin:.3f}) = {N * D * 2 / kp / 1e6:.0f} GB/s of plane bytes, call {cp:.3f} ms, "
              f"form {sp[1]} fallbacks {sp[0]}")
