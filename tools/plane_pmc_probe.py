"""PMC target (round 4): a few launches of the fp16-plane sweep (prepared store, 16 resident queries) and of the fp32 ring sweep
over the same 10M x 768 store -- nothing else of note on the device.  Run under rocprofv3 --pmc by tools/r04_sweep_pmc.sh."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import index as ix

dev = torch.device("cuda:0")
N, D, k = 10_000_000, 768, 32
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 16
P = ix.synth_unit_rows(N, D, 1, device=dev)
prep = ix.prepare_store(P, N, D)
Q = ix.synth_unit_rows(nq, D, 2, device=dev)
ws = torch.empty(max(ix.knn_batch_workspace_bytes(N, D, nq, k), ix.knn_workspace_bytes(N, D, nq, k)), dtype=torch.uint8, device=dev)
out = (torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev))
for _ in range(6):
    ix.knn_l2_topk(P, N, D, Q, k, out=out, workspace=ws, prepared=prep)
a = out[1].clone()
for _ in range(6):
    ix.knn_l2_topk(P, N, D, Q, k, out=out, workspace=ws)
torch.cuda.synchronize()
print("ids equal:", bool(torch.equal(a, out[1])))
