"""Scratch probe (round 5): which native entry points read workspace bytes they did not write?

A fresh process gets zeroed pages from the driver, so `torch.empty` workspaces look clean; in a long-lived process the caching
allocator hands back blocks full of old data.  Every workspace is filled with 0xFF bytes (fp32 / fp64 NaN, int -1) before the
call and the result compared with a zero-filled run.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from adaptive_classifier import AdaptiveHead, _native as nv
from adaptive_classifier.training import HeadTrainer
from adaptive_classifier.encoder import HipBertEncoder
from adaptive_classifier import index as ix
from oracle import bert_oracle

dev = torch.device("cuda:0")


def fill(t, byte):
    t.view(torch.uint8).fill_(byte)


def report(name, a, b):
    fin = bool(torch.isfinite(b.float()).all())
    same = bool(torch.equal(a, b)) if fin else False
    print(f"{'OK  ' if same else 'BAD '} {name}: poisoned run finite={fin} equal_to_clean={same}" +
          ("" if same or not fin else f" maxdiff={float((a.double() - b.double()).abs().max()):.3g}"))


# ---- head training ------------------------------------------------------------------------------------------------
for D, C, n, B in ((768, 3, 5, 5), (768, 4, 64, 32), (768, 3, 40, 32), (128, 3, 5, 5), (1024, 64, 70, 32)):
    g = torch.Generator().manual_seed(1)
    X = torch.nn.functional.normalize(torch.randn(n, D, generator=g), dim=1).to(dev)
    y = (torch.arange(n) % C).to(dev)
    for stepwise in (False, True):
        res = []
        for byte in (0, 0xFF):
            head = AdaptiveHead(D, C, [D, D // 2]).to(dev)
            tr = HeadTrainer(head)
            ws = tr._workspace(min(B, n))
            fill(ws, byte)
            fill(tr.grads, byte)                        # scratch for the gradients: documented as scratch
            tr.loss_accum.zero_()
            for ep in range(2):
                tr.fused_epoch(X, y, None, min(B, n), 0.1, 1234 + ep, stepwise=stepwise)
            torch.cuda.synchronize()
            res.append(torch.cat([head.flat_params().detach().clone(), tr.loss_accum.clone()]))
        report(f"head_train_epoch D={D} C={C} n={n} B={B} stepwise={stepwise}", res[0], res[1])
    # forward
    res = []
    for byte in (0, 0xFF):
        head = AdaptiveHead(D, C, [D, D // 2]).to(dev)
        for nb in (1, 5, n):
            ws = head._workspace(nb); fill(ws, byte)
        res.append(torch.cat([head.forward_native(X[:nb]).reshape(-1) for nb in (1, 5, n)]))
    report(f"head_forward D={D} C={C}", res[0], res[1])

# ---- encoder ------------------------------------------------------------------------------------------------------
for (H, L, A, I) in ((128, 3, 2, 512), (768, 2, 12, 3072)):
    model = bert_oracle.make_bert(H, L, A, I, vocab=2000, seed=3)
    enc = HipBertEncoder(model, device=dev)
    for (b, S, ragged) in ((1, 5, False), (2, 16, False), (4, 12, True), (24, 16, False), (24, 16, True), (40, 32, True)):
        ids, types, mask = bert_oracle.synthetic_batch(b, S, vocab=2000, seed=9, ragged=ragged)
        res = []
        for byte in (0, 0xFF):
            enc._ws = None
            enc.encode_cls(ids, types, mask)                # sizes the workspace
            fill(enc._ws, byte)
            enc._ws[:256].zero_()                            # the verdict head is documented as caller-cleared
            res.append(enc.encode_cls(ids, types, mask, verify=False).clone())
        report(f"bert_encode H={H} L={L} b={b} S={S} ragged={ragged} one_launch={enc.last_one_launch}", res[0], res[1])

# ---- kNN ----------------------------------------------------------------------------------------------------------
for (N, nq, k) in ((1000, 1, 5), (1000, 7, 16), (100_000, 16, 16), (100_000, 256, 16), (300_000, 40, 32)):
    P = ix.synth_unit_rows(N, 768, 1, device=dev)
    Q = ix.synth_unit_rows(nq, 768, 2, device=dev)
    for prepared in (False, True):
        if prepared and not ix.batch_applies(N, nq, k):
            continue
        res = []
        for byte in (0, 0xFF):
            prep = ix.prepare_store(P, N, 768) if prepared else None
            need = max(ix.knn_workspace_bytes(N, 768, nq, k), ix.knn_batch_workspace_bytes(N, 768, nq, k) if prepared else 0)
            ws = torch.empty(need, dtype=torch.uint8, device=dev); fill(ws, byte)
            stats = torch.zeros(4, dtype=torch.int32, device=dev)
            D_, I_ = ix.knn_l2_topk(P, N, 768, Q, k, workspace=ws, stats=stats, prepared=prep)
            res.append(torch.cat([D_.reshape(-1).double(), I_.reshape(-1).double()]))
        report(f"knn N={N} nq={nq} k={k} prepared={prepared}", res[0], res[1])
print("done")
