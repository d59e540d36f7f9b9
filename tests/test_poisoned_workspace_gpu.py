"""GPU suite: no native entry point may read workspace bytes it did not write (round 5).

A fresh process gets zeroed pages from the driver, so `torch.empty` workspaces look clean; in a long-lived process the caching
allocator hands back blocks full of old data.  Found by the reference's own tests/test_classifier.py::test_prediction running
against the product in one process with other tests: the persistent training epoch (head_epoch.hip) with a batch of 5 < 32 rows
multiplied never-written rows of its activation scratch by zero weights -- NaN x 0 -- and the head came out all NaN.
Every workspace is filled with 0xFF bytes (fp32 / fp64 NaN, int -1) before the call; the result must equal the zero-filled run
bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fill(t, byte):
    t.view(torch.uint8).fill_(byte)


@pytest.mark.parametrize("D,C,n,B", [(768, 3, 5, 5), (768, 4, 64, 32), (768, 3, 40, 32), (128, 3, 5, 5), (128, 2, 1, 1),
                                     (1024, 64, 70, 32), (768, 3, 33, 32)])
@pytest.mark.parametrize("stepwise", [False, True])
def test_head_training_ignores_workspace_garbage(cuda_dev, D, C, n, B, stepwise):
    from adaptive_classifier import AdaptiveHead
    from adaptive_classifier.training import HeadTrainer
    g = torch.Generator().manual_seed(1)
    X = torch.nn.functional.normalize(torch.randn(n, D, generator=g), dim=1).to(cuda_dev)
    y = (torch.arange(n) % C).to(cuda_dev)
    res = []
    for byte in (0, 0xFF):
        head = AdaptiveHead(D, C, [D, D // 2]).to(cuda_dev)
        tr = HeadTrainer(head)
        _fill(tr._workspace(min(B, n)), byte)
        _fill(tr.grads, byte)                            # the gradient block is scratch to the step (written before it is read)
        tr.loss_accum.zero_()
        for ep in range(2):
            tr.fused_epoch(X, y, None, min(B, n), 0.1, 1234 + ep, stepwise=stepwise)
        torch.cuda.synchronize()
        res.append(torch.cat([head.flat_params().detach().clone(), tr.loss_accum.clone()]))
    assert torch.isfinite(res[1]).all()
    assert torch.equal(res[0], res[1])


def test_head_forward_ignores_workspace_garbage(cuda_dev):
    from adaptive_classifier import AdaptiveHead
    X = torch.nn.functional.normalize(torch.randn(300, 768, generator=torch.Generator().manual_seed(2)), dim=1).to(cuda_dev)
    res = []
    for byte in (0, 0xFF):
        head = AdaptiveHead(768, 5, [768, 384]).to(cuda_dev)
        outs = []
        for nb in (1, 5, 33, 300):
            _fill(head._workspace(nb), byte)
            outs.append(head.forward_native(X[:nb]).reshape(-1))
        res.append(torch.cat(outs))
    assert torch.isfinite(res[1]).all() and torch.equal(res[0], res[1])


@pytest.mark.parametrize("H,L,A,I", [(128, 3, 2, 512), (768, 2, 12, 3072)])
def test_encoder_ignores_workspace_garbage(cuda_dev, H, L, A, I):
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    enc = HipBertEncoder(bert_oracle.make_bert(H, L, A, I, vocab=2000, seed=3), device=cuda_dev)
    for (b, S, ragged) in ((1, 5, False), (2, 16, False), (4, 12, True), (24, 16, False), (24, 16, True), (40, 32, True)):
        ids, types, mask = bert_oracle.synthetic_batch(b, S, vocab=2000, seed=9, ragged=ragged)
        res = []
        for byte in (0, 0xFF):
            enc._ws = None
            enc.encode_cls(ids, types, mask)                # sizes the workspace
            _fill(enc._ws, byte)
            enc._ws[:256].zero_()                            # (the verdict words: cleared by the caller, include/acamd.h)
            res.append(enc.encode_cls(ids, types, mask, verify=False).clone())
        assert torch.isfinite(res[1]).all() and torch.equal(res[0], res[1]), (b, S, ragged)


@pytest.mark.parametrize("N,nq,k", [(1000, 1, 5), (1000, 7, 16), (100_000, 16, 16), (100_000, 256, 16), (300_000, 40, 32)])
def test_knn_ignores_workspace_garbage(cuda_dev, N, nq, k):
    from adaptive_classifier import index as ix
    P = ix.synth_unit_rows(N, 768, 1, device=cuda_dev)
    Q = ix.synth_unit_rows(nq, 768, 2, device=cuda_dev)
    for prepared in (False, True):
        if prepared and not ix.batch_applies(N, nq, k):
            continue
        res = []
        for byte in (0, 0xFF):
            prep = ix.prepare_store(P, N, 768) if prepared else None
            need = max(ix.knn_workspace_bytes(N, 768, nq, k), ix.knn_batch_workspace_bytes(N, 768, nq, k) if prepared else 0)
            ws = torch.empty(need, dtype=torch.uint8, device=cuda_dev)
            _fill(ws, byte)
            stats = torch.zeros(4, dtype=torch.int32, device=cuda_dev)
            D_, I_ = ix.knn_l2_topk(P, N, 768, Q, k, workspace=ws, stats=stats, prepared=prep)
            res.append((D_.clone(), I_.clone()))
        assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][0], res[1][0]), prepared


def test_classifier_in_a_process_with_a_dirty_allocator(cuda_dev):
    """The scenario of the reference's tests/test_classifier.py::test_prediction: five texts, three classes, then predict --
    with torch's caching allocator full of NaN-patterned free blocks."""
    from adaptive_classifier import AdaptiveClassifier
    from adaptive_classifier.encoder import HipBertEncoder
    from helpers import HashTokenizer, small_bert
    keep = []
    for sz in (256, 4096, 65536, 1 << 20, 1 << 22, 1 << 24, 1 << 26):
        for _ in range(6):
            keep.append(torch.full((sz // 4,), float("nan"), device=cuda_dev))
    torch.cuda.synchronize()
    del keep
    clf = AdaptiveClassifier("synthetic", device="cuda:0", encoder=HipBertEncoder(small_bert(hidden=768, heads=12, inter=1024),
                                                                                  device=cuda_dev), tokenizer=HashTokenizer())
    clf.add_examples(["This is amazing", "Terrible experience", "Just okay", "Love it", "Hate it"],
                     ["positive", "negative", "neutral", "positive", "negative"])
    assert np.isfinite(clf.last_train_info["final_loss"]) and torch.isfinite(clf.adaptive_head.flat_params()).all()
    preds = clf.predict("This is fantastic")
    assert len(preds) == 3 and all(s == s for _, s in preds) and abs(sum(s for _, s in preds) - 1.0) < 1e-6
