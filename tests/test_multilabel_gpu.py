"""GPU suite for the N3 widening: MultiLabelAdaptiveHead / MultiLabelAdaptiveClassifier on the HIP path
against fixtures produced by the reference's multilabel.py (tests/golden/multilabel.json)."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import HashTokenizer, small_bert
from oracle import head_oracle, synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _check_summary(t, s, atol):
    flat = t.detach().reshape(-1).double().cpu().numpy()
    idx = np.asarray(s["idx"]) % flat.size
    assert np.abs(flat[idx] - np.asarray(s["vals"])).max() < atol


def test_reference_bce_and_ce_sigmoid_steps(cuda_dev):
    from adaptive_classifier import MultiLabelAdaptiveHead
    from adaptive_classifier.training import LOSS_BCE_SIGMOID, LOSS_CE_SIGMOID, HeadTrainer
    g = json.load(open(os.path.join(G, "multilabel.json")))
    m = np.load(os.path.join(G, "multilabel_masks.npz"))
    D, C, B = 768, 5, 32
    torch.manual_seed(7)
    head = MultiLabelAdaptiveHead(D, C, [D, D // 2])
    for k, v in head.state_dict().items():
        _check_summary(v, g["init"][k], 1e-9)
    head = head.to(cuda_dev)
    tr = HeadTrainer(head)
    X = torch.from_numpy(synth.synth_unit_rows(B, D, g["x_seed"])).to(cuda_dev)
    T = torch.from_numpy(((np.arange(B)[:, None] * 3 + np.arange(C)[None, :] * 5) % 7 < 2).astype(np.float32)).to(cuda_dev)
    y = torch.from_numpy((np.arange(B) * 3 % C).astype(np.int64)).to(cuda_dev)
    for s, step in enumerate(g["steps"]):
        m1, m2 = (torch.from_numpy(m[f"m{i}_{s}"]).to(cuda_dev) for i in (1, 2))
        if step["kind"] == "bce":
            tr.forward_backward_loss(X, targets=T, loss_kind=LOSS_BCE_SIGMOID, mask1=m1, mask2=m2)
        else:
            tr.forward_backward_loss(X, y=y, loss_kind=LOSS_CE_SIGMOID, mask1=m1, mask2=m2)
        out = tr.optimizer_step()
        assert abs(tr.loss.item() - step["loss"]) < 1e-4, step["kind"]
        assert abs(out[1].item() - step["grad_norm"]) < 1e-4
        for k, v in head.state_dict().items():
            _check_summary(v, step["params"][k], 5e-5)
    head.eval()
    with torch.no_grad():
        probs = head(X[:4])
    assert np.abs(probs.cpu().numpy() - np.asarray(g["probs_after"])).max() < 1e-4


def test_fused_bce_step_matches_oracle(cuda_dev):
    """ac_head_train_step with AC_LOSS_BCE_SIGMOID, gathered targets, no dropout."""
    from adaptive_classifier import MultiLabelAdaptiveHead
    from adaptive_classifier.training import LOSS_BCE_SIGMOID, HeadTrainer
    D, C, n, B = 128, 6, 90, 32
    ref = head_oracle.make_multilabel_head(D, C, [D, D // 2], seed=3).train()
    opt = torch.optim.AdamW(ref.parameters(), lr=0.001, weight_decay=0.01)
    torch.manual_seed(3)
    head = MultiLabelAdaptiveHead(D, C, [D, D // 2]).to(cuda_dev)
    tr = HeadTrainer(head)
    g = torch.Generator().manual_seed(0)
    X = torch.nn.functional.normalize(torch.randn(n, D, generator=g), dim=1)
    T = (torch.rand(n, C, generator=g) < 0.3).float()
    Xd, Td = X.to(cuda_dev), T.to(cuda_dev)
    for step in range(4):
        idx = torch.randperm(n, generator=g)[:B]
        loss, gn = head_oracle.train_step_loss(ref, opt, X[idx], T[idx], "bce", masks=None)
        out = tr.fused_step(Xd, None, idx.to(cuda_dev), 0.0, 0, loss_kind=LOSS_BCE_SIGMOID, targets_all=Td)
        assert abs(out[0].item() - loss) < 1e-4 and abs(out[2].item() - gn) < 1e-4 * max(1.0, gn)
        assert (tr.flat.cpu() - head_oracle.flat(ref)).abs().max().item() < 5e-5


def test_multilabel_classifier_api(cuda_dev):
    """Mirrors the reference's tests/test_multilabel.py behaviours with the offline encoder/tokenizer."""
    from adaptive_classifier import MultiLabelAdaptiveClassifier
    from adaptive_classifier.encoder import HipBertEncoder
    enc = HipBertEncoder(small_bert(), device=cuda_dev)
    c = MultiLabelAdaptiveClassifier("synthetic", device="cuda:0", encoder=enc, tokenizer=HashTokenizer(),
                                     min_predictions=1, max_predictions=3)
    texts = ["ai research in hospitals", "solar farms and climate policy", "election results and the economy",
             "new vaccine trial", "stock market rally", "machine learning chips"]
    labels = [["technology", "healthcare"], ["environment", "politics"], ["politics", "business"],
              ["healthcare", "science"], ["business"], ["technology", "science"]]
    c.add_examples(texts, labels)
    assert set(c.label_to_id) == {"technology", "healthcare", "environment", "politics", "business", "science"}
    assert c.adaptive_head.num_classes == 6 and c.memory.index.ntotal == 6
    assert sum(len(v) for v in c.memory.examples.values()) == 11          # one example per (text, label)
    assert set(c.label_thresholds) == set(c.label_to_id)
    p = c.predict_multilabel("ai research in hospitals")
    assert 1 <= len(p) <= 3 and all(isinstance(l, str) and isinstance(s, float) and 0 <= s <= 1 for l, s in p)
    assert p == sorted(p, key=lambda x: -x[1])
    saved, c.label_thresholds = c.label_thresholds, {}     # per-label thresholds take precedence (multilabel.py:178)
    assert len(c.predict_multilabel("ai research", threshold=0.999)) == 1   # min_predictions guarantee
    c.label_thresholds = saved
    assert len(c.predict("ai research in hospitals", k=2)) <= 2
    batch = c.predict_multilabel_batch(texts[:3])
    assert [l for l, _ in batch[0]] == [l for l, _ in c.predict_multilabel(texts[0])]
    with pytest.raises(ValueError):
        c.predict_multilabel("")
    with pytest.raises(ValueError):
        c.add_examples(["a"], [["x"], ["y"]])
    st = c.get_label_statistics()
    assert st["default_threshold"] == 0.5 and st["adaptive_threshold"] == 0.5 * 0.6 and st["max_predictions"] == 3
    # new labels on an existing classifier: base-class new-class loop (CE on sigmoid outputs) + grown head
    np.random.seed(0)
    c.add_examples(["quantum sensors in sport"], [["sports", "science"]])
    assert c.adaptive_head.num_classes == 7 and "sports" in c.label_to_id
    assert len(c.predict_multilabel("quantum sensors in sport")) >= 1
