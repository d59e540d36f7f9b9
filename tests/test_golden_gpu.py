"""GPU suite: the HIP product path against the fixtures generated from the reference itself
(tests/golden/gen_golden.py) -- the reference is not present on the GPU box."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_knn_cases(cuda_dev):
    from adaptive_classifier.index import HipFlatL2Index
    cases = json.load(open(os.path.join(G, "knn_cases.json")))
    for name, c in cases.items():
        if name == "dups":
            P = np.concatenate([synth.synth_unit_rows(10, 768, 9)] * 3)
            Q, k = P[:2], 6
        else:
            P = synth.synth_unit_rows(c["N"], c["D"], c["seed"])
            Q = synth.synth_unit_rows(c["nq"], c["D"], c["seed"] + 100)
            k = c["k"]
        idx = HipFlatL2Index(P.shape[1], device=cuda_dev)
        idx.add(P)                                   # faiss protocol: numpy in, numpy out
        D, I = idx.search(Q, k)
        assert D.dtype == np.float32 and I.dtype == np.int64
        assert np.array_equal(I, np.asarray(c["I"])), name
        assert np.allclose(D, np.asarray(c["D_out"], np.float32), rtol=2e-7, atol=0), name


def test_router_fixture_memory_scores_and_logits(cuda_dev):
    from adaptive_classifier import AdaptiveHead, PrototypeMemory
    f = np.load(os.path.join(G, "router_fixture.npz"))
    labels = ["HIGH", "LOW"]
    mem = PrototypeMemory(768, device=cuda_dev)
    for l, p in zip(labels, f["protos"]):
        mem.prototypes[l] = torch.from_numpy(p)
    mem._restore_from_save()
    for i in range(10):
        res = mem.get_nearest_prototypes(torch.from_numpy(f["emb"][i]), k=2)
        assert [labels.index(l) for l, _ in res] == f["order"][i].tolist()
        assert all(isinstance(s, float) for _, s in res)
        assert np.allclose([s for _, s in res], f["scores"][i], atol=1e-6)
        assert abs(sum(s for _, s in res) - 1.0) < 1e-5
    head = AdaptiveHead(768, 2, [768, 384]).to(cuda_dev).eval()
    with torch.no_grad():
        got = head(torch.from_numpy(f["emb"]).to(cuda_dev)).cpu().numpy()
    assert np.abs(got - f["seed42_logits"]).max() < 1e-4


def _check_summary(t, s, atol):
    flat = t.detach().reshape(-1).double().cpu().numpy()
    idx = np.asarray(s["idx"]) % flat.size
    assert flat.size == s["n"]
    assert np.abs(flat[idx] - np.asarray(s["vals"])).max() < atol


def test_reference_training_steps(cuda_dev):
    from adaptive_classifier import AdaptiveHead
    from adaptive_classifier.training import HeadTrainer
    g = json.load(open(os.path.join(G, "head_step.json")))
    m = np.load(os.path.join(G, "head_step_masks.npz"))
    X = torch.from_numpy(synth.synth_unit_rows(32, 768, g["x_seed"])).to(cuda_dev)
    y = torch.from_numpy((np.arange(32) * 7 % 4).astype(np.int64)).to(cuda_dev)
    head = AdaptiveHead(768, 4, [768, 384]).to(cuda_dev)
    tr = HeadTrainer(head)
    for s, step in enumerate(g["steps"]):
        loss, out = tr.step(X, y, torch.from_numpy(m[f"m1_{s}"]).to(cuda_dev), torch.from_numpy(m[f"m2_{s}"]).to(cuda_dev))
        assert abs(loss.item() - step["loss"]) < 1e-4
        assert abs(out[1].item() - step["grad_norm"]) < 1e-4
        for k, v in head.state_dict().items():
            _check_summary(v, step["params"][k], 2e-5)
        assert np.abs(head.model[-1].bias.detach().cpu().numpy() - np.asarray(step["out_bias"])).max() < 2e-5


def test_reference_ewc_fisher_and_penalty(cuda_dev):
    from adaptive_classifier import EWC, AdaptiveHead
    g = json.load(open(os.path.join(G, "ewc.json")))
    X = torch.from_numpy(synth.synth_unit_rows(20, 768, 31))
    head = AdaptiveHead(768, 3, [768, 384]).to(cuda_dev)
    ewc = EWC.__new__(EWC)
    ewc.model, ewc.device, ewc.ewc_lambda, ewc._native = head, str(cuda_dev), 100.0, True
    ewc.old_flat = head.flat_params().detach().clone()
    head.eval()
    ewc.fisher_info = ewc._compute_fisher_native([(X[g["order"]], None)],
                                                 sampled_labels=[torch.tensor(g["sampled"]).to(cuda_dev)])
    for name, summ in g["fisher"].items():
        ref_scale = max(abs(v) for v in summ["vals"]) + 1e-12
        _check_summary(ewc.fisher_info[name], summ, 1e-4 * ref_scale + 1e-10)
    with torch.no_grad():
        assert ewc.ewc_loss(batch_size=32).item() == 0.0
        for p in head.parameters():
            p += 0.1
        assert abs(ewc.ewc_loss().item() - g["loss_p01"]) < 1e-4
        assert abs(ewc.ewc_loss(batch_size=32).item() - g["loss_p01_b32"]) < 1e-4


def _golden_classifier(cuda_dev):
    """Same construction as gen_golden.build_memory / gen_memory_and_blend, on the product classes."""
    from adaptive_classifier import AdaptiveClassifier, AdaptiveHead, Example, ModelConfig, PrototypeMemory
    C, per, D = 4, 25, 768
    labels = [f"c{c}" for c in range(C)]
    mem = PrototypeMemory(D, device=cuda_dev)
    X = synth.synth_unit_rows(C * per, D, 10)
    cent = synth.synth_unit_rows(C, D, 11)
    for i in range(C * per):
        c = i % C
        v = X[i] * 0.5 + cent[c]
        v = (v / np.linalg.norm(v)).astype(np.float32)
        mem.add_example(Example(f"t{i:03d}", labels[c], torch.from_numpy(v)), labels[c])
    mem._rebuild_index()
    Q = synth.synth_unit_rows(8, D, 77)
    Q = np.stack([(q * 0.5 + cent[i % 4]) / np.linalg.norm(q * 0.5 + cent[i % 4]) for i, q in enumerate(Q)]).astype(np.float32)
    clf = AdaptiveClassifier.__new__(AdaptiveClassifier)
    clf.config = ModelConfig(); clf.device = str(cuda_dev); clf.memory = mem; clf.embedding_dim = D
    clf.label_to_id = {l: i for i, l in enumerate(labels)}; clf.id_to_label = {i: l for i, l in enumerate(labels)}
    clf.training_history = {"c0": 25, "c1": 5, "c2": 25, "c3": 9}
    clf.adaptive_head = AdaptiveHead(D, 4, [768, 384]).to(cuda_dev)
    texts = [f"q{i}" for i in range(8)]
    table = {t: torch.from_numpy(q) for t, q in zip(texts, Q)}
    clf._embed_device = lambda ts, **kw: torch.stack([table[t] for t in ts]).to(cuda_dev)
    return clf, mem, texts, Q


def _same_preds(got, want, tol=1e-5):
    assert [l for l, _ in got] == [l for l, _ in want], (got, want)
    assert np.allclose([s for _, s in got], [s for _, s in want], atol=tol), (got, want)


def test_memory_and_both_blend_formulas(cuda_dev):
    """reference get_nearest_prototypes, _predict_regular and predict_batch outputs (tolerance 1e-5,
    the reference's own CPU-vs-GPU bar, tests/test_classifier.py:151-167)."""
    g = json.load(open(os.path.join(G, "memory_blend.json")))
    clf, mem, texts, Q = _golden_classifier(cuda_dev)
    for l, vals in g["protos"].items():
        assert np.allclose(mem.prototypes[l].numpy()[:8], vals, atol=1e-7)
    for i, q in enumerate(Q):
        _same_preds(mem.get_nearest_prototypes(torch.from_numpy(q), k=4), g["nearest"][i], 1e-6)
        _same_preds(mem.get_nearest_prototypes(torch.from_numpy(q), k=2), g["nearest_k2"][i], 1e-6)
    for key, want in g["predict"].items():
        k = int(key[1:])
        for t, w in zip(texts, want):
            got = clf._predict_regular(t, k)
            assert all(isinstance(s, float) for _, s in got)
            _same_preds(got, w)
    for key, want in g["predict_batch"].items():
        k = int(key[1:])
        for got, w in zip(clf.predict_batch(texts, k=k), want):
            _same_preds(got, w)


def test_multilabel_inherited_predict_paths_blend_softmax_of_sigmoid(cuda_dev):
    """The reference's inherited predict_batch / _predict_regular on a multi-label classifier softmax the head's
    SIGMOID outputs (classifier.py:1342-1345, :432-435; multilabel.py:43).  Fixture: reference outputs
    (tests/golden/multilabel_predict.json), incl. predict() falling through to super().predict."""
    from adaptive_classifier import MultiLabelAdaptiveClassifier, MultiLabelAdaptiveHead
    g = json.load(open(os.path.join(G, "multilabel_predict.json")))
    base, mem, texts, _ = _golden_classifier(cuda_dev)
    cent = synth.synth_unit_rows(4, 768, 11)
    Q = synth.synth_unit_rows(8, 768, g["q_seed"])
    Q = np.stack([(q * 0.5 + cent[i % 4]) / np.linalg.norm(q * 0.5 + cent[i % 4]) for i, q in enumerate(Q)]).astype(np.float32)
    clf = MultiLabelAdaptiveClassifier.__new__(MultiLabelAdaptiveClassifier)
    clf.__dict__.update(base.__dict__)
    torch.manual_seed(g["head_seed"])
    head = MultiLabelAdaptiveHead(768, 4, [768, 384])
    for k, v in head.state_dict().items():                              # same init as the reference's head
        flat = v.detach().reshape(-1).double().numpy()
        s = g["head_init"][k]
        assert np.abs(flat[np.asarray(s["idx"]) % flat.size] - np.asarray(s["vals"])).max() < 1e-9
    clf.adaptive_head = head.to(cuda_dev).eval()
    clf.default_threshold, clf.min_predictions, clf.max_predictions, clf.label_thresholds = 0.5, 1, None, {}
    table = {t: torch.from_numpy(q) for t, q in zip(texts, Q)}
    clf._embed_device = lambda ts, **kw: torch.stack([table[t] for t in ts]).to(cuda_dev)
    for key, want in g["predict_batch"].items():
        for got, w in zip(clf.predict_batch(texts, k=int(key[1:])), want):
            _same_preds(got, w)
    for key, want in g["predict_regular"].items():
        for t, w in zip(texts, want):
            _same_preds(clf._predict_regular(t, int(key[1:])), w)
    clf.min_predictions, clf.default_threshold = 0, 5.0                 # nothing passes -> super().predict
    for t, w in zip(texts, g["predict_fallthrough_k3"]):
        _same_preds(clf.predict(t, k=3), w)


class _StubEncoder:
    """Stands in for the HF encoder when a saved classifier is loaded offline (no weights to download)."""

    def __init__(self, hidden, name):
        from adaptive_classifier.encoder import _Cfg
        self.config = _Cfg(hidden, name)


def _queries(n, D, seed, cent_seed, C):
    Q = synth.synth_unit_rows(n, D, seed)
    cent = synth.synth_unit_rows(C, D, cent_seed)
    return np.stack([(q * 0.5 + cent[i % C]) / np.linalg.norm(q * 0.5 + cent[i % C]) for i, q in enumerate(Q)]).astype(np.float32)


def test_load_directory_written_by_the_reference(cuda_dev):
    """N1: tests/golden/ref_saved_d64 was written by the REFERENCE's _save_pretrained (classifier.py:524-628:
    config.json sorted keys, examples.json with the k-means representatives, model.safetensors).  AdaptiveClassifier.load
    restores it and predicts what the reference predicts from the same restored state (expected.json)."""
    from adaptive_classifier import AdaptiveClassifier
    d = os.path.join(G, "ref_saved_d64")
    exp = json.load(open(os.path.join(d, "expected.json")))
    D = exp["D"]
    clf = AdaptiveClassifier.load(d, device="cuda:0", encoder=_StubEncoder(D, "stub-encoder-d64"), tokenizer=None)
    assert clf.embedding_dim == D and clf.label_to_id == {"c0": 0, "c1": 1, "c2": 2, "c3": 3}
    assert clf.training_history == exp["training_history"] and clf.train_steps == 4    # 3 + the reference's training call
    assert {l: len(e) for l, e in clf.memory.examples.items()} == exp["stats"]["examples_per_class"]
    assert clf.memory.index.ntotal == 4 and clf.adaptive_head.model[-1].out_features == 4
    Q = torch.from_numpy(_queries(8, D, exp["q_seed"], exp["cent_seed"], 4)).to(cuda_dev)
    for key, want in exp["predict_batch"].items():
        for got, w in zip(clf.predict_embeddings(Q, int(key[1:])), want):
            _same_preds(got, w)
    for key, want in exp["predict"].items():
        for i, w in enumerate(want):
            S, I, P = clf._device_stage(Q[i:i + 1], len(clf.id_to_label))
            _same_preds(clf._finish(S, I, P, int(key[1:]), regular=True, b=1)[0], w)
    # and a loaded classifier keeps learning: add_embeddings on top of the restored state
    extra = _queries(6, D, 80, 51, 4)
    clf.add_embeddings([f"n{i}" for i in range(6)], [torch.from_numpy(e) for e in extra], [f"c{i % 4}" for i in range(6)])
    assert sum(len(e) for e in clf.memory.examples.values()) == sum(exp["stats"]["examples_per_class"].values()) + 6


def test_load_reference_legacy_router_layout(cuda_dev):
    """N1: the reference's shipped scripts/adaptive_router fixture (older layout: examples inline in config.json +
    tensors.safetensors; 768-d, trained head), byte-for-byte under tests/golden/adaptive_router_legacy."""
    from adaptive_classifier import AdaptiveClassifier
    d = os.path.join(G, "adaptive_router_legacy")
    exp = json.load(open(os.path.join(d, "expected.json")))
    cfg = json.load(open(os.path.join(d, "config.json")))
    clf = AdaptiveClassifier.load(d, device="cuda:0", encoder=_StubEncoder(768, cfg["model_name"]), tokenizer=None)
    assert clf.label_to_id == {"HIGH": 0, "LOW": 1} and clf.train_steps == 20
    assert clf.training_history == exp["training_history"] == {"HIGH": 100, "LOW": 100}        # :909-913 estimate
    emb = np.stack([np.asarray(e["embedding"], np.float32) for l in ("HIGH", "LOW") for e in cfg["examples"][l]])
    Q = torch.from_numpy(emb).to(cuda_dev)
    for key, want in exp["predict_batch"].items():
        for got, w in zip(clf.predict_embeddings(Q, int(key[1:])), want):
            _same_preds(got, w)
    for key, want in exp["predict"].items():
        for i, w in enumerate(want):
            S, I, P = clf._device_stage(Q[i:i + 1], 2)
            _same_preds(clf._finish(S, I, P, int(key[1:]), regular=True, b=1)[0], w)
