"""GPU suite: API-level behaviour of the drop-in AdaptiveClassifier (mirrors the reference's
tests/test_classifier.py with an offline tokenizer stub and a small random BERT) and the reference's
memory tests against the real device search."""
import os

import numpy as np
import pytest
import torch

from helpers import HashTokenizer, small_bert

pytestmark = pytest.mark.gpu

TEXTS = ["great product works well", "love it so much", "best purchase ever made",
         "terrible waste of money", "awful do not buy", "broke after one day",
         "it is fine nothing special", "average product okay", "neither good nor bad"]
LABELS = ["positive"] * 3 + ["negative"] * 3 + ["neutral"] * 3


@pytest.fixture(scope="module")
def clf(cuda_dev):
    from adaptive_classifier import AdaptiveClassifier
    from adaptive_classifier.encoder import HipBertEncoder
    enc = HipBertEncoder(small_bert(), device=cuda_dev)
    c = AdaptiveClassifier("synthetic-bert-tiny", device="cuda:0", encoder=enc, tokenizer=HashTokenizer())
    c.add_examples(TEXTS, LABELS)
    return c


def test_init_and_add(clf):
    assert clf.embedding_dim == 128 and clf.adaptive_head is not None
    assert clf.label_to_id == {"negative": 0, "neutral": 1, "positive": 2}       # sorted ids (:147-150)
    assert clf.memory.index.ntotal == 3 and clf.memory.updates_since_rebuild == 0
    assert clf.training_history == {"positive": 3, "negative": 3, "neutral": 3}
    assert clf.train_steps == 1 and clf.last_train_info["steps"] >= 1
    st = clf.get_memory_stats()
    assert st["num_classes"] == 3 and st["total_examples"] == 9
    assert clf.get_example_statistics()["model_params"] == sum(p.numel() for p in clf.adaptive_head.parameters())


def test_predict_types_and_normalisation(clf):
    preds = clf.predict("really great product", k=3)
    assert len(preds) == 3 and all(isinstance(l, str) and isinstance(s, float) for l, s in preds)
    assert abs(sum(s for _, s in preds) - 1.0) < 1e-6
    assert preds == sorted(preds, key=lambda x: -x[1])
    assert len(clf.predict("really great product", k=1)) == 1
    with pytest.raises(ValueError, match="Empty input text"):
        clf.predict("")


def test_predict_batch_and_errors(clf):
    out = clf.predict_batch(["great product", "awful thing", "fine I guess", "x"], k=2, batch_size=3)
    assert len(out) == 4 and all(len(p) <= 2 for p in out)
    assert all(isinstance(s, float) for p in out for _, s in p)
    # batch_size is a lower bound on the device batch: a text's result does not depend on what shares its batch
    texts = ["great product", "awful thing", "fine I guess", "x", "the worst", "lovely, would buy again", "meh"]
    big = clf.predict_batch(texts, k=2, batch_size=3)
    clf.config.config["min_device_batch"] = 1
    try:
        small = clf.predict_batch(texts, k=2, batch_size=3)
    finally:
        del clf.config.config["min_device_batch"]
    assert [[l for l, _ in p] for p in big] == [[l for l, _ in p] for p in small]
    for a, b in zip(big, small):
        assert all(abs(x[1] - y[1]) < 1e-5 for x, y in zip(a, b))
    with pytest.raises(ValueError, match="Empty input batch"):
        clf.predict_batch([])
    with pytest.raises(ValueError, match="Empty input lists"):
        clf.add_examples([], [])
    with pytest.raises(ValueError, match="Mismatched"):
        clf.add_examples(["a"], ["x", "y"])


def test_training_learns_separable_embeddings(cuda_dev):
    """add_embeddings (add_examples after the encoder) on 3 separable synthetic clusters: the native
    training loop (DataLoader order, dropout, CE, clip, AdamW, plateau scheduler, early stop) learns them."""
    from adaptive_classifier import AdaptiveClassifier
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import synth
    enc = HipBertEncoder(small_bert(), device=cuda_dev)
    c = AdaptiveClassifier("synthetic", device="cuda:0", encoder=enc, tokenizer=HashTokenizer())
    D, per = 128, 30
    cent = synth.synth_unit_rows(3, D, 5)
    noise = synth.synth_unit_rows(3 * per + 30, D, 6)
    def sample(i):
        v = cent[i % 3] + 0.6 * noise[i]
        return torch.from_numpy((v / np.linalg.norm(v)).astype(np.float32))
    labels = [f"cls{i % 3}" for i in range(3 * per)]
    c.add_embeddings([f"t{i}" for i in range(3 * per)], [sample(i) for i in range(3 * per)], labels)
    assert c.last_train_info["steps"] >= 3 and c.last_train_info["final_loss"] < 0.9
    test = torch.stack([sample(3 * per + i) for i in range(30)]).to(cuda_dev)
    preds = c.predict_embeddings(test, k=3)
    acc = np.mean([p[0][0] == f"cls{i % 3}" for i, p in enumerate(preds)])
    assert acc >= 0.9, acc
    assert all(abs(sum(s for _, s in p) - 1) < 1e-6 for p in preds)


def test_get_embeddings_contract(clf):
    embs = clf._get_embeddings(["hello world", "another longer piece of text here"])
    assert len(embs) == 2 and all(e.device.type == "cpu" and e.shape == (128,) for e in embs)
    assert all(abs(float(e.norm()) - 1) < 1e-5 for e in embs)
    # padding independence: same text alone or next to a longer one
    alone = clf._get_embeddings(["hello world"])[0]
    assert (alone - embs[0]).abs().max().item() < 1e-5


def test_dynamic_class_addition_and_as_wired_ewc(cuda_dev):
    from adaptive_classifier import AdaptiveClassifier
    from adaptive_classifier.encoder import HipBertEncoder
    enc = HipBertEncoder(small_bert(), device=cuda_dev)
    c = AdaptiveClassifier("synthetic", device="cuda:0", encoder=enc, tokenizer=HashTokenizer())
    c.add_examples(TEXTS[:6], LABELS[:6])
    w_before = c.adaptive_head.model[-1].weight.detach().clone()
    assert w_before.shape[0] == 2
    np.random.seed(0)
    c.add_examples(["how do i reset my password", "need help with login", "support please help"], ["support"] * 3)
    assert c.adaptive_head.model[-1].weight.shape[0] == 3                      # grown, not re-initialised
    assert c.label_to_id["support"] == 2 and c.memory.index.ntotal == 3
    assert c.train_steps == 2
    assert any(l == "support" for l, _ in c.predict("help me login", k=3))
    # intended-EWC mode trains too and reports a non-negative penalty path
    c2 = AdaptiveClassifier("synthetic", device="cuda:0", config={"ewc_mode": "intended"}, encoder=enc,
                            tokenizer=HashTokenizer())
    c2.add_examples(TEXTS[:6], LABELS[:6])
    c2.add_examples(["how do i reset my password", "need help with login", "support please help"], ["support"] * 3)
    assert c2.adaptive_head.model[-1].weight.shape[0] == 3


def test_save_load_roundtrip(clf, tmp_path, cuda_dev):
    from adaptive_classifier import AdaptiveClassifier
    clf.save(str(tmp_path))
    assert (tmp_path / "config.json").exists() and (tmp_path / "examples.json").exists() and (tmp_path / "model.safetensors").exists()
    from safetensors.torch import load_file
    keys = set(load_file(str(tmp_path / "model.safetensors")))
    assert {"prototype_positive", "adaptive_head_model.0.weight", "adaptive_head_model.6.bias"} <= keys
    c2 = AdaptiveClassifier("synthetic-bert-tiny", device="cuda:0", encoder=clf.model, tokenizer=HashTokenizer())
    c2.load_state(str(tmp_path))
    for t in ["really great product", "awful thing"]:
        a, b = clf.predict(t, k=3), c2.predict(t, k=3)
        assert [l for l, _ in a] == [l for l, _ in b]
        assert np.allclose([s for _, s in a], [s for _, s in b], atol=1e-6)


def test_from_pretrained_and_save_pretrained_on_a_local_directory(clf, tmp_path, cuda_dev):
    """The reference's usual entry point (README: AdaptiveClassifier.from_pretrained(...), ModelHubMixin): a local directory goes
    through load(); save_pretrained writes what save() writes; the reference's save(dir, include_onnx, quantize_onnx) keywords are
    accepted; push_to_hub says that it is outside this build."""
    from adaptive_classifier import AdaptiveClassifier
    out = clf.save_pretrained(tmp_path / "m")
    assert (tmp_path / "m" / "config.json").exists() and out == str(tmp_path / "m")
    clf.save(str(tmp_path / "m2"), include_onnx=True, quantize_onnx=False)
    c2 = AdaptiveClassifier.from_pretrained(str(tmp_path / "m"), device="cuda:0", encoder=clf.model, tokenizer=HashTokenizer())
    assert c2.label_to_id == clf.label_to_id
    a, b = clf.predict("really great product", k=3), c2.predict("really great product", k=3)
    assert [l for l, _ in a] == [l for l, _ in b] and np.allclose([s for _, s in a], [s for _, s in b], atol=1e-6)
    with pytest.raises(NotImplementedError, match="push_to_hub"):
        clf.push_to_hub("someone/some-model")


def test_cpu_device_rejected():
    from adaptive_classifier import AdaptiveClassifier, _native as nv
    with pytest.raises(nv.NativeError):
        AdaptiveClassifier("x", device="cpu", encoder=object(), tokenizer=None)


# ---- reference tests/test_memory.py cases that need the search ---------------------------------
def test_nearest_prototypes_reference_case(cuda_dev):
    from adaptive_classifier import Example, PrototypeMemory
    memory = PrototypeMemory(768, device=cuda_dev)
    e = torch.randn(768)
    for i, cls in enumerate(["positive", "negative", "neutral"]):
        shift = torch.zeros_like(e); shift[i] = 1.0
        memory.add_example(Example(f"text_{cls}", cls, e + shift), cls)
    memory._rebuild_index()
    q = e.clone(); q[0] = 1.0
    res = memory.get_nearest_prototypes(q, k=3)
    assert len(res) == 3 and all(isinstance(l, str) and isinstance(s, float) for l, s in res)
    assert abs(sum(s for _, s in res) - 1.0) < 1e-5
    assert memory.get_nearest_prototypes(q, k=10) == res                       # k clamps to ntotal
    assert PrototypeMemory(768, device=cuda_dev).get_nearest_prototypes(q) == []


def test_prototype_row_updates_in_place(cuda_dev):
    """Deliberate fix of memory.py:156-159: after an update the row<->label map stays valid."""
    from adaptive_classifier import Example, PrototypeMemory
    m = PrototypeMemory(8, device=cuda_dev)
    a, b = torch.zeros(8), torch.zeros(8)
    a[0], b[1] = 1, 1
    m.add_example(Example("a", "A", a), "A"); m.add_example(Example("b", "B", b), "B")
    m._rebuild_index()
    m.add_example(Example("a2", "A", a * 3), "A")                             # prototype A moves to 2*a
    res = m.get_nearest_prototypes(a * 2, k=2)
    assert res[0][0] == "A"


def test_memory_device_handling(cuda_dev):
    from adaptive_classifier import Example, PrototypeMemory
    m = PrototypeMemory(768, device=cuda_dev)
    m.add_example(Example("t", "positive", torch.randn(768, device=cuda_dev)), "positive")
    assert m.prototypes["positive"].device == torch.device("cpu")             # tests/test_memory.py:156-164


def test_generalised_store_rows_to_classes(cuda_dev):
    """M6: N >> C rows with a row->class map behind the same search (BASELINE configs[1] shape, scaled)."""
    from adaptive_classifier import PrototypeMemory
    from adaptive_classifier import index as ix
    from oracle import c_oracle
    N, D, C = 5000, 768, 4
    rows = ix.synth_unit_rows(N, D, 1, device=cuda_dev)
    m = PrototypeMemory(D, device=cuda_dev)
    m.load_rows(rows, torch.arange(N) % C, [f"c{i}" for i in range(C)])
    Q = ix.synth_unit_rows(3, D, 2, device=cuda_dev)
    S, I, Dd = m.search_batch(Q, 16)
    oD, oI = c_oracle.knn_l2_topk(rows.cpu().numpy(), Q.cpu().numpy(), 16)
    assert np.array_equal(I.cpu().numpy(), oI)
    res = m.get_nearest_prototypes(Q[0].cpu(), k=16)
    assert [l for l, _ in res] == [f"c{int(i) % C}" for i in oI[0]]


# ---- reference tests/test_order_independence.py + tests/test_ewc.py:156-191 analogues -------------
def _fresh(cuda_dev, cfg=None):
    from adaptive_classifier import AdaptiveClassifier
    from adaptive_classifier.encoder import HipBertEncoder
    enc = HipBertEncoder(small_bert(), device=cuda_dev)
    return AdaptiveClassifier("synthetic", device="cuda:0", config=cfg, encoder=enc, tokenizer=HashTokenizer())


def test_label_ids_are_order_independent(cuda_dev):
    """New classes get ids in sorted order regardless of the order examples arrive in (:147-150)."""
    a, b = _fresh(cuda_dev), _fresh(cuda_dev)
    a.add_examples(TEXTS, LABELS)
    perm = [8, 2, 5, 0, 7, 3, 1, 6, 4]
    b.add_examples([TEXTS[i] for i in perm], [LABELS[i] for i in perm])
    assert a.label_to_id == b.label_to_id == {"negative": 0, "neutral": 1, "positive": 2}
    assert a.memory.index_to_label == b.memory.index_to_label
    for l in a.memory.prototypes:
        assert torch.allclose(a.memory.prototypes[l], b.memory.prototypes[l], atol=1e-6)


def test_progressive_class_addition(cuda_dev):
    """tests/test_ewc.py:156-191: four phases, new classes twice; every label stays addressable."""
    c = _fresh(cuda_dev)
    c.add_examples(["Good product", "Bad service", "Average quality"], ["positive", "negative", "neutral"])
    c.add_examples(["Need help", "Bug report", "Feature request"], ["support", "bug", "feature"])
    c.add_examples(["Excellent!", "Terrible!", "It's okay"], ["positive", "negative", "neutral"])
    c.add_examples(["Urgent issue", "Question about pricing"], ["urgent", "inquiry"])
    expected = {"positive", "negative", "neutral", "support", "bug", "feature", "urgent", "inquiry"}
    assert set(c.label_to_id) == expected and c.adaptive_head.model[-1].out_features == 8
    assert c.memory.index.ntotal == 8 and c.train_steps == 4
    preds = c.predict("This is wonderful!", k=3)
    assert len(preds) == 3 and all(l in expected for l, _ in preds)
    many = c.predict("This is wonderful!", k=20)
    assert len(many) == 8 and abs(sum(s for _, s in many) - 1) < 1e-6


def test_many_classes_path(cuda_dev):
    """> 20 classes takes the stratified-sampling branch of _train_new_classes (:224-246)."""
    c = _fresh(cuda_dev)
    np.random.seed(42)
    base_t = [f"class {i} example {j} word{i}" for i in range(22) for j in range(3)]
    base_l = [f"class_{i:02d}" for i in range(22) for j in range(3)]
    c.add_examples(base_t, base_l)
    c.add_examples([f"brand new thing {j}" for j in range(3)], ["class_99"] * 3)
    assert len(c.label_to_id) == 23 and c.adaptive_head.model[-1].out_features == 23
    assert len(c.predict("brand new thing", k=5)) == 5


def test_continuous_learning_loop_scaled(cuda_dev):
    """BASELINE configs[3] at reduced scale: a stream of pre-computed embeddings in chunks of 32 with
    max_examples_per_class capping the store, then a 5th class (EWC path, both modes)."""
    from oracle import synth
    D, C, n = 128, 4, 640
    cent = synth.synth_unit_rows(C + 1, D, 3)
    noise = synth.synth_unit_rows(n + 64, D, 4)
    def emb(i, c):
        v = cent[c] + 0.5 * noise[i]
        return torch.from_numpy((v / np.linalg.norm(v)).astype(np.float32))
    for mode in ("as_wired", "intended"):
        c = _fresh(cuda_dev, {"max_examples_per_class": 50, "ewc_mode": mode})
        for s in range(0, n, 32):
            idx = list(range(s, s + 32))
            c.add_embeddings([f"t{i}" for i in idx], [emb(i, i % C) for i in idx], [f"c{i % C}" for i in idx])
        st = c.get_memory_stats()
        assert all(v == 50 for v in st["examples_per_class"].values()) and st["total_examples"] == 200
        assert c.training_history == {f"c{k}": n // C for k in range(C)}
        np.random.seed(1)
        c.add_embeddings([f"n{i}" for i in range(32)], [emb(n + i, C) for i in range(32)], ["zzz_new"] * 32)
        assert c.adaptive_head.model[-1].out_features == 5
        test = torch.stack([emb(n + 32 + i, i % (C + 1)) for i in range(30)]).to(cuda_dev)
        want = [f"c{i % (C + 1)}" if i % (C + 1) < C else "zzz_new" for i in range(30)]
        acc = np.mean([p[0][0] == w for p, w in zip(c.predict_embeddings(test, k=2), want)])
        assert acc >= 0.8, (mode, acc)


def test_load_classmethod_and_legacy_layout(clf, tmp_path, cuda_dev):
    """N1: directories in the reference's current layout (config.json + examples.json + model.safetensors)
    and in its older one (examples inline + tensors.safetensors, as scripts/adaptive_router) both load."""
    import json
    import shutil
    from adaptive_classifier import AdaptiveClassifier
    cur = tmp_path / "cur"
    clf.save(str(cur))
    cfg = json.loads((cur / "config.json").read_text())
    assert cfg["library_name"] == "adaptive-classifier" and cfg["embedding_dim"] == 128
    a = AdaptiveClassifier.load(str(cur), device="cuda:0", encoder=clf.model, tokenizer=HashTokenizer())
    old = tmp_path / "old"
    old.mkdir()
    cfg["examples"] = json.loads((cur / "examples.json").read_text())
    cfg.pop("training_history")
    (old / "config.json").write_text(json.dumps(cfg))
    shutil.copy(cur / "model.safetensors", old / "tensors.safetensors")
    b = AdaptiveClassifier.load(str(old), device="cuda:0", encoder=clf.model, tokenizer=HashTokenizer())
    assert b.training_history == {l: 20 * 3 for l in ("positive", "negative", "neutral")}      # :909-913 estimate
    for t in ["really great product", "awful thing"]:
        want = clf.predict_batch([t], k=3)[0]
        for other in (a, b):
            got = other.predict_batch([t], k=3)[0]        # predict_batch weights do not depend on history
            assert [l for l, _ in got] == [l for l, _ in want]
            assert np.allclose([s for _, s in got], [s for _, s in want], atol=1e-6)


def test_history_and_confidence_survive_save_load(tmp_path, cuda_dev):
    """tests/test_confidence_consistency.py:9-86, tests/test_single_example_confidence.py:7-54: the confidences of a
    restored classifier equal the saved one's (prototypes, head and training_history travel), examples.json holds the
    k-means representatives (<= num_representative_examples per class), and learning continues cumulatively."""
    import json
    from adaptive_classifier import AdaptiveClassifier
    c = _fresh(cuda_dev)
    np.random.seed(0)
    c.add_examples(["This is a foo example number %d" % (i % 7) for i in range(100)] +
                   ["This is a bar example number %d" % (i % 7) for i in range(100)], ["foo"] * 100 + ["bar"] * 100)
    before = dict(c.predict("This is a foo example"))
    c.save(str(tmp_path / "m"))
    saved = json.loads((tmp_path / "m" / "examples.json").read_text())
    assert {l: len(v) for l, v in saved.items()} == {"foo": 5, "bar": 5}
    r = AdaptiveClassifier.load(str(tmp_path / "m"), device="cuda:0", encoder=c.model, tokenizer=HashTokenizer())
    assert r.training_history == {"foo": 100, "bar": 100} and r.train_steps == c.train_steps
    assert {l: len(v) for l, v in r.memory.examples.items()} == {"foo": 5, "bar": 5}
    after = dict(r.predict("This is a foo example"))
    assert abs(before["foo"] - after["foo"]) < 1e-6 and abs(before["bar"] - after["bar"]) < 1e-6
    assert max(before, key=before.get) == "foo"
    np.random.seed(1)
    r.add_examples(["Additional foo example"] * 20 + ["Additional bar example"] * 20, ["foo"] * 20 + ["bar"] * 20)
    assert r.training_history == {"foo": 120, "bar": 120}
    assert set(dict(r.predict("This is a foo example"))) == {"foo", "bar"}
    # one example per class: the estimate path must not kick in when the history was saved (:909-913)
    s1 = _fresh(cuda_dev)
    s1.add_examples(["fish swim in water", "cats say meow"], ["foo", "bar"])
    b1 = dict(s1.predict("fish swim"))
    s1.save(str(tmp_path / "s"))
    r1 = AdaptiveClassifier.load(str(tmp_path / "s"), device="cuda:0", encoder=s1.model, tokenizer=HashTokenizer())
    assert r1.training_history == {"foo": 1, "bar": 1}
    a1 = dict(r1.predict("fish swim"))
    assert all(abs(b1[l] - a1[l]) < 1e-6 for l in b1)


@pytest.mark.parametrize("C,kp,k,regular", [(4, 4, 3, False), (4, 4, 3, True), (3, 16, 5, False), (37, 16, 5, False),
                                            (37, 37, 37, True), (300, 24, 7, False), (2048, 8, 4, True)])
def test_device_blend_equals_numpy_formula(C, kp, k, regular, cuda_dev):
    """ac_blend_topk (device, fp64) against AdaptiveClassifier._blend (the numpy statement of classifier.py:447-480
    / :1359-1384 pinned on the reference's outputs in test_host_logic): same labels in the same order, scores to
    1e-12, on random inputs with several hits per class, padding ids, ties and exact duplicates."""
    from adaptive_classifier import AdaptiveClassifier
    from adaptive_classifier.encoder import HipBertEncoder
    enc = HipBertEncoder(small_bert(), device=cuda_dev)
    c = AdaptiveClassifier("synthetic-bert-tiny", device="cuda:0", encoder=enc, tokenizer=HashTokenizer())
    c.id_to_label = {i: f"class{i:04d}" for i in range(C)}
    c.label_to_id = {v: i for i, v in c.id_to_label.items()}
    rng = np.random.default_rng(C * 1000 + kp)
    c.training_history = {c.id_to_label[i]: int(rng.integers(0, 20)) for i in range(C)}
    b = 33
    S = rng.random((b, kp)).astype(np.float32)
    S[:, 1:] = np.minimum.accumulate(S, axis=1)[:, 1:]                  # descending like real hits
    Cid = rng.integers(-1, C, (b, kp)).astype(np.int64)                  # -1 = padding; repeats = several rows per class
    P = rng.random((b, C)).astype(np.float32)
    P /= P.sum(1, keepdims=True)
    P[3, :] = 1.0 / C                                                    # all head probabilities tied
    S[5, :] = 0.25                                                       # all hit scores tied
    if C > 2:
        P[7, 1] = P[7, 2]                                                # an exact duplicate pair
    for use_s, use_p in ((True, True), (True, False), (False, True)):
        Sn, Cn, Pn = (S if use_s else None), (Cid if use_s else None), (P if use_p else None)
        want = c._blend(Sn, Cn, Pn, k, regular)
        dev = lambda a: None if a is None else torch.from_numpy(a).to(cuda_dev)
        got = c._finish(dev(Sn), dev(Cn), dev(Pn), k, regular, b=b)
        assert len(got) == b
        for q, (g, w) in enumerate(zip(got, want)):
            assert [l for l, _ in g] == [l for l, _ in w], (q, use_s, use_p, g, w)
            assert np.allclose([s for _, s in g], [s for _, s in w], rtol=0, atol=1e-12), (q, g, w)


@pytest.mark.parametrize("cap,D,stream", [(50, 128, 400), (10, 64, 90), (1000, 768, 1200)])
def test_batched_device_prune_equals_per_example_host_logic(cap, D, stream, cuda_dev):
    """PrototypeMemory.add_examples_batch (ac_memory_add_prune: the add -> mean -> drop-farthest loop per class on
    the device) leaves exactly the state the per-example add_example loop (the statement of memory.py:41-83 /
    :196-217 checked step by step against the reference algorithm in test_host_logic) leaves: same examples in
    the same order per class, same prototypes, same counters."""
    from adaptive_classifier.memory import PrototypeMemory
    from adaptive_classifier.models import Example, ModelConfig
    cfg = ModelConfig({"max_examples_per_class": cap, "prototype_update_frequency": 37})
    a = PrototypeMemory(D, cfg, device=cuda_dev)
    b = PrototypeMemory(D, cfg, device=cuda_dev)
    g = torch.Generator().manual_seed(cap + D)
    labels = ["x", "y", "z"]
    cents = torch.randn(3, D, generator=g)
    i = 0
    for chunk in (7, 32, 32, 1, 64, 32, 200, 32):              # ragged call sizes, crossing the cap mid-call
        if i >= stream:
            break
        exs, labs = [], []
        for _ in range(chunk):
            c = int(torch.randint(0, 3, (1,), generator=g))
            v = torch.nn.functional.normalize(cents[c] + 0.5 * torch.randn(D, generator=g), dim=0)
            exs.append(v); labs.append(labels[c]); i += 1
        for v, l in zip(exs, labs):
            a.add_example(Example(f"t{id(v)}", l, v.clone()), l)
        b.add_examples_batch([Example(f"t{id(v)}", l, v.clone()) for v, l in zip(exs, labs)], labs)
        for l in labels:
            assert [e.text for e in a.examples[l]] == [e.text for e in b.examples[l]], (l, i)
            if l in a.prototypes:
                assert torch.allclose(a.prototypes[l], b.prototypes[l], atol=1e-6, rtol=0)
        assert a.updates_since_rebuild == b.updates_since_rebuild
        assert a.label_to_index == b.label_to_index
    assert max(len(b.examples[l]) for l in labels) <= cap
    # and the searchable state agrees
    q = torch.nn.functional.normalize(torch.randn(D, generator=g), dim=0)
    a._rebuild_index(); b._rebuild_index()
    ra, rb = a.get_nearest_prototypes(q, 3), b.get_nearest_prototypes(q, 3)
    assert [l for l, _ in ra] == [l for l, _ in rb] and np.allclose([s for _, s in ra], [s for _, s in rb], atol=1e-6)



def test_device_class_matrix_cache_hits_with_the_default_device_string(cuda_dev):
    """PrototypeMemory(device="cuda") (the classifier's default string): the device-resident class matrix must be REUSED by
    the next add_examples_batch call, not re-stacked and re-uploaded (tensors report cuda:0, the string says cuda)."""
    from adaptive_classifier.memory import PrototypeMemory
    from adaptive_classifier.models import Example, ModelConfig
    m = PrototypeMemory(32, ModelConfig({"max_examples_per_class": 1000}), device="cuda")
    g = torch.Generator().manual_seed(5)
    mk = lambda n: [Example(f"t{i}", "a", torch.randn(32, generator=g)) for i in range(n)]
    m.add_examples_batch(mk(5), ["a"] * 5)
    first = m._dmats["a"][0]
    ptr = first.data_ptr()
    m.add_examples_batch(mk(7), ["a"] * 7)
    assert m._dmats["a"][0].data_ptr() == ptr and m._dmats["a"][1] == 12          # same storage, appended in place
    want = torch.stack([e.embedding for e in m.examples["a"]])
    assert torch.equal(m._dmats["a"][0][:12].cpu(), want)
    # an outside edit of the list (identity changes) invalidates it
    m.examples["a"][3] = Example("swap", "a", torch.randn(32, generator=g))
    ent = m.device_class_matrix("a", "cuda")
    assert torch.equal(ent[0][:12].cpu(), torch.stack([e.embedding for e in m.examples["a"]]))


def test_void_persistent_launches_are_repeated_launch_by_launch(cuda_dev, monkeypatch, caplog):
    """The one-launch kernels poison their outputs with NaNs when a grid barrier gives up (a device shared with another
    compute process) instead of hanging; the host notices and repeats the work through the ordinary launches.  Simulated
    here by corrupting what the persistent launches return: training and predict() must come out finite and equal to a
    run that never used them."""
    import logging
    from adaptive_classifier import AdaptiveClassifier, _native as nv
    from adaptive_classifier.encoder import HipBertEncoder
    from adaptive_classifier.training import HeadTrainer
    enc = HipBertEncoder(small_bert(), device=cuda_dev)

    def build():
        c = AdaptiveClassifier("synthetic-bert-tiny", device="cuda:0", encoder=enc, tokenizer=HashTokenizer())
        c.add_examples(TEXTS, LABELS)
        return c

    prev = nv.lib().ac_set_persistent_kernels(0)          # reference run: launch-by-launch everywhere
    try:
        want_clf = build()
        want = want_clf.predict("great product", k=3)
    finally:
        nv.lib().ac_set_persistent_kernels(prev)

    real_epoch = HeadTrainer.fused_epoch
    calls = {"n": 0}

    def void_epoch(self, *a, **k):
        done = real_epoch(self, *a, **k)
        if (nv.lib().ac_set_persistent_kernels(-1) & 1) and not k.get("stepwise"):
            # a "void" persistent epoch: NaN loss, the output layer half-written, the moments half-written too (a barrier
            # that gives up at the last step lets some workgroups write their slices back)
            calls["n"] += 1
            self.loss_accum.fill_(float("nan"))
            self.flat[-16:] += 1.0
            self.m[: self.m.numel() // 2] += 0.5
            self.v[: self.v.numel() // 3] += 0.25
        return done

    monkeypatch.setattr(HeadTrainer, "fused_epoch", void_epoch)
    real_encode = enc.encode_cls

    def void_encode(ids, *a, **k):
        out = real_encode(ids, *a, **k)
        if ids.shape[0] * ids.shape[1] <= 32 and (nv.lib().ac_set_persistent_kernels(-1) & 2) and not k.get("force_layered"):
            calls["n"] += 1
            out = out * float("nan")
        return out

    with caplog.at_level(logging.WARNING):
        got_clf = build()
        monkeypatch.setattr(enc, "encode_cls", void_encode)
        got = got_clf.predict("great product", k=3)
    assert calls["n"] >= 2 and "persistent kernel" in caplog.text
    assert nv.lib().ac_set_persistent_kernels(-1) == prev                      # the process-wide switch was never touched
    assert np.isfinite(got_clf.last_train_info["final_loss"])
    assert torch.allclose(got_clf.adaptive_head.flat_params(), want_clf.adaptive_head.flat_params(), atol=1e-6)
    assert [l for l, _ in got] == [l for l, _ in want] and np.allclose([s for _, s in got], [s for _, s in want], atol=1e-6)


def test_gave_up_layernorm_exchange_is_reported_loudly(clf, monkeypatch):
    """The fused-LayerNorm GEMM epilogues poison their rows with NaN when the tiles of a row panel do not all arrive (a
    device that cannot hold one workgroup per CU at once); predict paths must then raise, naming the cause, and switch the
    fusion off for that encoder (a per-object option since round 5) -- never hand out NaN scores silently.  Simulated: NaN
    embeddings + the encoder's verdict."""
    from adaptive_classifier import _native as nv
    emb = clf._embed_device(["great product", "awful thing"])
    monkeypatch.setattr(clf.model, "ln_fusion_aborted", lambda: True, raising=False)
    before = nv.lib().ac_gemm_ln_fusion_launches()
    try:
        with pytest.raises(nv.NativeError, match="fused LayerNorm"):
            clf.predict_embeddings(emb * float("nan"), k=2)
        # the switch is off now: an encoder call at a fusable shape launches no fused GEMM
        clf.model.encode_cls(torch.randint(1000, 2000, (24, 16)), None, torch.ones(24, 16, dtype=torch.int64))
        assert nv.lib().ac_gemm_ln_fusion_launches() == before
        assert clf.model.ln_fusion is False                  # ... for THIS encoder; no process-wide switch was touched
    finally:
        clf.model.ln_fusion = None


def _long_texts(n, words=14):
    vocab = ["alpha", "bravo", "charlie", "delta", "echo", "foxtrot", "golf", "hotel", "india", "juliet", "kilo", "lima"]
    return [" ".join(vocab[(i * 7 + j * 3) % len(vocab)] for j in range(words)) for i in range(n)]


def test_starved_layernorm_exchange_never_reaches_memory_or_caller(cuda_dev, caplog):
    """With the fused-LayerNorm exchange starved for real (ac_gemm_set_ln_fusion(2)) on shapes where the fusion applies:
    add_examples() stores finite prototypes (the encoder repairs its own call), predict_batch() and predict_tokens() return
    finite scores WITHOUT raising (transparent repeat), and the results equal a run that never fused."""
    import logging
    from adaptive_classifier import AdaptiveClassifier, _native as nv
    from adaptive_classifier.encoder import HipBertEncoder
    enc = HipBertEncoder(small_bert(hidden=128, layers=3), device=cuda_dev)
    texts = _long_texts(24)                                # 24 x 16 tokens = 384 rows: fused epilogues run
    labels = [("a", "b", "c")[i % 3] for i in range(24)]
    lib = nv.lib()

    def build():
        c = AdaptiveClassifier("synthetic-bert-tiny", device="cuda:0", encoder=enc, tokenizer=HashTokenizer())
        c.add_examples(texts, labels)
        return c
    try:
        nv.check(lib.ac_gemm_set_ln_fusion(0), "ac_gemm_set_ln_fusion")
        want_clf = build()
        want = want_clf.predict_batch(texts, k=3, batch_size=24)
        nv.check(lib.ac_gemm_set_ln_fusion(2), "ac_gemm_set_ln_fusion")
        n0 = enc.ln_gave_up
        with caplog.at_level(logging.WARNING):
            got_clf = build()                              # add_examples under a starved exchange
        assert enc.ln_gave_up == n0 + 1 and "LayerNorm" in caplog.text
        for l, p in got_clf.memory.prototypes.items():
            assert torch.isfinite(p).all() and torch.allclose(p, want_clf.memory.prototypes[l], atol=1e-5)
        assert all(torch.isfinite(e.embedding).all() for exs in got_clf.memory.examples.values() for e in exs)
        # predict paths: verify=False inside, NaN scores noticed at the result copy, batch repeated, no exception
        assert enc.ln_fusion is False                      # the encoder switched ITS fusion off (per object) ...
        for call in (lambda: got_clf.predict_batch(texts, k=3, batch_size=24),
                     lambda: got_clf.predict_tokens(**HashTokenizer()(texts), k=3)):
            enc.ln_fusion = None                           # ... re-armed here: back to the (starved) process-wide test hook
            nv.check(lib.ac_gemm_set_ln_fusion(2), "ac_gemm_set_ln_fusion")
            caplog.clear()
            with caplog.at_level(logging.WARNING):
                got = call()
            assert "encoded again" in caplog.text
            assert all(s == s for p in got for _, s in p)
            for g, w in zip(got, want):
                assert [l for l, _ in g] == [l for l, _ in w] and np.allclose([s for _, s in g], [s for _, s in w], atol=1e-4)
    finally:
        nv.check(lib.ac_gemm_set_ln_fusion(1), "ac_gemm_set_ln_fusion")


def test_non_finite_embeddings_are_refused_before_anything_is_stored(clf):
    """add_embeddings() with a NaN / inf row: ValueError, and neither the memory, the label maps nor the head have changed."""
    before = (dict(clf.label_to_id), dict(clf.id_to_label), {l: len(v) for l, v in clf.memory.examples.items()},
              {l: p.clone() for l, p in clf.memory.prototypes.items()}, clf.train_steps)
    good = clf._get_embeddings(["fine text", "more text"])
    for poison in (float("nan"), float("inf")):
        bad = good[1].clone(); bad[3] = poison
        with pytest.raises(ValueError, match="non-finite embedding"):
            clf.add_embeddings(["fine text", "more text"], [good[0], bad], ["positive", "brand_new_label"])
        assert (dict(clf.label_to_id), dict(clf.id_to_label)) == before[:2] and clf.train_steps == before[4]
        assert {l: len(v) for l, v in clf.memory.examples.items()} == before[2]
        assert all(torch.equal(p, before[3][l]) for l, p in clf.memory.prototypes.items())

    class NanEncoder:                                      # a user-supplied encoder that fails: add_examples raises, stores nothing
        config = clf.model.config
        def encode_cls(self, ids, types=None, mask=None):
            return torch.full((ids.shape[0], clf.embedding_dim), float("nan"), device="cuda:0")
    real = clf.model
    try:
        clf.model = NanEncoder()
        from adaptive_classifier import _native as nv
        with pytest.raises(nv.NativeError, match="non-finite embeddings"):
            clf.add_examples(["some text"], ["yet_another_label"])
        assert dict(clf.label_to_id) == before[0]
    finally:
        clf.model = real


CU_MASK_SCRIPT = r'''
import json, logging, os, sys
sys.path[:0] = [%(root)r, %(pkg)r]
import ctypes
import numpy as np
import torch
from adaptive_classifier import AdaptiveClassifier, _native as nv
from adaptive_classifier.encoder import HipBertEncoder
from oracle import bert_oracle
records = []
class H(logging.Handler):
    def emit(self, r): records.append(r.getMessage())
logging.getLogger().addHandler(H()); logging.getLogger().setLevel(logging.WARNING)
L = nv.lib()
chip, act = ctypes.c_int(0), ctypes.c_int(0)
nv.check(L.ac_device_cus(ctypes.byref(chip), ctypes.byref(act)), "ac_device_cus")
out = {"chip": chip.value, "active": act.value}
dev = torch.device("cuda:0")
model = bert_oracle.make_bert(768, 3, 12, 3072, vocab=2000, seed=3)
enc = HipBertEncoder(model, device=dev)
ids, types, mask = bert_oracle.synthetic_batch(200, 32, vocab=2000, seed=99, ragged=True)     # ~4000 rows: 32 panels x 6 tiles = 192 tiles
n0 = L.ac_gemm_ln_fusion_launches()
got = enc.encode_cls(ids, types, mask).cpu()
out["ln_fused_launches"] = int(L.ac_gemm_ln_fusion_launches() - n0)
out["ln_gave_up"] = int(enc.ln_gave_up)
out["enc_err"] = float((got - bert_oracle.encode_cls(model, ids, types, mask)).abs().max())
he, bs = ctypes.c_int64(0), ctypes.c_int64(0)
q1 = enc.encode_cls(ids[:1, :12], types[:1, :12], torch.ones_like(mask[:1, :12])).cpu()
nv.check(L.ac_persistent_launches(ctypes.byref(he), ctypes.byref(bs)), "ac_persistent_launches")
out["bert_small_launches"] = int(bs.value); out["one_launch"] = bool(enc.last_one_launch)
out["q1_err"] = float((q1 - bert_oracle.encode_cls(model, ids[:1, :12], types[:1, :12], torch.ones_like(mask[:1, :12]))).abs().max())
clf = AdaptiveClassifier("synthetic", device="cuda:0", encoder=enc)
rng = np.random.default_rng(0)
emb = rng.standard_normal((256, 768)).astype(np.float32); emb /= np.linalg.norm(emb, axis=1, keepdims=True)
clf.add_embeddings([f"t{i}" for i in range(256)], torch.from_numpy(emb), [f"c{i %% 4}" for i in range(256)])
nv.check(L.ac_persistent_launches(ctypes.byref(he), ctypes.byref(bs)), "ac_persistent_launches")
out["head_epoch_launches"] = int(he.value)
out["final_loss"] = float(clf.last_train_info["final_loss"])
# every kNN path on the CUs this process reaches (their grids are sized from the same count): ids against each other and the oracle
from adaptive_classifier import index as ix
from oracle import c_oracle
P = ix.synth_unit_rows(100_000, 128, 1, device=dev)
prep = ix.prepare_store(P, 100_000, 128)
ids_ok = True
for nq in (3, 40, 300, 4100):
    Q = ix.synth_unit_rows(nq, 128, 2, device=dev)
    _, i0 = ix.knn_l2_topk(P, 100_000, 128, Q, 10)                      # fp32 sweep (ring / plain)
    _, i1 = ix.knn_l2_topk(P, 100_000, 128, Q, 10, prepared=prep)       # fp16 plane sweep (< 64 queries) / GEMM-form batch sweep
    _, oi = c_oracle.knn_l2_topk(P.cpu().numpy()[:, :128], Q[:8].cpu().numpy()[:, :128], 10)
    ids_ok = ids_ok and bool(torch.equal(i0, i1)) and bool((i0[:8].cpu().numpy() == oi).all())
out["knn_ids_ok"] = ids_ok
out["warnings"] = [m for m in records if "gave up" in m or "NaN" in m or "repeating" in m]
print("RESULT " + json.dumps(out))
'''


@pytest.mark.parametrize("cu_mask", [None, "0:0-127", "0:0-118"])
def test_residency_is_proven_at_launch_not_discovered_by_a_wait(cuda_dev, cu_mask):
    """VERDICT r05 item 5.  The library MEASURES how many CUs its workgroups reach (ac_device_cus: a probe launch) and makes every
    co-residency decision against that count.  Unmasked: all the chip's CUs are seen, the fused LayerNorm epilogues, the one-launch
    encoder and the persistent training epoch are selected and nothing gives up.  Under HSA_CU_MASK (half the CUs) the forms whose
    workgroups could not all be resident are NOT SELECTED -- no bounded wait expires, no NaN row is produced, no call is repeated
    (the warnings the retry paths log stay absent) -- and the results are the same numbers."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("HSA_CU_MASK", None)
    if cu_mask:
        env["HSA_CU_MASK"] = cu_mask
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = CU_MASK_SCRIPT % {"root": root, "pkg": os.path.join(root, "adaptive-classifier_amd")}
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    print(cu_mask, out)
    assert out["enc_err"] < 1e-4 and out["q1_err"] < 1e-4 and out["final_loss"] == out["final_loss"]
    assert out["ln_gave_up"] == 0 and out["warnings"] == [] and out["knn_ids_ok"], out
    if cu_mask is None:
        assert out["active"] == out["chip"] >= 64
        assert out["ln_fused_launches"] == 4 and out["one_launch"] and out["bert_small_launches"] == 1 and out["head_epoch_launches"] >= 1
    else:
        if out["active"] == out["chip"]:
            pytest.skip("HSA_CU_MASK has no effect on this box's runtime (active CUs == chip CUs)")
        assert out["active"] < out["chip"]
        # 192 LayerNorm tiles / 192 one-launch workgroups do not fit the masked CUs: the unfused forms run, chosen up front
        assert out["ln_fused_launches"] == 0 and not out["one_launch"] and out["bert_small_launches"] == 0


@pytest.mark.parametrize("C,kp,k,regular", [(4, 16, 16, False), (4, 4, 3, True), (37, 16, 5, False), (300, 24, 7, False),
                                            (2048, 8, 4, True), (5, 1024, 5, False), (70, 1, 70, True)])
def test_predict_post_equals_the_four_kernels_it_replaces(C, kp, k, regular, cuda_dev):
    """ac_predict_post (prototype scores + hit classes + F.softmax + blend + top-k in one launch, the packed result written
    into host-mapped memory and waited for on a completion flag) against ac_proto_scores -> ac_rows_to_class ->
    ac_softmax_rows -> ac_blend_topk -> D2H: the same packed bytes, over random distances with padding ids (-1), ids beyond
    the store, labels unknown to the classifier, a row -> class map with several rows per class, tied logits; with and
    without hits / head; without a host buffer (asynchronous form) the device buffer holds the same bytes."""
    import ctypes
    from adaptive_classifier import _native as nv
    from adaptive_classifier.ops import softmax_rows
    from adaptive_classifier.index import proto_scores
    rng = np.random.default_rng(C * 7 + kp)
    b, nrows, nlut = 41, 5000, C + 3
    D = np.sort(rng.random((b, kp)).astype(np.float32) * 2, axis=1)
    I = rng.integers(-1, nrows + 40, (b, kp)).astype(np.int64)
    I[2, :] = -1
    row_class = rng.integers(0, nlut, nrows).astype(np.int32)
    lut = np.concatenate([rng.permutation(C), [-1, -1, -1]]).astype(np.int64)[rng.permutation(nlut)]
    Z = (rng.standard_normal((b, C)) * 3).astype(np.float32)
    Z[3, :] = 0.5
    wts = torch.tensor(rng.choice([0.3, 0.7], size=(2, C)), dtype=torch.float64, device=cuda_dev)
    dev = lambda a: torch.from_numpy(a).to(cuda_dev)
    Dd, Id, rc, lt, Zd = dev(D), dev(I), dev(row_class), dev(lut), dev(Z)
    kk = max(1, min(k, C)); ncls = C if regular else min(k, C)
    off_cls = 4 * b; off_val = (off_cls + 4 * b * kk + 7) // 8 * 8; need = (off_val + 8 * b * kk + 15) // 16 * 16
    lib = nv.lib()
    sp = nv.stream_ptr(cuda_dev)
    hp = ctypes.c_void_p()
    nv.check(lib.ac_host_alloc(need + 64, ctypes.byref(hp)), "ac_host_alloc")
    host = np.frombuffer((ctypes.c_ubyte * (need + 64)).from_address(hp.value), dtype=np.uint8)
    try:
        for use_hits, use_head in ((True, True), (True, False), (False, True)):
            # the separate kernels
            S = Cid = P = None
            if use_hits:
                S = proto_scores(Dd, Id)
                Cid = torch.empty_like(Id)
                nv.check(lib.ac_rows_to_class(nv.ptr(Id), Id.numel(), nv.ptr(rc), nrows, nv.ptr(lt), nlut, nv.ptr(Cid), sp), "rows_to_class")
            if use_head:
                P = softmax_rows(Zd)
            want = torch.zeros(need, dtype=torch.uint8, device=cuda_dev)
            base = want.data_ptr()
            nv.check(lib.ac_blend_topk(nv.ptr(S), nv.ptr(Cid), kp if use_hits else 0, nv.ptr(P), C, wts[0].data_ptr(), wts[1].data_ptr(), ncls, kk, b,
                                       base, base + off_cls, base + off_val, sp), "blend")
            want = want.cpu().numpy()
            n = want[:off_cls].view(np.int32)
            for wait_host in (1, 0):
                host[:] = 0xAB
                got_d = torch.zeros(need, dtype=torch.uint8, device=cuda_dev)
                nv.check(lib.ac_predict_post(nv.ptr(Dd) if use_hits else None, nv.ptr(Id) if use_hits else None, kp, nv.ptr(rc), nrows, nv.ptr(lt), nlut,
                                             nv.ptr(Zd) if use_head else None, C, 1, wts[0].data_ptr(), wts[1].data_ptr(), ncls, kk, b, nv.ptr(got_d), need,
                                             hp if wait_host else None, sp), "ac_predict_post")
                got = host[:need].copy() if wait_host else got_d.cpu().numpy()      # (h_out given: no synchronisation by the caller)
                assert np.array_equal(got[:off_cls], want[:off_cls]), (use_hits, use_head, wait_host, got[:off_cls].view(np.int32), n)
                gc, wc = got[off_cls:off_cls + 4 * b * kk].view(np.int32).reshape(b, kk), want[off_cls:off_cls + 4 * b * kk].view(np.int32).reshape(b, kk)
                gv, wv = (a[off_val:off_val + 8 * b * kk].view(np.float64).reshape(b, kk) for a in (got, want))
                for q in range(b):               # (entries beyond n[q] are not written by either form)
                    assert np.array_equal(gc[q, :n[q]], wc[q, :n[q]]) and np.array_equal(gv[q, :n[q]], wv[q, :n[q]]), (q, use_hits, use_head, wait_host)
        assert lib.ac_predict_post(nv.ptr(Dd), nv.ptr(Id), 1025, None, 0, None, 0, None, C, 1, wts[0].data_ptr(), wts[1].data_ptr(), ncls, kk, b, nv.ptr(Dd), need, hp, sp) == -2
        assert lib.ac_predict_post(nv.ptr(Dd), nv.ptr(Id), kp, None, 0, None, 0, None, C, 1, wts[0].data_ptr(), wts[1].data_ptr(), ncls, kk, b, nv.ptr(Dd), need - 16, hp, sp) == -3
    finally:
        nv.check(lib.ac_host_free(hp), "ac_host_free")


def test_predictions_do_not_depend_on_the_post_kernel(clf, monkeypatch):
    """predict / predict_batch / predict_embeddings through ac_predict_post (default) and through the separate kernels
    (AC_PREDICT_POST=0): identical lists."""
    texts = TEXTS + ["something else entirely", "good"]
    monkeypatch.setenv("AC_PREDICT_POST", "0")
    a_batch = clf.predict_batch(texts, k=3)
    a_one = [clf.predict(t, k=2) for t in texts[:4]]
    monkeypatch.setenv("AC_PREDICT_POST", "1")
    assert clf.predict_batch(texts, k=3) == a_batch
    assert [clf.predict(t, k=2) for t in texts[:4]] == a_one
    emb = clf._embed_device(texts)
    assert clf.predict_embeddings(emb, k=3) == a_batch
