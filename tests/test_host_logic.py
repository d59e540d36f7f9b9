"""CPU suite: host-side logic of the drop-in classes, mirroring the model-free tests of the
reference (tests/test_memory.py, tests/test_ewc.py:34-84,128-153,194-215) that do not need a search."""
import threading

import pytest
import torch
import torch.nn as nn

from adaptive_classifier import EWC, AdaptiveHead, Example, ModelConfig, PrototypeMemory


@pytest.fixture
def memory():
    return PrototypeMemory(embedding_dim=768)


@pytest.fixture
def emb():
    return torch.randn(768)


@pytest.fixture
def config():
    return ModelConfig({"max_examples_per_class": 5, "prototype_update_frequency": 3, "similarity_threshold": 0.95})


def test_initialization(memory):
    assert memory.embedding_dim == 768 and len(memory.examples) == 0 and len(memory.prototypes) == 0
    assert memory.index is not None


def test_add_example_and_counter(memory, emb):
    memory.add_example(Example("test text", "positive", emb), "positive")
    assert len(memory.examples["positive"]) == 1 and "positive" in memory.prototypes
    assert memory.updates_since_rebuild == 1


def test_prototype_is_mean(memory, emb):
    exs = [Example(f"text_{i}", "positive", emb + i) for i in range(3)]
    for ex in exs:
        memory.add_example(ex, "positive")
    assert torch.allclose(torch.stack([e.embedding for e in exs]).mean(0), memory.prototypes["positive"])


def test_pruning(emb, config):
    m = PrototypeMemory(768, config=config)
    for i in range(config.max_examples_per_class + 3):
        m.add_example(Example(f"text_{i}", "positive", emb + i), "positive")
    assert len(m.examples["positive"]) == config.max_examples_per_class
    # kept = the ones closest to the running mean; prototype == mean of what is kept
    kept = torch.stack([e.embedding for e in m.examples["positive"]])
    assert torch.allclose(kept.mean(0), m.prototypes["positive"], atol=1e-5)


def test_index_rebuild_counter(emb, config):
    m = PrototypeMemory(768, config=config)
    for i in range(config.prototype_update_frequency + 1):
        m.add_example(Example(f"text_{i}", "positive", emb + i), "positive")
    assert m.updates_since_rebuild == 0 and len(m.label_to_index) == 1 and len(m.index_to_label) == 1
    assert m.index.ntotal == 1


def test_rebuild_sorted_label_rows(memory, emb):
    for lab in ["zeta", "alpha", "mid"]:
        memory.add_example(Example("t", lab, emb), lab)
    memory._rebuild_index()
    assert memory.index_to_label == {0: "alpha", 1: "mid", 2: "zeta"} and memory.index.ntotal == 3


def test_clear_and_stats(memory, emb):
    for cls in ["positive", "negative"]:
        for i in range(3):
            memory.add_example(Example(f"text_{cls}_{i}", cls, emb + i), cls)
    st = memory.get_stats()
    assert st["num_classes"] == 2 and st["examples_per_class"]["positive"] == 3 and st["total_examples"] == 6
    assert st["prototype_dimensions"] == 768 and "updates_since_rebuild" in st
    memory.clear()
    assert not memory.examples and not memory.prototypes and not memory.label_to_index and memory.updates_since_rebuild == 0


def test_invalid_inputs(memory):
    with pytest.raises(ValueError):
        memory.add_example(Example("t", "positive", torch.randn(100)), "positive")
    with pytest.raises(ValueError):
        memory.add_example(Example("t", "positive", None), "positive")


def test_prototype_stability(memory, emb):
    ex = Example("test text", "positive", emb)
    for _ in range(5):
        memory.add_example(ex, "positive")
    assert torch.allclose(memory.prototypes["positive"], emb, atol=1e-6)


def test_concurrent_access(memory, emb):
    def add(label):
        for i in range(100):
            memory.add_example(Example(f"text_{label}_{i}", label, emb + i), label)
    ts = [threading.Thread(target=add, args=(l,)) for l in ["positive", "negative", "neutral"]]
    [t.start() for t in ts]
    [t.join() for t in ts]
    st = memory.get_stats()
    assert st["num_classes"] == 3 and all(len(memory.examples[l]) == 100 for l in ["positive", "negative", "neutral"])


def test_external_edit_of_examples_is_detected(memory, emb):
    """classifier.py assigns memory.examples[label] directly (:883); cached sums must not go stale."""
    for i in range(4):
        memory.add_example(Example(f"t{i}", "a", emb + i), "a")
    memory.examples["a"] = memory.examples["a"][:2]
    memory.add_example(Example("new", "a", emb + 10), "a")
    want = torch.stack([e.embedding for e in memory.examples["a"]]).mean(0)
    assert torch.allclose(memory.prototypes["a"], want, atol=1e-6)


# ---- models ---------------------------------------------------------------------------------
def test_config_surface():
    c = ModelConfig({"max_length": 128})
    assert c.max_length == 128 and c.max_examples_per_class == 1000 and c.prototype_update_frequency == 100
    c.update(batch_size=8, not_a_key=1)
    assert c.batch_size == 8 and not hasattr(c, "not_a_key")
    d = c.to_dict()
    assert d["ewc_lambda"] == 100.0 and d["prototype_weight"] == 0.7 and len(d) == 29


def test_example_roundtrip():
    e = Example("t", "l", torch.arange(4.0))
    assert torch.equal(Example.from_dict(e.to_dict()).embedding, e.embedding)
    assert Example.from_dict(Example("t", "l").to_dict()).embedding is None


def test_head_anatomy_and_determinism():
    h1, h2 = AdaptiveHead(768, 4, [768, 384]), AdaptiveHead(768, 4, [768, 384])
    assert list(h1.state_dict()) == [f"model.{i}.{p}" for i in (0, 3, 6) for p in ("weight", "bias")]
    assert all(torch.equal(a, b) for a, b in zip(h1.state_dict().values(), h2.state_dict().values()))
    assert isinstance(h1.model[0], nn.Linear) and isinstance(h1.model[2], nn.Dropout) and h1.model[-1].out_features == 4
    assert h1(torch.randn(768)).shape == (1, 4) and h1(torch.randn(5, 768)).shape == (5, 4)
    old_w = h1.model[-1].weight.detach().clone()
    h1.update_num_classes(6)
    assert h1.model[-1].weight.shape == (6, 384) and torch.equal(h1.model[-1].weight[:4], old_w)
    assert sum(p.numel() for p in AdaptiveHead(768, 4, [768, 384]).parameters()) == 887428


def test_head_flat_block_views():
    h = AdaptiveHead(16, 3, [16, 8])
    flat = h.flat_params()
    assert flat.numel() == sum(p.numel() for p in h.parameters())
    with torch.no_grad():
        flat.zero_()
    assert all(float(p.abs().sum()) == 0 for p in h.parameters())        # parameters are views
    sd = {k: torch.ones_like(v) for k, v in h.state_dict().items()}
    h.load_state_dict(sd)
    assert float(h.flat_params().sum()) == flat.numel()                 # load_state_dict writes through


# ---- EWC generic-module contract (reference tests/test_ewc.py) ---------------------------------
class _Simple(nn.Module):
    def __init__(self, i=10, c=3):
        super().__init__()
        self.fc = nn.Linear(i, c)

    def forward(self, x):
        return self.fc(x)


@pytest.mark.parametrize("size", [1, 31, 32, 33, 64, 65, 100])
def test_ewc_various_sizes(size):
    model = _Simple()
    ds = torch.utils.data.TensorDataset(torch.randn(size, 10), torch.randint(0, 3, (size,)))
    ewc = EWC(model, ds, device="cpu", ewc_lambda=100.0)
    assert ewc.fisher_info and set(ewc.fisher_info) == set(ewc.old_params) == {"fc.weight", "fc.bias"}
    assert ewc.ewc_loss(batch_size=32).item() >= 0


def test_ewc_loss_after_perturbation():
    model = _Simple()
    ds = torch.utils.data.TensorDataset(torch.randn(33, 10), torch.tensor([0, 1, 2] * 11))
    ewc = EWC(model, ds, device="cpu", ewc_lambda=100.0)
    assert ewc.ewc_loss().item() == 0.0
    for p in model.parameters():
        p.data += 0.1
    l, ln = ewc.ewc_loss(), ewc.ewc_loss(batch_size=32)
    assert l.item() > 0 and ln.item() > 0 and ln.item() != l.item()
    l.backward()                                                       # autograd flows into the model
    assert model.fc.weight.grad is not None


def test_ewc_single_sample():
    model = _Simple(5, 2)
    ewc = EWC(model, torch.utils.data.TensorDataset(torch.randn(1, 5), torch.tensor([0])), ewc_lambda=50.0)
    assert ewc.ewc_loss() is not None


# ---- blend formulas (host logic of predict / predict_batch) against the reference's outputs -------
def test_blend_matches_reference_golden():
    """AdaptiveClassifier._blend (vectorised, fp64) reproduces reference _predict_regular / predict_batch
    (tests/golden/memory_blend.json) when fed the oracle's kNN scores and head probabilities."""
    import json
    import os
    import numpy as np
    from adaptive_classifier import AdaptiveClassifier
    from oracle import head_oracle, knn_oracle, synth
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "memory_blend.json")))
    C, per, D = 4, 25, 768
    labels = [f"c{c}" for c in range(C)]
    X, cent = synth.synth_unit_rows(C * per, D, 10), synth.synth_unit_rows(C, D, 11)
    protos = np.zeros((C, D), np.float64)
    for i in range(C * per):
        v = X[i] * 0.5 + cent[i % C]
        protos[i % C] += (v / np.linalg.norm(v)).astype(np.float32)
    protos = (protos / per).astype(np.float32)
    Q = synth.synth_unit_rows(8, D, 77)
    Q = np.stack([(q * 0.5 + cent[i % 4]) / np.linalg.norm(q * 0.5 + cent[i % 4]) for i, q in enumerate(Q)]).astype(np.float32)
    clf = AdaptiveClassifier.__new__(AdaptiveClassifier)
    clf.label_to_id = {l: i for i, l in enumerate(labels)}
    clf.id_to_label = {i: l for i, l in enumerate(labels)}
    clf.training_history = {"c0": 25, "c1": 5, "c2": 25, "c3": 9}
    clf.memory = type("M", (), {"_row_labels": None})()
    with torch.no_grad():
        P = torch.softmax(head_oracle.make_head(768, 4).eval()(torch.from_numpy(Q)), 1).numpy()

    def same(got, want):
        assert [l for l, _ in got] == [l for l, _ in want]
        assert np.allclose([s for _, s in got], [s for _, s in want], atol=1e-6)
        assert all(isinstance(s, float) for _, s in got)

    for key, want in g["predict"].items():
        Dd, I = knn_oracle.knn_l2_topk(protos, Q, 4)
        for a, w in zip(clf._blend(knn_oracle.proto_scores(Dd, I), I, P, int(key[1:]), True), want):
            same(a, w)
    for key, want in g["predict_batch"].items():
        k = int(key[1:])
        Dd, I = knn_oracle.knn_l2_topk(protos, Q, k)
        for a, w in zip(clf._blend(knn_oracle.proto_scores(Dd, I), I, P, k, False), want):
            same(a, w)


# ---- multi-label decision logic (multilabel.py:112-226) against the reference's outputs ------------
def test_multilabel_thresholds_and_decisions_match_reference():
    import json
    import os
    from adaptive_classifier import MultiLabelAdaptiveClassifier, MultiLabelAdaptiveHead
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "multilabel.json")))
    clf = MultiLabelAdaptiveClassifier.__new__(MultiLabelAdaptiveClassifier)
    labels = [f"l{i}" for i in range(6)]
    clf.label_to_id = {l: i for i, l in enumerate(labels)}
    clf.id_to_label = {i: l for i, l in enumerate(labels)}
    clf.default_threshold = 0.5
    for n, want in g["adaptive_thresholds"].items():
        assert clf._get_adaptive_threshold(int(n)) == want
    for d in g["decisions"]:
        clf.min_predictions, clf.max_predictions, clf.label_thresholds = d["min"], d["max"], d["label_thresholds"]
        thr = d["threshold"] if d["threshold"] is not None else clf._get_adaptive_threshold(6)
        got = clf._decide(d["probs"], thr, d["max_labels"] or clf.max_predictions)
        assert [l for l, _ in got] == [l for l, _ in d["out"]], d
        assert all(abs(a - b) < 1e-6 for (_, a), (_, b) in zip(got, d["out"]))
    h = MultiLabelAdaptiveHead(16, 3, [16, 8])
    assert h.num_classes == 3 and h(torch.randn(2, 16)).min() >= 0 and h(torch.randn(2, 16)).max() <= 1
    w = h.model[-1].weight.detach().clone()
    h.update_num_classes(5)
    assert h.num_classes == 5 and torch.equal(h.model[-1].weight[:3], w) and h(torch.randn(4, 16)).shape == (4, 5)


def test_prune_matches_reference_algorithm_step_by_step():
    """memory.py:196-217 restated literally (stack, mean, per-example torch.norm, argsort, keep closest in
    ascending-distance order) vs the vectorised class-matrix implementation, after every add."""
    import numpy as np
    m = PrototypeMemory(16, ModelConfig({"max_examples_per_class": 7}))
    E = torch.randn(60, 16, generator=torch.Generator().manual_seed(0))
    ref = []
    for i in range(60):
        m.add_example(Example(f"t{i}", "x", E[i]), "x")
        ref.append(E[i])
        if len(ref) > 7:
            mean = torch.stack(ref).mean(0)
            keep = np.argsort([torch.norm(e - mean).item() for e in ref])[:7]
            ref = [ref[j] for j in keep]
        assert len(m.examples["x"]) == len(ref)
        assert all(torch.equal(a.embedding, b) for a, b in zip(m.examples["x"], ref)), i
        assert torch.allclose(m.prototypes["x"], torch.stack(ref).mean(0), atol=1e-6)


def test_epoch_order_equals_the_seeded_dataloader():
    """AdaptiveClassifier._EpochOrder reproduces, epoch after epoch, the index order of the reference's
    DataLoader(shuffle=True, generator=Generator().manual_seed(42)) (classifier.py:1454-1459) without iterating a
    DataLoader -- it depends on how many numbers torch's loader / sampler draw per epoch, so it is pinned here."""
    import torch
    from adaptive_classifier import AdaptiveClassifier
    for n, bs in ((1, 1), (5, 32), (33, 32), (64, 32), (100, 32), (777, 32)):
        loader = AdaptiveClassifier._index_loader(n, min(bs, n))
        fast = AdaptiveClassifier._EpochOrder(n)
        for epoch in range(5):
            want = torch.cat([idx for (idx,) in loader])
            assert torch.equal(fast.next_epoch(), want), (n, epoch)


def test_add_examples_batch_host_fallback_equals_add_example():
    """Without a GPU `add_examples_batch` runs the per-example host logic for overflowing classes and the O(k)
    append path otherwise; either way the state equals a loop of `add_example` (order, prototypes, counters)."""
    import torch
    from adaptive_classifier.memory import PrototypeMemory
    from adaptive_classifier.models import Example, ModelConfig
    cfg = ModelConfig({"max_examples_per_class": 9, "prototype_update_frequency": 5})
    a, b = PrototypeMemory(16, cfg), PrototypeMemory(16, cfg)
    g = torch.Generator().manual_seed(4)
    n = 0
    for chunk in (3, 8, 1, 20, 6):
        vs = [torch.nn.functional.normalize(torch.randn(16, generator=g), dim=0) for _ in range(chunk)]
        ls = ["p" if (n + i) % 3 else "q" for i in range(chunk)]
        n += chunk
        try:
            for v, l in zip(vs, ls):
                a.add_example(Example(f"t{id(v)}", l, v.clone()), l)
            b.add_examples_batch([Example(f"t{id(v)}", l, v.clone()) for v, l in zip(vs, ls)], ls)
        except Exception as e:                       # index rebuild needs a GPU: state up to that point still comparable
            if "GPU" not in str(e) and "gpu" not in str(e):
                raise
            return
        for l in ("p", "q"):
            assert [e.text for e in a.examples[l]] == [e.text for e in b.examples[l]]
            assert torch.allclose(a.prototypes[l], b.prototypes[l], atol=1e-6)
        assert a.updates_since_rebuild == b.updates_since_rebuild


def test_ewc_generic_module_contract():
    """The EWC class on an arbitrary nn.Module (the reference's tests/test_ewc.py:34-84,128-153 scenarios: nn.Linear
    model, dataset sizes that leave a 1-sample last batch, penalty after p += 0.1 with and without batch_size), on CPU
    through the generic autograd route -- and equal to the reference's formula evaluated by hand with the same RNG."""
    import torch.nn as nn
    import torch.nn.functional as F
    from adaptive_classifier.ewc import EWC

    class SimpleModel(nn.Module):
        def __init__(self, input_dim=10, num_classes=3):
            super().__init__()
            self.fc = nn.Linear(input_dim, num_classes)

        def forward(self, x):
            return self.fc(x)

    for size in (1, 31, 32, 33, 64, 65, 100):                       # test_ewc.py:56-84
        torch.manual_seed(size)
        model = SimpleModel()
        ds = torch.utils.data.TensorDataset(torch.randn(size, 10), torch.randint(0, 3, (size,)))
        # the reference algorithm by hand (ewc.py:51-94), consuming the global RNG the same way
        state = torch.get_rng_state()
        loader = torch.utils.data.DataLoader(ds, batch_size=32, shuffle=True)
        want = {n: torch.zeros_like(p) for n, p in model.named_parameters()}
        for xb, _ in loader:
            model.zero_grad()
            out = model(xb)
            y = torch.multinomial(F.softmax(out, dim=1), 1).squeeze(-1)
            F.nll_loss(F.log_softmax(out, dim=1), y).backward()
            for n, p in model.named_parameters():
                want[n] += p.grad.data ** 2 / len(loader)
        model.zero_grad()
        torch.set_rng_state(state)
        ewc = EWC(model, ds, device="cpu", ewc_lambda=100.0)
        assert set(ewc.fisher_info) == set(ewc.old_params) == {"fc.weight", "fc.bias"}
        for n in want:
            assert torch.equal(ewc.fisher_info[n], want[n])
            assert torch.equal(ewc.old_params[n], dict(model.named_parameters())[n].data)
        assert ewc.ewc_loss(batch_size=32).item() == 0.0            # unchanged parameters: penalty exactly 0
        for p_ in model.parameters():                               # test_ewc.py:140-153
            p_.data += 0.1
        loss = ewc.ewc_loss()
        loss_n = ewc.ewc_loss(batch_size=32)
        ref = 100.0 * sum((want[n] * (p_ - ewc.old_params[n]) ** 2).sum() for n, p_ in model.named_parameters())
        assert loss.item() > 0 and abs(loss.item() - ref.item()) <= 1e-6 * abs(ref.item())
        assert abs(loss_n.item() - loss.item() / 32) <= 1e-6 * abs(loss.item())
        loss.backward()                                             # autograd reaches the live parameters
        assert all(p_.grad is not None and p_.grad.abs().sum() > 0 for p_ in model.parameters())


def test_mirrors_follow_direct_edits_of_examples_with_unchanged_length():
    """`memory.examples` is public and the reference's callers edit it directly.  Replacing entries while keeping
    the list LENGTH must not leave the cached fp64 sums / class matrices stale: the next prototype update and prune
    use the edited list, exactly like the reference, which always recomputes from the list (memory.py:138-159)."""
    from adaptive_classifier import Example, ModelConfig, PrototypeMemory
    from oracle import synth
    D = 32
    X = torch.from_numpy(synth.synth_unit_rows(40, D, 5))
    mem = PrototypeMemory(D, ModelConfig({"max_examples_per_class": 12}))
    for i in range(10):
        mem.add_example(Example(f"t{i}", "a", X[i].clone()), "a")
    assert torch.allclose(mem.prototypes["a"], X[:10].mean(0), atol=1e-6)
    # caller swaps two stored examples for new ones: same length, different content
    mem.examples["a"][3] = Example("new3", "a", X[20].clone())
    mem.examples["a"][7].embedding = X[21].clone()
    want_rows = [X[i] for i in range(10)]
    want_rows[3], want_rows[7] = X[20], X[21]
    mem._update_prototype("a")
    assert torch.allclose(mem.prototypes["a"], torch.stack(want_rows).mean(0), atol=1e-6)
    # the next add (per-example and batched paths) builds on the edited list, not on the stale mirror
    mem.add_example(Example("t10", "a", X[10].clone()), "a")
    want_rows.append(X[10])
    assert torch.allclose(mem.prototypes["a"], torch.stack(want_rows).mean(0), atol=1e-6)
    mem.examples["a"][0] = Example("new0", "a", X[22].clone())
    want_rows[0] = X[22]
    mem.add_examples_batch([Example("t11", "a", X[11].clone())], ["a"])
    want_rows.append(X[11])
    assert torch.allclose(mem.prototypes["a"], torch.stack(want_rows).mean(0), atol=1e-6)
    # ... and a prune after an edit keeps the 12 closest to the mean of the EDITED list
    mem.examples["a"][1] = Example("far", "a", (-X[1]).clone())
    want_rows[1] = -X[1]
    mem.add_example(Example("t12", "a", X[12].clone()), "a")
    want_rows.append(X[12])
    allr = torch.stack(want_rows)
    dist = (allr - allr.mean(0)).norm(dim=1)
    keep = torch.argsort(dist)[:12]
    assert sorted(e.text for e in mem.examples["a"]) == sorted(["new0", "far", "t2", "new3", "t4", "t5", "t6", "t7", "t8",
                                                                 "t9", "t10", "t11", "t12"][i] for i in keep.tolist())
    assert torch.allclose(mem.prototypes["a"], allr[keep].mean(0), atol=1e-6)


def test_wordpiece_hash_and_tokenizer_recognition():
    """CPU side of the device tokenizer: ac_wordpiece_hash is FNV-1a 64 with the "##" prefix folded in, and only BERT
    WordPiece tokenizers (BertNormalizer + BertPreTokenizer + WordPiece) are taken over."""
    import ctypes
    from transformers import BertTokenizer
    from adaptive_classifier import _native as nv
    from adaptive_classifier.tokenizer import _wordpiece_spec

    def fnv(b, cont):
        h = 0xcbf29ce484222325
        for c in (b"##" if cont else b"") + b:
            h = ((h ^ c) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
        return h or 1

    for raw, cont in ((b"play", 0), (b"ing", 1), (b"a", 0), (b"x" * 40, 1)):
        buf = (ctypes.c_uint8 * len(raw)).from_buffer_copy(raw)
        assert nv.lib().ac_wordpiece_hash(buf, len(raw), cont) == fnv(raw, cont)
    vocab = {t: i for i, t in enumerate(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "a", "##b"])}
    spec = _wordpiece_spec(BertTokenizer(vocab=vocab, do_lower_case=False))
    assert spec is not None and spec[0] == vocab and spec[1] is False and "[SEP]" in spec[2]
    from helpers import HashTokenizer
    assert _wordpiece_spec(HashTokenizer()) is None


def test_representative_examples_equal_the_reference_written_file():
    """N1: tests/golden/ref_saved_d64/examples.json was written by the reference's _save_pretrained from the store that
    gen_golden.py's build_memory(C=4, per_class=12, D=64, seed=50) makes; select_representative_examples must keep the
    same examples in the same order (classifier.py:560-566, :1533-1573)."""
    import json
    import os
    import numpy as np
    from adaptive_classifier.classifier import select_representative_examples
    from oracle import synth
    C, per, D, seed = 4, 12, 64, 50
    X, cent = synth.synth_unit_rows(C * per, D, seed), synth.synth_unit_rows(C, D, seed + 1)
    store = {f"c{c}": [] for c in range(C)}
    for i in range(C * per):
        c = i % C
        v = X[i] * 0.5 + cent[c]
        store[f"c{c}"].append(Example(f"t{i:03d}", f"c{c}", torch.from_numpy((v / np.linalg.norm(v)).astype(np.float32))))
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_saved_d64", "examples.json")))
    for label, exs in store.items():
        got = select_representative_examples(exs, k=5)
        assert [e.text for e in got] == [e["text"] for e in ref[label]]
        assert all(np.array_equal(np.asarray(r["embedding"], np.float32), g.embedding.numpy()) for r, g in zip(ref[label], got))
    few = store["c0"][:4]
    assert select_representative_examples(few, k=5) is few               # <= k: returned as they are (:1543-1544)


def test_unpack_of_the_packed_device_result():
    """AdaptiveClassifier._unpack (packed n | class ids | fp64 scores -> the reference's list of (label, score) per query):
    ragged result counts, k below / above the stored width, class ids mapped to labels, cached label array refreshed when
    the label set changes."""
    import numpy as np
    from adaptive_classifier.classifier import AdaptiveClassifier

    class Host:
        id_to_label = {0: "a", 1: "b", 2: "c", 3: "d"}

        def _raise_if_encoder_gave_up(self):
            raise AssertionError("no NaN in this test")

    h = Host()
    b, kk, C = 37, 4, 4
    rng = np.random.default_rng(0)
    n = rng.integers(1, kk + 1, b).astype(np.int32)
    cls = rng.integers(0, C, (b, kk)).astype(np.int32)
    val = rng.random((b, kk))
    off_cls = 4 * b
    off_val = (off_cls + 4 * b * kk + 7) // 8 * 8
    host = np.zeros(off_val + 8 * b * kk, np.uint8)
    host[:off_cls] = n.view(np.uint8)
    host[off_cls:off_cls + 4 * b * kk] = cls.view(np.uint8).ravel()
    host[off_val:] = val.view(np.uint8).ravel()
    layout = (b, kk, off_cls, off_val, C)
    for k in (1, 3, 4, 9):
        got = AdaptiveClassifier._unpack(h, host, layout, k)
        want = [[("abcd"[cls[q, j]], float(val[q, j])) for j in range(min(int(n[q]), k, kk))] for q in range(b)]
        assert got == want, k
    n[:] = kk                                                    # the common case: every query has kk results (one slice per row)
    host[:off_cls] = n.view(np.uint8)
    got = AdaptiveClassifier._unpack(h, host, layout, 5)
    assert got == [[("abcd"[cls[q, j]], float(val[q, j])) for j in range(kk)] for q in range(b)]
    h.id_to_label = {0: "w", 1: "x", 2: "y", 3: "z"}             # relabelled: no stale names
    assert AdaptiveClassifier._unpack(h, host, layout, 1)[0][0][0] == "wxyz"[cls[0, 0]]


def test_predict_retry_contract_without_a_gpu():
    """AdaptiveClassifier._predict_with_retry / _encoder_options (host logic only): the chain runs with verify=False first; NaN
    scores after a one-launch forward repeat it layer by layer; NaN scores with the encoder's fused-LayerNorm verdict set switch
    the fusion off (stubbed here) and repeat with verify=True; anything else is returned as computed.  A user-supplied encoder
    without the options is called without them."""
    from adaptive_classifier import AdaptiveClassifier
    from adaptive_classifier import classifier as cmod

    class Enc:
        def __init__(self, one_launch=False, ln_aborted=False):
            self.last_one_launch, self._ln, self.calls = one_launch, ln_aborted, []

        def encode_cls(self, ids, types=None, mask=None, verify=True, force_layered=False):
            self.calls.append((verify, force_layered))
            return "emb%d" % len(self.calls)

        def ln_fusion_aborted(self):
            return self._ln

    def run(enc, nan_until):
        c = AdaptiveClassifier.__new__(AdaptiveClassifier)
        c.model = enc
        seen = []

        def finish(emb):
            seen.append(emb)
            return ["result of " + emb], len(seen) <= nan_until
        return c, c._predict_with_retry(lambda v, f: c._encode_tokens(None, None, None, verify=v, force_layered=f), finish), seen

    c, res, seen = run(Enc(), 0)                                    # the common case: one pass, no sync asked of the encoder
    assert res == ["result of emb1"] and c.model.calls == [(False, False)]
    c, res, seen = run(Enc(one_launch=True), 1)                     # one-launch forward gave NaN -> layer by layer, verified
    assert res == ["result of emb2"] and c.model.calls == [(False, False), (True, True)]
    switched = []                                                   # the product switches the fusion off PER OBJECT (the process-wide
    enc_ln = Enc(ln_aborted=True)                                   # setters of the ABI are test hooks it never calls)
    enc_ln.disable_ln_fusion = lambda: switched.append(0)
    real_lib = cmod.nv.lib
    cmod.nv.lib = lambda: type("L", (), {"ac_gemm_set_ln_fusion": staticmethod(lambda on: (_ for _ in ()).throw(AssertionError("process-wide switch touched")))})
    try:
        c, res, seen = run(enc_ln, 1)                               # fused LayerNorm gave up -> fusion off, batch encoded again
    finally:
        cmod.nv.lib = real_lib
    assert res == ["result of emb2"] and c.model.calls == [(False, False), (True, False)] and switched == [0]
    c, res, seen = run(Enc(), 5)                                    # NaN that the encoder does not explain: returned as computed
    assert res == ["result of emb1"] and c.model.calls == [(False, False)]

    class EncF16(Enc):                                              # opt-in fp16x2 arithmetic: NaN = an operand left fp16's range
        f16x2_overflows, _f16 = 0, True

        def f16x2_active(self):
            return self._f16

        def disable_f16x2(self):
            self._f16 = False

    class EncF16New(EncF16):                                        # ... and the current interface: the choice is demoted, planes stay
        def set_arith(self, a):
            self._f16 = False
            self.arith_set = a

        def disable_f16x2(self):
            raise AssertionError("the planes of a shared encoder must not be dropped")
    c, res, seen = run(EncF16New(), 1)
    assert res == ["result of emb2"] and c.model.f16x2_overflows == 1 and not c.model.f16x2_active() and c.model.arith_set == 1
    c, res, seen = run(EncF16(), 1)                                 # -> the encoder goes back to bf16x3, the batch is encoded again
    assert res == ["result of emb2"] and c.model.calls == [(False, False), (True, False)]
    assert c.model.f16x2_overflows == 1 and not c.model.f16x2_active()
    c, res, seen = run(EncF16(), 0)                                 # finite scores: fp16x2 stays on, one pass
    assert res == ["result of emb1"] and c.model.f16x2_overflows == 0 and c.model.f16x2_active()

    class Plain:                                                    # a user encoder: no options in its signature
        last_one_launch = False

        def __init__(self):
            self.calls = 0

        def encode_cls(self, ids, types=None, mask=None):
            self.calls += 1
            return "e"
    c = AdaptiveClassifier.__new__(AdaptiveClassifier)
    c.model = Plain()
    assert c._encoder_options() == set() and c._encode_tokens(None, verify=False, force_layered=True) == "e" and c.model.calls == 1


def test_hostfast_unpack_builds_the_lists_of_the_python_form():
    """csrc/host/hostfast.c (CPython extension, built by the same Makefile as the library): the packed device result ->
    the reference's list of (label, score) lists.  Same objects as classifier.py::_unpack's Python form for ragged counts,
    k smaller than the packed width, out-of-range class ids (clamped like np.clip) and NaN scores; a layout that does not
    fit the buffer is refused."""
    import numpy as np
    from adaptive_classifier import classifier as cm
    assert cm._hostfast is not None, "the _hostfast extension was not built (make -C adaptive-classifier_amd/csrc)"
    rng = np.random.default_rng(3)

    class Obj:
        _last_nan = False
        def _raise_if_encoder_gave_up(self):
            raise AssertionError("check=False must not ask the encoder")

    for (b, kk, C, k) in [(256, 4, 4, 16), (7, 16, 100, 5), (1, 1, 1, 1), (33, 5, 3, 0), (64, 8, 70, 8)]:
        off_cls = 4 * b
        off_val = (off_cls + 4 * b * kk + 7) // 8 * 8
        host = np.zeros(off_val + 8 * b * kk, np.uint8)
        host[:off_cls].view(np.int32)[:] = rng.integers(0, kk + 1, b)
        host[off_cls:off_cls + 4 * b * kk].view(np.int32)[:] = rng.integers(-1, C + 1, b * kk)
        host[off_val:].view(np.float64)[:] = rng.random(b * kk)
        o = Obj(); o.id_to_label = {i: "label-%d" % i for i in range(C)}
        layout = (b, kk, off_cls, off_val, C)
        for nan in (False, True):
            if nan:
                host[off_val:].view(np.float64)[b * kk // 2] = np.nan
            fast = cm.AdaptiveClassifier._unpack(o, host, layout, k, False)
            assert o._last_nan == nan
            saved, cm._hostfast = cm._hostfast, None
            try:
                slow = cm.AdaptiveClassifier._unpack(o, host, layout, k, False)
            finally:
                cm._hostfast = saved
            assert o._last_nan == nan
            assert len(fast) == b and all(type(r) is list for r in fast)
            assert repr(fast) == repr(slow)              # (repr: nan == nan)
            assert all(type(t) is tuple and type(t[1]) is float for r in fast for t in r)
        if C <= 64:                     # (small label sets are re-read at every call; big ones are cached by dict identity + size)
            o.id_to_label[0] = "renamed"
            assert all(t[0] != "label-0" for r in cm.AdaptiveClassifier._unpack(o, host, layout, k, False) for t in r)
    with pytest.raises(ValueError):
        cm._hostfast.unpack(("a",), np.zeros(16, np.uint8), 4, 4, 16, 80, 4)
