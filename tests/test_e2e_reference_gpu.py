"""GPU suite: END-TO-END differential against the reference's own classifier.py, FROM TEXT.

tests/golden/e2e_bert_mini/ was written by tests/golden/gen_e2e.py running the UNMODIFIED reference on CPU -- tokenizer,
encoder, memory, head and blend all in the loop: `AdaptiveClassifier(name)`, two `add_examples` calls (the second adds a class),
`save`, then `predict` (/root/reference/src/adaptive_classifier/classifier.py:392-480) and `predict_batch` (:1308-1388).
Here the PRODUCT is built on the same checkpoint name through the same offline Hub stand-in (oracle/hub_standin.py: seeded
random-init 4-layer BERT, synthetic WordPiece vocabulary -> bit-identical weights and ids on both sides) and must return the
same labels in the same order with every score within 1e-4 (the reference's own CPU-vs-GPU bar is 1e-5,
tests/test_classifier.py:151-167; the 1e-4 is SURVEY 8c's fp32 tolerance).  The product's chain is the shipping one: device
WordPiece (ac_wordpiece_encode) -> padding-free encoder -> device kNN / scores -> head -> ac_blend_topk.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = os.path.join(GOLD, "e2e_bert_mini")
TOL = 1e-4
# further architectures through the same recipe (24 texts each): the encoders the reference's own tests name
CASES = {"bert": "e2e_bert_mini", "distilbert": "e2e_distilbert_mini", "modernbert": "e2e_modernbert_mini"}


@pytest.fixture(scope="module")
def standin():
    from oracle import hub_standin
    hub_standin.install()
    yield hub_standin
    hub_standin.uninstall()


@pytest.fixture(scope="module")
def exp():
    return json.load(open(os.path.join(G, "expected.json")))


def _same(got, want, what):
    assert [l for l, _ in got] == [l for l, _ in want], (what, got, want)
    assert all(isinstance(s, float) for _, s in got)
    d = max(abs(a - b) for (_, a), (_, b) in zip(got, want)) if want else 0.0
    assert d <= TOL, (what, d, got, want)
    return d


def test_hub_standin_is_in_the_loop(standin, cuda_dev, exp):
    """The product builds its encoder and tokenizer from the NAME, exactly as the reference does (classifier.py:83-85)."""
    from adaptive_classifier import AdaptiveClassifier
    from adaptive_classifier.tokenizer import HipWordPieceTokenizer
    clf = AdaptiveClassifier(exp["model_name"], device="cuda:0")
    assert type(clf.model).__name__ == "HipBertEncoder" and clf.embedding_dim == 128
    assert isinstance(clf.tokenizer, HipWordPieceTokenizer)            # the device tokenizer took the BERT vocabulary
    ref_tok = standin.make_tokenizer(exp["model_name"])
    texts = exp["texts"]
    want = ref_tok(texts, max_length=512, truncation=True, padding=True, return_tensors="pt")
    got = clf.tokenizer(texts, max_length=512, truncation=True, padding=True, return_tensors="pt")
    assert torch.equal(got["input_ids"].cpu(), want["input_ids"]) and torch.equal(got["attention_mask"].cpu(), want["attention_mask"])
    assert clf.tokenizer.host_texts >= 1 and clf.tokenizer.device_texts >= len(texts) - 2     # (the accented text goes to the host)


def test_embeddings_from_text_equal_the_references(standin, cuda_dev, exp):
    """_get_embeddings (classifier.py:1249-1282): tokenizer + encoder + CLS + normalise, list of CPU tensors."""
    from adaptive_classifier import AdaptiveClassifier
    clf = AdaptiveClassifier(exp["model_name"], device="cuda:0")
    want = np.asarray(exp["embeddings"])
    got = clf._get_embeddings(exp["texts"])
    assert isinstance(got, list) and all(e.device.type == "cpu" and e.shape == (128,) for e in got)
    d = np.abs(torch.stack(got).double().numpy() - want).max()
    assert d <= 1e-5, d
    one = np.stack([clf._get_embeddings([t])[0].double().numpy() for t in exp["texts"][:6]])      # the one-launch path (<= 32 rows)
    assert np.abs(one - want[:6]).max() <= 1e-5


def test_load_then_predict_and_predict_batch_equal_the_reference(standin, cuda_dev, exp):
    """The directory the reference wrote -> AdaptiveClassifier.load -> predict / predict_batch from text."""
    from adaptive_classifier import AdaptiveClassifier
    clf = AdaptiveClassifier.load(G, device="cuda:0")
    assert clf.label_to_id == exp["label_to_id"] and clf.training_history == exp["training_history"]
    texts = exp["texts"]
    assert len(texts) >= 32
    worst = 0.0
    for i, t in enumerate(texts):
        worst = max(worst, _same(clf.predict(t, k=2), [tuple(p) for p in exp["predict_k2"][i]], ("predict k=2", t)))
        worst = max(worst, _same(clf.predict(t, k=5), [tuple(p) for p in exp["predict_k5"][i]], ("predict k=5", t)))
    for key, k, kw in (("predict_batch_k1", 1, {}), ("predict_batch_k3", 3, {"batch_size": 16})):
        got = clf.predict_batch(texts, k=k, **kw)
        assert len(got) == len(texts)
        for i, t in enumerate(texts):
            worst = max(worst, _same(got[i], [tuple(p) for p in exp[key][i]], (key, t)))
    print("e2e from text: %d texts, 4 entry points, max |dscore| = %.2e" % (len(texts), worst))
    # and with the reference's chunking (min_device_batch = 1 -> batches of `batch_size` texts, classifier.py:1320-1322)
    clf.config.config["min_device_batch"] = 1
    got = clf.predict_batch(texts, k=3, batch_size=16)
    for i, t in enumerate(texts):
        _same(got[i], [tuple(p) for p in exp["predict_batch_k3"][i]], ("predict_batch k=3, 16-text chunks", t))


def test_add_examples_from_text_builds_the_references_memory(standin, cuda_dev, exp):
    """add_examples (classifier.py:132-200) from the same texts: same label ids, same training history, same prototypes
    (means of the encoder's embeddings, memory.py:138-159); then, with the REFERENCE's trained head loaded into the product's
    classifier (head training draws dropout masks from a different generator), the same predictions from text."""
    from safetensors.torch import load_file
    from adaptive_classifier import AdaptiveClassifier
    clf = AdaptiveClassifier(exp["model_name"], device="cuda:0")
    for part in ("train_1", "train_2"):
        clf.add_examples([t for t, _ in exp[part]], [l for _, l in exp[part]])
    assert clf.label_to_id == exp["label_to_id"] and clf.training_history == exp["training_history"]
    assert clf.adaptive_head.model[-1].out_features == 4
    for label, want in exp["prototypes"].items():
        d = np.abs(clf.memory.prototypes[label].double().cpu().numpy() - np.asarray(want)).max()
        assert d <= 1e-5, (label, d)
    # the product's own head (trained here) must already classify its training texts sensibly: scores sum to 1, all finite
    for t, _ in exp["train_2"][:3]:
        p = clf.predict(t, k=4)
        assert abs(sum(s for _, s in p) - 1.0) < 1e-6 and len(p) == 4
    tensors = load_file(os.path.join(G, "model.safetensors"))
    clf.adaptive_head.load_state_dict({k[len("adaptive_head_"):]: v for k, v in tensors.items() if k.startswith("adaptive_head_")})
    clf.adaptive_head = clf.adaptive_head.to(clf.device)
    for i, t in enumerate(exp["texts"]):
        _same(clf.predict(t, k=5), [tuple(p) for p in exp["predict_k5"][i]], ("predict after add_examples", t))
    got = clf.predict_batch(exp["texts"], k=3)
    for i, t in enumerate(exp["texts"]):
        _same(got[i], [tuple(p) for p in exp["predict_batch_k3"][i]], ("predict_batch after add_examples", t))


@pytest.mark.parametrize("family", ["distilbert", "modernbert"])
def test_other_encoder_families_from_text_equal_the_reference(standin, cuda_dev, family):
    """The same differential for the two other encoder families the reference's tests use (DistilBERT: tests/test_ewc.py:94,
    test_multilabel.py:35; ModernBERT: test_order_independence.py:10, test_confidence_consistency.py:14): directory written by the
    reference -> load -> embeddings, predict, predict_batch from text; then the product's own add_examples -> the reference's
    label ids, history and prototypes."""
    from adaptive_classifier import AdaptiveClassifier
    g = os.path.join(GOLD, CASES[family])
    ex = json.load(open(os.path.join(g, "expected.json")))
    clf = AdaptiveClassifier.load(g, device="cuda:0")
    assert type(clf.model).__name__ == {"distilbert": "HipBertEncoder", "modernbert": "HipModernBertEncoder"}[family]
    texts = ex["texts"]
    assert len(texts) >= 16
    d = np.abs(torch.stack(clf._get_embeddings(texts)).double().numpy() - np.asarray(ex["embeddings"])).max()
    assert d <= 1e-5, d
    for i, t in enumerate(texts):
        _same(clf.predict(t, k=2), [tuple(p) for p in ex["predict_k2"][i]], ("predict k=2", family, t))
        _same(clf.predict(t, k=5), [tuple(p) for p in ex["predict_k5"][i]], ("predict k=5", family, t))
    for key, k, kw in (("predict_batch_k1", 1, {}), ("predict_batch_k3", 3, {"batch_size": 16})):
        got = clf.predict_batch(texts, k=k, **kw)
        for i, t in enumerate(texts):
            _same(got[i], [tuple(p) for p in ex[key][i]], (key, family, t))
    fresh = AdaptiveClassifier(ex["model_name"], device="cuda:0")
    for part in ("train_1", "train_2"):
        fresh.add_examples([t for t, _ in ex[part]], [l for _, l in ex[part]])
    assert fresh.label_to_id == ex["label_to_id"] and fresh.training_history == ex["training_history"]
    for label, want in ex["prototypes"].items():
        assert np.abs(fresh.memory.prototypes[label].double().cpu().numpy() - np.asarray(want)).max() <= 1e-5, (family, label)


def test_multilabel_classifier_from_text_equals_the_reference(standin, cuda_dev):
    """MultiLabelAdaptiveClassifier (multilabel.py:70-413) from text: add_examples with label lists (flattened pairs, memory, label
    thresholds by frequency), then -- with the reference's trained sigmoid head loaded (its BCE training draws dropout masks from
    another generator) -- predict_multilabel with the default / an explicit threshold / max_labels and the min_predictions fill,
    and predict().  (The reference's own load() raises a TypeError, so the head travels as model.safetensors: gen_e2e.py.)"""
    from safetensors.torch import load_file
    from adaptive_classifier import MultiLabelAdaptiveClassifier
    g = os.path.join(GOLD, "e2e_multilabel_mini")
    ex = json.load(open(os.path.join(g, "expected.json")))
    clf = MultiLabelAdaptiveClassifier(ex["model_name"], device="cuda:0", **ex["ctor"])
    texts, labels = [t for t, _ in ex["train"]], [l for _, l in ex["train"]]
    clf.add_examples(texts, labels)
    clf.add_examples(texts[:6], labels[:6])
    assert clf.label_to_id == ex["label_to_id"] and clf.training_history == ex["training_history"]
    assert clf.label_thresholds == pytest.approx(ex["label_thresholds"])
    assert {l: len(v) for l, v in clf.memory.examples.items()} == ex["examples_per_class"]
    for label, want in ex["prototypes"].items():
        assert np.abs(clf.memory.prototypes[label].double().cpu().numpy() - np.asarray(want)).max() <= 1e-5, label
    tensors = load_file(os.path.join(g, "model.safetensors"))
    clf.adaptive_head.load_state_dict({k[len("adaptive_head_"):]: v for k, v in tensors.items()})
    clf.adaptive_head = clf.adaptive_head.to(clf.device)
    assert len(ex["texts"]) >= 20
    for i, t in enumerate(ex["texts"]):
        _same(clf.predict_multilabel(t), [tuple(p) for p in ex["multilabel_default"][i]], ("default", t))
        _same(clf.predict_multilabel(t, threshold=0.51), [tuple(p) for p in ex["multilabel_thr_0.51"][i]], ("thr 0.51", t))
        _same(clf.predict_multilabel(t, threshold=0.9, max_labels=2), [tuple(p) for p in ex["multilabel_thr_0.9_max2"][i]], ("thr 0.9 max 2", t))
        _same(clf.predict(t, k=3), [tuple(p) for p in ex["predict_k3"][i]], ("predict k=3", t))
    batch = clf.predict_multilabel_batch(ex["texts"])
    for i, t in enumerate(ex["texts"]):
        _same(batch[i], [tuple(p) for p in ex["multilabel_default"][i]], ("batched default", t))


def _rng_hashes():
    import hashlib
    st = np.random.get_state()
    return {"torch_cpu": hashlib.sha256(torch.get_rng_state().numpy().tobytes()).hexdigest(),
            "numpy": hashlib.sha256(st[1].tobytes() + str(st[2:]).encode()).hexdigest()}


@pytest.mark.parametrize("case", ["bert_mini", "bert_base"])
def test_training_trajectory_replays_the_reference_from_text(standin, cuda_dev, case):
    """The TRAINING differential: `add_examples` -> the product's OWN trained head -> `predict`, differenced
    against the unmodified reference's CPU run from the same texts and seeds (tests/golden/gen_e2e_train.py; reference
    classifier.py:1428-1522 `_train_adaptive_head`, :202-367 `_train_new_classes`, ewc.py:39-94).  With
    config={"dropout_source": "torch_cpu"} the product draws every dropout mask -- and every other draw the reference makes from
    torch's global generator and numpy's -- as the reference does, so the two runs see the same masks, batches and samples:
    same number of steps (= same early-stopping epoch), per-epoch average loss within 1e-5 relative (the verdict's bar: 1e-4; first
    run on MI355X: 7e-7), both generators left in the reference's final state after each call, and from the product's own head
    identical label order with |dscore| <= 1e-5 on every fixture text (bar: 1e-3; first run: 4e-7).  No head trained by the reference is loaded anywhere in this test.  `bert_base` is the 12 x 768 architecture
    (device WordPiece -> packed 768-d encoder -> 768-d kNN -> 768 -> 768 -> 384 head -> blend, three batches per epoch)."""
    from adaptive_classifier import AdaptiveClassifier
    ex = json.load(open(os.path.join(GOLD, "e2e_train_%s.json" % case)))
    torch.manual_seed(0)
    np.random.seed(0)
    clf = AdaptiveClassifier(ex["model_name"], device="cuda:0", config={"dropout_source": "torch_cpu"})
    texts = ex["texts"]
    assert len(texts) >= 32
    worst_score = worst_loss = 0.0
    for ci, (part, call) in enumerate(zip(("train_1", "train_2"), ex["calls"])):
        n_logs = len(clf.train_log)
        clf.add_examples([t for t, _ in ex[part]], [l for _, l in ex[part]])
        assert clf.label_to_id == call["label_to_id"]
        assert {l: len(v) for l, v in clf.memory.examples.items()} == call["examples_per_class"]
        log = clf.train_log[n_logs:]
        assert len(log) == 1
        want_steps = call["step_losses"]
        assert log[0]["steps"] == len(want_steps), (part, log[0]["steps"], len(want_steps))          # same early-stopping epoch
        n_ep = len(log[0]["epoch_losses"])
        per = len(want_steps) // n_ep
        assert per * n_ep == len(want_steps) and per == -(-log[0]["rows"] // 32)
        for e, got in enumerate(log[0]["epoch_losses"]):
            want = sum(want_steps[e * per:(e + 1) * per]) / per
            worst_loss = max(worst_loss, abs(got - want) / abs(want))
            assert abs(got - want) <= 1e-5 * abs(want), (part, e, got, want)
        assert _rng_hashes() == call["rng_after"], (part, "a generator is not where the reference left it")
        k_all = len(clf.label_to_id)
        got_b = clf.predict_batch(texts, k=3)
        for i, t in enumerate(texts):
            for got, want, what in ((clf.predict(t, k=k_all), call["predict_all"][i], "predict"), (got_b[i], call["predict_batch_k3"][i], "predict_batch")):
                assert [l for l, _ in got] == [l for l, _ in want], (part, what, t, got, want)
                d = max(abs(a - b) for (_, a), (_, b) in zip(got, want))
                worst_score = max(worst_score, d)
                assert d <= 1e-5, (part, what, t, d, got, want)
    print("training differential %s: %d + %d steps, worst relative epoch-loss difference %.2e, worst |dscore| %.2e over %d texts x 2 calls x 2 "
          "entry points" % (case, len(ex["calls"][0]["step_losses"]), len(ex["calls"][1]["step_losses"]), worst_loss, worst_score, len(texts)))


def test_unknown_dropout_source_is_refused(standin, cuda_dev):
    from adaptive_classifier import AdaptiveClassifier
    with pytest.raises(ValueError, match="dropout_source"):
        AdaptiveClassifier("standin/bert-mini-4l", device="cuda:0", config={"dropout_source": "gpu"})
