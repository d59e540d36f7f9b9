"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED reference
(/root/reference/src/adaptive_classifier) in the build container.

    python tests/golden/gen_golden.py

`faiss` is not installable here, so oracle/faiss_shim.py (exact-L2 numpy shim) is injected as
sys.modules["faiss"]; everything else (memory.py, models.py, ewc.py, classifier.py blend formulas,
torch's CrossEntropyLoss / clip_grad_norm_ / AdamW) is the reference's own code.  Inputs are
regenerated at test time from oracle/synth.py (bit-identical everywhere) or stored when they come
from the reference's own fixture (scripts/adaptive_router).  /root/reference is NOT available on the
GPU box, hence the committed outputs.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import faiss_shim, synth  # noqa: E402

faiss_shim.install()
sys.path.insert(0, "/root/reference/src")
import adaptive_classifier as ref  # noqa: E402  (the reference package)
from adaptive_classifier.ewc import EWC as RefEWC  # noqa: E402

assert ref.__file__.startswith("/root/reference"), ref.__file__

SAMPLE_IDX = np.random.default_rng(0).integers(0, 2 ** 31, size=64)


def summarize(t):
    """Compact, order-sensitive fingerprint of a big tensor: 64 sampled entries + sum + abs-sum."""
    f = t.detach().reshape(-1).double().numpy()
    idx = SAMPLE_IDX % f.size
    return {"n": int(f.size), "idx": idx.tolist(), "vals": f[idx].tolist(), "sum": float(f.sum()),
            "abssum": float(np.abs(f).sum())}


def build_memory(C=4, per_class=25, D=768, seed=10):
    """cfg0 'plumbing' store: C classes x per_class unit-norm synthetic embeddings, reference memory."""
    mem = ref.PrototypeMemory(D)
    labels = [f"c{c}" for c in range(C)]
    X = synth.synth_unit_rows(C * per_class, D, seed)
    # class structure: add a class-specific offset direction then renormalise (all deterministic)
    cent = synth.synth_unit_rows(C, D, seed + 1)
    for i in range(C * per_class):
        c = i % C
        v = X[i] * 0.5 + cent[c]
        v = (v / np.linalg.norm(v)).astype(np.float32)
        mem.add_example(ref.Example(f"t{i:03d}", labels[c], torch.from_numpy(v)), labels[c])
    mem._rebuild_index()
    return mem, labels


def gen_router():
    """The reference's own saved-classifier fixture (scripts/adaptive_router): real 768-d embeddings
    and prototypes -> reference memory scores and seed-42-head logits."""
    from safetensors.torch import load_file
    cfg = json.load(open("/root/reference/scripts/adaptive_router/config.json"))
    tens = load_file("/root/reference/scripts/adaptive_router/tensors.safetensors")
    labels = ["HIGH", "LOW"]
    emb = np.stack([np.asarray(e["embedding"], np.float32) for l in labels for e in cfg["examples"][l]])
    protos = np.stack([tens[f"prototype_{l}"].numpy() for l in labels])
    mem = ref.PrototypeMemory(768)
    for l, p in zip(labels, protos):
        mem.prototypes[l] = torch.from_numpy(p)
    mem._restore_from_save()
    scores = np.zeros((10, 2)); order = np.zeros((10, 2), np.int64)
    for i in range(10):
        res = mem.get_nearest_prototypes(torch.from_numpy(emb[i]), k=2)
        for j, (lab, sc) in enumerate(res):
            order[i, j] = labels.index(lab); scores[i, j] = sc
    dist = ((emb[:, None, :].astype(np.float64) - protos[None].astype(np.float64)) ** 2).sum(-1)
    head = ref.AdaptiveHead(768, 2, [768, 384]).eval()          # seed-42 init, reproducible anywhere
    with torch.no_grad():
        logits = head(torch.from_numpy(emb)).numpy()
    # the fixture's trained head: logits stored, weights are NOT copied (3.5 MB); checked via checksum only
    np.savez_compressed(os.path.join(HERE, "router_fixture.npz"), emb=emb, protos=protos, order=order,
                        scores=scores, dist=dist, seed42_logits=logits)


def gen_knn():
    cases = {}
    specs = [("small_k1", 50, 768, 4, 1, 1), ("k_eq_N", 20, 128, 3, 20, 2), ("ragged", 333, 1024, 5, 16, 3),
             ("tiles", 257, 768, 2, 32, 4)]
    for name, N, D, nq, k, seed in specs:
        P = synth.synth_unit_rows(N, D, seed); Q = synth.synth_unit_rows(nq, D, seed + 100)
        idx = ref.memory.faiss.IndexFlatL2(D); idx.add(P)
        Dd, Ii = idx.search(Q, k)
        cases[name] = {"N": N, "D": D, "nq": nq, "k": k, "seed": seed, "I": Ii.tolist(), "D_out": Dd.tolist()}
    # duplicates: ids must come back lowest-first
    P = np.concatenate([synth.synth_unit_rows(10, 768, 9)] * 3); Q = P[:2]
    idx = ref.memory.faiss.IndexFlatL2(768); idx.add(P)
    Dd, Ii = idx.search(Q, 6)
    cases["dups"] = {"I": Ii.tolist(), "D_out": Dd.tolist()}
    json.dump(cases, open(os.path.join(HERE, "knn_cases.json"), "w"))


def gen_memory_and_blend():
    """Reference PrototypeMemory scores + the two blend formulas (_predict_regular, predict_batch)
    driven with given embeddings (encoder bypassed by patching _get_embeddings)."""
    mem, labels = build_memory()
    Q = synth.synth_unit_rows(8, 768, 77)
    cent = synth.synth_unit_rows(4, 768, 11)
    Q = np.stack([(q * 0.5 + cent[i % 4]) / np.linalg.norm(q * 0.5 + cent[i % 4]) for i, q in enumerate(Q)]).astype(np.float32)
    out = {"protos": {l: mem.prototypes[l].numpy().tolist()[:8] for l in labels}}
    out["nearest"] = [[(l, s) for l, s in mem.get_nearest_prototypes(torch.from_numpy(q), k=4)] for q in Q]
    out["nearest_k2"] = [[(l, s) for l, s in mem.get_nearest_prototypes(torch.from_numpy(q), k=2)] for q in Q]
    clf = ref.AdaptiveClassifier.__new__(ref.AdaptiveClassifier)
    clf.config = ref.ModelConfig(); clf.device = "cpu"; clf.memory = mem; clf.embedding_dim = 768
    clf.label_to_id = {l: i for i, l in enumerate(labels)}; clf.id_to_label = {i: l for i, l in enumerate(labels)}
    clf.training_history = {"c0": 25, "c1": 5, "c2": 25, "c3": 9}      # mixes the <10 / >=10 weight branches
    clf.strategic_cost_function = None
    clf.adaptive_head = ref.AdaptiveHead(768, 4, [768, 384])
    texts = [f"q{i}" for i in range(8)]
    table = {t: torch.from_numpy(q) for t, q in zip(texts, Q)}
    clf._get_embeddings = lambda ts: [table[t] for t in ts]
    out["predict"] = {f"k{k}": [clf._predict_regular(t, k) for t in texts] for k in (1, 3, 5)}
    out["predict_batch"] = {f"k{k}": clf.predict_batch(texts, k=k) for k in (1, 2, 5)}
    json.dump(out, open(os.path.join(HERE, "memory_blend.json"), "w"))


def gen_head_step():
    """Two reference training steps (classifier.py:1489-1505) with the dropout masks captured."""
    torch.manual_seed(0)
    head = ref.AdaptiveHead(768, 4, [768, 384])
    head.train()
    opt = torch.optim.AdamW(head.parameters(), lr=0.001, weight_decay=0.01, betas=(0.9, 0.999))
    crit = torch.nn.CrossEntropyLoss()
    X = torch.from_numpy(synth.synth_unit_rows(32, 768, 21))
    y = torch.from_numpy((np.arange(32) * 7 % 4).astype(np.int64))
    masks, steps = [], []
    cap = []
    hooks = [m.register_forward_hook(lambda mod, inp, out: cap.append(((out != 0) | (inp[0] == 0)).to(torch.uint8)))
             for m in head.model if isinstance(m, torch.nn.Dropout)]
    torch.manual_seed(123)
    for s in range(2):
        cap.clear()
        opt.zero_grad()
        loss = crit(head(X), y)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(head.parameters(), max_norm=1.0)
        opt.step()
        masks.append([c.numpy().copy() for c in cap])
        steps.append({"loss": float(loss), "grad_norm": float(gn),
                      "params": {k: summarize(v) for k, v in head.state_dict().items()},
                      "out_bias": head.model[-1].bias.detach().numpy().tolist()})
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(HERE, "head_step_masks.npz"), m1_0=masks[0][0], m2_0=masks[0][1],
                        m1_1=masks[1][0], m2_1=masks[1][1])
    json.dump({"steps": steps, "x_seed": 21}, open(os.path.join(HERE, "head_step.json"), "w"))


def gen_ewc():
    """Reference EWC: Fisher with recorded batch order + sampled labels; penalty after p += 0.1
    (tests/test_ewc.py:128-153); and the as-wired construct of _train_new_classes (== 0.0)."""
    head = ref.AdaptiveHead(768, 3, [768, 384])
    X = torch.from_numpy(synth.synth_unit_rows(20, 768, 31))
    y = torch.arange(20) % 3

    class Rec(torch.utils.data.Dataset):
        def __init__(self): self.order = []
        def __len__(self): return 20
        def __getitem__(self, i): self.order.append(int(i)); return X[i], y[i]

    ds = Rec()
    sampled = []
    orig = torch.multinomial
    torch.multinomial = lambda *a, **k: (sampled.append(orig(*a, **k)) or sampled[-1])
    try:
        torch.manual_seed(5)
        ewc = RefEWC(head, ds, device="cpu", ewc_lambda=100.0)
    finally:
        torch.multinomial = orig
    zero = float(ewc.ewc_loss(batch_size=32))
    with torch.no_grad():
        for p in head.parameters():
            p += 0.1
    out = {"order": ds.order, "sampled": sampled[0].squeeze(-1).tolist(), "loss_unperturbed": zero,
           "fisher": {k: summarize(v) for k, v in ewc.fisher_info.items()},
           "loss_p01": float(ewc.ewc_loss()), "loss_p01_b32": float(ewc.ewc_loss(batch_size=32))}
    # as wired in classifier.py:171,298-303,339: EWC on a deepcopy, penalty evaluated while another head trains
    import copy
    live = ref.AdaptiveHead(768, 3, [768, 384])
    old = copy.deepcopy(live)
    live.update_num_classes(4)
    e2 = RefEWC(old, torch.utils.data.TensorDataset(X[:10], y[:10]), device="cpu", ewc_lambda=5.0)
    with torch.no_grad():
        for p in live.parameters():
            p += 0.3                                   # training moves the live head ...
    out["as_wired_penalty"] = float(e2.ewc_loss(batch_size=32))   # ... but the penalty looks at `old` only
    json.dump(out, open(os.path.join(HERE, "ewc.json"), "w"))


def gen_multilabel():
    """Reference MultiLabelAdaptiveHead: BCE steps (multilabel.py:361-384) and one CrossEntropy-on-sigmoid
    step (what classifier.py:337-351 does to a sigmoid head), dropout masks captured; plus the threshold /
    min / max decision logic of predict_multilabel driven with fixed probabilities."""
    from adaptive_classifier.multilabel import MultiLabelAdaptiveClassifier, MultiLabelAdaptiveHead
    D, C, B = 768, 5, 32
    torch.manual_seed(7)
    head = MultiLabelAdaptiveHead(D, C, [D, D // 2])
    init = {k: summarize(v) for k, v in head.state_dict().items()}
    head.train()
    opt = torch.optim.AdamW(head.parameters(), lr=0.001, weight_decay=0.01)
    X = torch.from_numpy(synth.synth_unit_rows(B, D, 41))
    T = torch.from_numpy(((np.arange(B)[:, None] * 3 + np.arange(C)[None, :] * 5) % 7 < 2).astype(np.float32))
    y = torch.from_numpy((np.arange(B) * 3 % C).astype(np.int64))
    cap, steps, masks = [], [], {}
    hooks = [m.register_forward_hook(lambda mod, inp, out: cap.append(((out != 0) | (inp[0] == 0)).to(torch.uint8)))
             for m in head.model if isinstance(m, torch.nn.Dropout)]
    torch.manual_seed(321)
    for s, kind in enumerate(["bce", "bce", "ce_sigmoid"]):
        cap.clear()
        opt.zero_grad()
        out = head(X)
        loss = torch.nn.BCELoss()(out, T) if kind == "bce" else torch.nn.CrossEntropyLoss()(out, y)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(head.parameters(), max_norm=1.0)
        opt.step()
        masks[f"m1_{s}"], masks[f"m2_{s}"] = cap[0].numpy().copy(), cap[1].numpy().copy()
        steps.append({"kind": kind, "loss": float(loss.detach()), "grad_norm": float(gn),
                      "params": {k: summarize(v) for k, v in head.state_dict().items()}})
    for h in hooks:
        h.remove()
    head.eval()
    with torch.no_grad():
        probs_after = head(X[:4]).numpy().tolist()
    np.savez_compressed(os.path.join(HERE, "multilabel_masks.npz"), **masks)

    # decision logic with fixed probabilities
    clf = MultiLabelAdaptiveClassifier.__new__(MultiLabelAdaptiveClassifier)
    labels = [f"l{i}" for i in range(6)]
    clf.label_to_id = {l: i for i, l in enumerate(labels)}; clf.id_to_label = {i: l for i, l in enumerate(labels)}
    clf.default_threshold = 0.5; clf.device = "cpu"
    table = {"a": [0.9, 0.45, 0.31, 0.05, 0.29, 0.6], "b": [0.1, 0.2, 0.05, 0.02, 0.01, 0.03],
             "c": [0.31, 0.31, 0.8, 0.7, 0.65, 0.62]}
    cur = {}
    clf._get_embeddings = lambda ts: [torch.zeros(4)]
    class FakeHead:
        def eval(self): return self
        def __call__(self, x): return torch.tensor([cur["p"]])
    clf.adaptive_head = FakeHead()
    decisions = []
    for (mn, mx, lt) in [(1, None, {}), (2, 3, {}), (1, 2, {"l1": 0.15, "l0": 0.95}), (3, None, {"l5": 0.0})]:
        clf.min_predictions, clf.max_predictions, clf.label_thresholds = mn, mx, lt
        for name, p in table.items():
            cur["p"] = p
            for thr, ml in [(None, None), (0.3, None), (None, 2)]:
                decisions.append({"min": mn, "max": mx, "label_thresholds": lt, "probs": p, "threshold": thr,
                                  "max_labels": ml, "out": clf.predict_multilabel(name, threshold=thr, max_labels=ml)})
    json.dump({"init": init, "steps": steps, "x_seed": 41, "probs_after": probs_after, "decisions": decisions,
               "adaptive_thresholds": {str(n): clf._get_adaptive_threshold(n) for n in (1, 2, 3, 5, 6, 10, 11, 20, 21, 50)}},
              open(os.path.join(HERE, "multilabel.json"), "w"))


def _ref_classifier(cls, mem, labels, head, history, name="stub-encoder"):
    """A reference classifier object without the HF download: everything the predict / save paths touch."""
    clf = cls.__new__(cls)
    clf.config = ref.ModelConfig(); clf.device = "cpu"; clf.memory = mem; clf.embedding_dim = mem.embedding_dim
    clf.label_to_id = {l: i for i, l in enumerate(labels)}; clf.id_to_label = {i: l for i, l in enumerate(labels)}
    clf.training_history = dict(history); clf.train_steps = 3
    clf.strategic_cost_function = None                     # strategic_mode (a property of config + this) -> False
    clf.adaptive_head = head

    class _Cfg:
        _name_or_path = name
        hidden_size = mem.embedding_dim

    class _Model:
        config = _Cfg()
    clf.model = _Model()
    return clf


def _queries(n, D, seed, cent_seed, C):
    Q = synth.synth_unit_rows(n, D, seed)
    cent = synth.synth_unit_rows(C, D, cent_seed)
    return np.stack([(q * 0.5 + cent[i % C]) / np.linalg.norm(q * 0.5 + cent[i % C]) for i, q in enumerate(Q)]).astype(np.float32)


def gen_multilabel_predict():
    """The reference's INHERITED predict_batch / _predict_regular on a multi-label classifier: they call
    self.adaptive_head(x) -- sigmoid outputs for MultiLabelAdaptiveHead -- and softmax THOSE (classifier.py:1342-1345,
    :432-435), i.e. the blend sees softmax(sigmoid(z)).  Also `predict()` falling through to super().predict
    (multilabel.py:228-240) when nothing passes the threshold and min_predictions = 0."""
    from adaptive_classifier.multilabel import MultiLabelAdaptiveClassifier, MultiLabelAdaptiveHead
    mem, labels = build_memory()
    torch.manual_seed(7)
    head = MultiLabelAdaptiveHead(768, 4, [768, 384]).eval()
    clf = _ref_classifier(MultiLabelAdaptiveClassifier, mem, labels, head, {"c0": 25, "c1": 5, "c2": 25, "c3": 9})
    clf.default_threshold = 0.5; clf.min_predictions = 1; clf.max_predictions = None; clf.label_thresholds = {}
    Q = _queries(8, 768, 78, 11, 4)
    texts = [f"q{i}" for i in range(8)]
    table = {t: torch.from_numpy(q) for t, q in zip(texts, Q)}
    clf._get_embeddings = lambda ts: [table[t] for t in ts]
    out = {"q_seed": 78, "head_seed": 7, "head_init": {k: summarize(v) for k, v in head.state_dict().items()}}
    out["predict_batch"] = {f"k{k}": clf.predict_batch(texts, k=k) for k in (1, 2, 5)}
    out["predict_regular"] = {f"k{k}": [ref.AdaptiveClassifier._predict_regular(clf, t, k) for t in texts] for k in (1, 3, 5)}
    clf.min_predictions = 0; clf.default_threshold = 5.0           # nothing can pass -> predict() uses super().predict
    out["predict_fallthrough_k3"] = [clf.predict(t, k=3) for t in texts]
    json.dump(out, open(os.path.join(HERE, "multilabel_predict.json"), "w"))


def gen_saved_dirs():
    """N1: directories in the reference's on-disk formats, WRITTEN BY THE REFERENCE, plus what the reference predicts
    from the restored state.
      ref_saved_d64/         written by the reference's own _save_pretrained (config.json, examples.json with the
                             k-means representatives, model.safetensors; classifier.py:524-628) from a 4-class D=64
                             classifier whose head went through the reference's _train_adaptive_head
      adaptive_router_legacy/  the reference's shipped scripts/adaptive_router fixture (older layout: examples inline
                             in config.json + tensors.safetensors), byte-for-byte
    expected.json in each: predict_batch / _predict_regular outputs of reference objects restored the way
    _from_pretrained restores them (classifier.py:864-913)."""
    import shutil
    from safetensors.torch import load_file
    # ---- (a) reference-written directory, D = 64 (small files)
    D = 64
    np.random.seed(0)                                    # k-means inside select_representative_examples
    torch.manual_seed(11)
    mem, labels = build_memory(C=4, per_class=12, D=D, seed=50)
    head = ref.AdaptiveHead(D, 4, [D, D // 2])
    clf = _ref_classifier(ref.AdaptiveClassifier, mem, labels, head, {"c0": 12, "c1": 12, "c2": 4, "c3": 12}, name="stub-encoder-d64")
    clf._train_adaptive_head(epochs=3)                   # the reference's own loop (classifier.py:1428-1522), torch CPU
    clf.adaptive_head.eval()
    out_dir = os.path.join(HERE, "ref_saved_d64")
    shutil.rmtree(out_dir, ignore_errors=True)
    clf._save_pretrained(out_dir, include_onnx=False)
    os.remove(os.path.join(out_dir, "README.md"))        # model card: not part of the format under test
    Q = _queries(8, D, 79, 51, 4)
    texts = [f"q{i}" for i in range(8)]

    def restored(save_dir, tensors_name, examples, D_, head_dims):
        """What _from_pretrained does after constructing the classifier (classifier.py:864-913)."""
        cfg = json.load(open(os.path.join(save_dir, "config.json")))
        tensors = load_file(os.path.join(save_dir, tensors_name))
        m = ref.PrototypeMemory(D_, config=ref.ModelConfig(cfg.get("config")))
        for label, exs in examples.items():
            m.examples[label] = [ref.Example.from_dict(e) for e in exs]
        for label in cfg["label_to_id"]:
            if f"prototype_{label}" in tensors:
                m.prototypes[label] = tensors[f"prototype_{label}"]
        m._restore_from_save()
        h = ref.AdaptiveHead(D_, len(cfg["label_to_id"]), head_dims)
        h.load_state_dict({k.replace("adaptive_head_", ""): v for k, v in tensors.items() if k.startswith("adaptive_head_")})
        hist = cfg.get("training_history") or {l: len(e) * 20 for l, e in examples.items()}
        c = _ref_classifier(ref.AdaptiveClassifier, m, [cfg["id_to_label"][str(i)] for i in range(len(cfg["id_to_label"]))],
                            h.eval(), hist, name=cfg["model_name"])
        c.label_to_id = cfg["label_to_id"]
        return c

    r = restored(out_dir, "model.safetensors", json.load(open(os.path.join(out_dir, "examples.json"))), D, [D, D // 2])
    table = {t: torch.from_numpy(q) for t, q in zip(texts, Q)}
    r._get_embeddings = lambda ts: [table[t] for t in ts]
    exp = {"q_seed": 79, "cent_seed": 51, "D": D,
           "predict_batch": {f"k{k}": r.predict_batch(texts, k=k) for k in (1, 2, 4)},
           "predict": {f"k{k}": [r._predict_regular(t, k) for t in texts] for k in (1, 3)},
           "training_history": r.training_history,
           "stats": {"examples_per_class": {l: len(e) for l, e in r.memory.examples.items()}}}
    json.dump(exp, open(os.path.join(out_dir, "expected.json"), "w"))
    # ---- (b) the shipped legacy-layout fixture
    src = "/root/reference/scripts/adaptive_router"
    dst = os.path.join(HERE, "adaptive_router_legacy")
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    for f in ("config.json", "tensors.safetensors"):
        shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
        os.chmod(os.path.join(dst, f), 0o644)
    cfg = json.load(open(os.path.join(dst, "config.json")))
    r = restored(dst, "tensors.safetensors", cfg["examples"], 768, [768, 384])
    emb = [np.asarray(e["embedding"], np.float32) for l in ("HIGH", "LOW") for e in cfg["examples"][l]]
    texts = [f"r{i}" for i in range(len(emb))]
    table = {t: torch.from_numpy(q) for t, q in zip(texts, emb)}
    r._get_embeddings = lambda ts: [table[t] for t in ts]
    exp = {"predict_batch": {f"k{k}": r.predict_batch(texts, k=k) for k in (1, 2)},
           "predict": {f"k{k}": [r._predict_regular(t, k) for t in texts] for k in (1, 2)},
           "training_history": r.training_history}
    json.dump(exp, open(os.path.join(dst, "expected.json"), "w"))


if __name__ == "__main__":
    gen_router(); gen_knn(); gen_memory_and_blend(); gen_head_step(); gen_ewc(); gen_multilabel()
    gen_multilabel_predict(); gen_saved_dirs()
    print("golden fixtures written to", HERE)
    for dp, _, files in sorted(os.walk(HERE)):
        for f in sorted(files):
            print(f"  {os.path.relpath(os.path.join(dp, f), HERE):44s} {os.path.getsize(os.path.join(dp, f)):8d} B")

