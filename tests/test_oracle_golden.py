"""CPU suite (no GPU): the oracle is pinned against fixtures produced by the reference itself
(tests/golden/gen_golden.py), and the C oracle agrees with the numpy oracle."""
import json
import os

import numpy as np
import torch

from oracle import c_oracle, head_oracle, knn_oracle, synth

G = os.path.join(os.path.dirname(__file__), "golden")


def test_knn_oracle_matches_reference_shim_cases():
    cases = json.load(open(os.path.join(G, "knn_cases.json")))
    for name, c in cases.items():
        if name == "dups":
            P = np.concatenate([synth.synth_unit_rows(10, 768, 9)] * 3)
            Q, k = P[:2], 6
        else:
            P = synth.synth_unit_rows(c["N"], c["D"], c["seed"])
            Q = synth.synth_unit_rows(c["nq"], c["D"], c["seed"] + 100)
            k = c["k"]
        for fn in (knn_oracle.knn_l2_topk, c_oracle.knn_l2_topk):
            D, I = fn(P, Q, k)
            assert np.array_equal(I, np.asarray(c["I"])), (name, fn.__module__)
            assert np.allclose(D, np.asarray(c["D_out"], np.float32), rtol=2e-7, atol=0)
    I = np.asarray(cases["dups"]["I"])
    assert (I[0, :3] == [0, 10, 20]).all()          # duplicates come back lowest id first


def test_router_fixture_distances_and_scores():
    """The reference's own saved classifier (scripts/adaptive_router): distances and
    softmax(exp(-d)) scores through reference memory.py vs the oracle restatement."""
    f = np.load(os.path.join(G, "router_fixture.npz"))
    D, I = knn_oracle.knn_l2_topk(f["protos"], f["emb"], 2)
    assert np.array_equal(I, f["order"])
    assert np.allclose(D, np.take_along_axis(f["dist"], f["order"], 1).astype(np.float32), rtol=1e-6)
    assert np.allclose(knn_oracle.proto_scores(D, I), f["scores"], atol=1e-7)


def test_c_oracle_equals_numpy_oracle_random():
    P = synth.synth_unit_rows(3000, 384, 4)
    Q = synth.synth_unit_rows(11, 384, 5)
    D1, I1 = knn_oracle.knn_l2_topk(P, Q, 17, row_offset=5)
    D2, I2 = c_oracle.knn_l2_topk(P, Q, 17, row_offset=5)
    assert np.array_equal(I1, I2) and np.allclose(D1, D2, rtol=2e-7)
    D3, I3 = c_oracle.knn_l2_topk(P[:4], Q, 9)                 # k > N pads like faiss
    assert (I3[:, 4:] == -1).all() and (D3[:, 4:] == np.finfo(np.float32).max).all()


def test_synth_generator_known_values():
    x = synth.synth_unit_rows(3, 768, 1)
    assert np.allclose(np.linalg.norm(x.astype(np.float64), axis=1), 1.0, atol=1e-6)
    # pinned bits (literal constants: any change of the generator -- which the HIP twin csrc/synth.hip must mirror bit for
    # bit -- breaks every seeded fixture and shows up here first)
    assert x.view(np.uint32)[0, :6].tolist() == [0xbc043e65, 0x3cc73598, 0x3c53bde0, 0xbcef93dd, 0xbc58f977, 0x3d114c70]
    assert x.view(np.uint32)[2, 765:768].tolist() == [0x3d0361b3, 0x3cf99a9e, 0xbc737864]
    assert synth.synth_unit_rows(2, 64, 3, row_offset=7).view(np.uint32)[1, :3].tolist() == [0x3ca2e667, 0x3c79beea, 0x3e3621ef]
    assert np.array_equal(synth.synth_unit_rows(5, 64, 3, row_offset=7), synth.synth_unit_rows(12, 64, 3)[7:])


def _check_summary(t, s, atol):
    f = t.detach().reshape(-1).double().numpy()
    idx = np.asarray(s["idx"]) % f.size
    assert f.size == s["n"]
    assert np.allclose(f[idx], s["vals"], atol=atol)
    assert abs(f.sum() - s["sum"]) < atol * f.size ** 0.5 * 10 + 1e-6


def test_head_oracle_reproduces_reference_training_steps():
    g = json.load(open(os.path.join(G, "head_step.json")))
    m = np.load(os.path.join(G, "head_step_masks.npz"))
    X = torch.from_numpy(synth.synth_unit_rows(32, 768, g["x_seed"]))
    y = torch.from_numpy((np.arange(32) * 7 % 4).astype(np.int64))
    head = head_oracle.make_head(768, 4).train()
    opt = torch.optim.AdamW(head.parameters(), lr=0.001, weight_decay=0.01, betas=(0.9, 0.999))
    for s, step in enumerate(g["steps"]):
        masks = [torch.from_numpy(m[f"m1_{s}"]), torch.from_numpy(m[f"m2_{s}"])]
        ce, _, gn = head_oracle.train_step(head, opt, X, y, masks=masks)
        assert abs(ce - step["loss"]) < 1e-6 and abs(gn - step["grad_norm"]) < 1e-6
        for (k, v) in zip(step["params"], [p for l in head_oracle.linears(head) for p in (l.weight, l.bias)]):
            _check_summary(v, step["params"][k], 1e-7)


def test_ewc_oracle_matches_reference():
    g = json.load(open(os.path.join(G, "ewc.json")))
    assert g["loss_unperturbed"] == 0.0 and g["as_wired_penalty"] == 0.0
    X = torch.from_numpy(synth.synth_unit_rows(20, 768, 31))
    head = head_oracle.make_head(768, 3)
    order = g["order"]
    f = head_oracle.fisher_from_labels(head, [X[order]], [torch.tensor(g["sampled"])])
    off = 0
    for (name, summ), p in zip(g["fisher"].items(), [p for l in head_oracle.linears(head) for p in (l.weight, l.bias)]):
        _check_summary(f[off: off + p.numel()], summ, 1e-9)
        off += p.numel()
    old = head_oracle.flat(head).clone()
    with torch.no_grad():
        for p in head.parameters():
            p += 0.1
    assert abs(float(head_oracle.ewc_penalty(head, f, old, 100.0)) - g["loss_p01"]) < 1e-6
    assert abs(float(head_oracle.ewc_penalty(head, f, old, 100.0 / 32)) - g["loss_p01_b32"]) < 1e-7


def test_multilabel_oracle_reproduces_reference_steps():
    """BCE and CE-on-sigmoid steps of the reference MultiLabelAdaptiveHead (tests/golden/multilabel.json)."""
    g = json.load(open(os.path.join(G, "multilabel.json")))
    m = np.load(os.path.join(G, "multilabel_masks.npz"))
    D, C, B = 768, 5, 32
    head = head_oracle.make_multilabel_head(D, C, seed=7)
    for (k, v) in zip(g["init"], [p for l in head_oracle.linears(head) for p in (l.weight, l.bias)]):
        _check_summary(v, g["init"][k], 1e-9)                 # same default init under the same seed
    head.train()
    opt = torch.optim.AdamW(head.parameters(), lr=0.001, weight_decay=0.01)
    X = torch.from_numpy(synth.synth_unit_rows(B, D, g["x_seed"]))
    T = torch.from_numpy(((np.arange(B)[:, None] * 3 + np.arange(C)[None, :] * 5) % 7 < 2).astype(np.float32))
    y = torch.from_numpy((np.arange(B) * 3 % C).astype(np.int64))
    for s, step in enumerate(g["steps"]):
        masks = [torch.from_numpy(m[f"m1_{s}"]), torch.from_numpy(m[f"m2_{s}"])]
        loss, gn = head_oracle.train_step_loss(head, opt, X, T if step["kind"] == "bce" else y, step["kind"], masks)
        assert abs(loss - step["loss"]) < 1e-6 and abs(gn - step["grad_norm"]) < 1e-6
        for (k, v) in zip(step["params"], [p for l in head_oracle.linears(head) for p in (l.weight, l.bias)]):
            _check_summary(v, step["params"][k], 1e-7)
