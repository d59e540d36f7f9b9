"""CPU suite: the generator bookkeeping of config["dropout_source"] = "torch_cpu" (AdaptiveClassifier._replay_epoch /
_replay_fisher_rng) against the states the UNMODIFIED reference left behind (tests/golden/e2e_train_*.json, written by
tests/golden/gen_e2e_train.py: sha256 of torch's global CPU generator and of numpy's after each add_examples call).

No GPU and no arithmetic here: the draws of a replayed training run depend only on shapes and step counts, so the product's
host-side sequence -- AdaptiveHead construction (models.py:49-66 seeds), two bernoulli_ masks per step, the np.random.choice
resampling of _train_new_classes (classifier.py:224-271), the Fisher pass's DataLoader + multinomial (ewc.py:60-64, :81) -- is run
with the step counts the reference logged and must end on the reference's generator states.  The GPU differential
(tests/test_e2e_reference_gpu.py) checks the same hashes with the real training in the loop."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _hashes():
    st = np.random.get_state()
    return {"torch_cpu": hashlib.sha256(torch.get_rng_state().numpy().tobytes()).hexdigest(),
            "numpy": hashlib.sha256(st[1].tobytes() + str(st[2:]).encode()).hexdigest()}


@pytest.mark.parametrize("case,dim", [("bert_mini", 128), ("bert_base", 768)])
def test_replay_draws_end_on_the_references_generator_states(case, dim):
    from adaptive_classifier.classifier import AdaptiveClassifier
    from adaptive_classifier.models import AdaptiveHead
    path = os.path.join(GOLD, "e2e_train_%s.json" % case)
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    ex = json.load(open(path))
    c1, c2 = ex["calls"]
    p = AdaptiveHead.DROPOUT_P
    H1, H2 = dim, dim // 2
    saved_t, saved_n = torch.get_rng_state(), np.random.get_state()
    try:
        torch.manual_seed(0)
        np.random.seed(0)
        torch.manual_seed(42)                                   # AdaptiveClassifier.__init__(seed=42), classifier.py:62
        # ---- call 1: _initialize_adaptive_head + _train_adaptive_head: one batch of min(32, n) rows per step
        head = AdaptiveHead(dim, len(c1["label_to_id"]), hidden_dims=[H1, H2])
        n1 = sum(c1["examples_per_class"].values())
        bs = min(32, n1)
        sizes = [min(bs, n1 - i) for i in range(0, n1, bs)]
        assert len(c1["step_losses"]) % len(sizes) == 0
        for _ in range(len(c1["step_losses"]) // len(sizes)):
            for B in sizes:
                torch.empty(B, H1).bernoulli_(1 - p)
                torch.empty(B, H2).bernoulli_(1 - p)
        assert _hashes() == c1["rng_after"]
        # ---- call 2: update_num_classes, the resampling, the (ineffective) Fisher pass, <= 15 epochs of batches of 32
        new_classes = set(c2["label_to_id"]) - set(c1["label_to_id"])
        head.update_num_classes(len(c2["label_to_id"]))
        per_class = c2["examples_per_class"]                    # the memory as _train_new_classes sees it (dict order = insertion order)
        min_examples = min(per_class.values())
        assert len(per_class) <= 20
        n2 = 0
        for label, n in per_class.items():
            weight = 2.0 if label in new_classes else min_examples / n
            num = max(min_examples, int(n * weight))
            np.random.choice(n, size=num, replace=num > n)
            n2 += num
        n_old = sum(min(5, n) for label, n in per_class.items() if label not in new_classes)
        AdaptiveClassifier._replay_fisher_rng(n_old, len(c1["label_to_id"]))
        sizes = [min(32, n2 - i) for i in range(0, n2, 32)]
        assert len(c2["step_losses"]) % len(sizes) == 0
        for _ in range(len(c2["step_losses"]) // len(sizes)):
            for B in sizes:
                torch.empty(B, H1).bernoulli_(1 - p)
                torch.empty(B, H2).bernoulli_(1 - p)
        assert _hashes() == c2["rng_after"]
    finally:
        torch.set_rng_state(saved_t)
        np.random.set_state(saved_n)
