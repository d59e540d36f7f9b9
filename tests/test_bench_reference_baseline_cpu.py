"""CPU suite: bench.py's `cpu_baseline` leg with kind "reference" -- the unmodified reference's predict_batch under the Hub stand-in
and the one-thread-per-query faiss stand-in -- runs end to end on a tiny instance of the workload (the driver's bench line must not
die in this leg; on a box where oracle/_ref is not staged it returns None and the port stands in)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_cpu_baseline_runs_on_a_small_instance(monkeypatch):
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_ac", "classifier.py")):
        pytest.skip("oracle/_ref not staged (run `python oracle/stage_ref.py` where /root/reference exists)")
    sys.path.insert(0, ROOT)
    import bench
    from oracle import hub_standin, synth
    monkeypatch.setattr(bench, "BATCH", 6)
    # a 2-layer stand-in instead of bert-base keeps this under ten seconds; the code path is the same
    monkeypatch.setitem(hub_standin.ARCHITECTURES, "bert-base-uncased",
                        ("bert", {"num_hidden_layers": 2, "intermediate_size": 512}, True))
    ids, types, mask = bench.synthetic_tokens("cpu", 0)
    ids, mask = ids[:6], mask[:6]
    P = torch.from_numpy(synth.synth_unit_rows(3000, bench.DIM, 1))
    before = (sys.modules.get("faiss"), torch.get_num_threads())
    r = bench.cpu_baseline_reference(ids, mask, P, seconds_hint=1.0)
    assert r["kind"] == "reference" and r["value"] > 0 and r["cores"] >= 1
    assert "unmodified reference" in r["sample"] and "NOT real faiss" in r["faiss"]
    assert sys.modules.get("faiss") is before[0]                    # the faiss stand-in does not outlive the leg
    import transformers
    assert "hub_standin" not in getattr(transformers.AutoModel.from_pretrained, "__module__", "")   # nor does the Hub stand-in
    # a failing extra never costs the line
    out = bench._guarded("x", lambda: 1 / 0)
    assert "error" in out and "ZeroDivisionError" in out["error"]


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_ac", "classifier.py")), reason="oracle/_ref not staged")
def test_add_examples_reference_baseline_runs_the_staged_reference_loop():
    """configs[3]'s cpu_baseline (kind "reference"): the unmodified reference's add_examples on pre-computed embeddings, memory at
    the cap -- here with a cap of 40 per class so the CPU suite stays short; the stand-ins are removed again afterwards."""
    import sys
    import torch
    import bench
    g = torch.Generator().manual_seed(0)
    E = torch.nn.functional.normalize(torch.randn(400, 768, generator=g), dim=1)
    out = bench.cpu_baseline_add_examples_reference(E, 4, cap=40, seconds_hint=1.0)
    assert out["kind"] == "reference" and out["unit"] == "examples/s" and out["value"] > 0 and out["cores"] >= 1
    assert out["stored_examples"] == 160 and out["calls_timed"] >= 1
    assert "ref_ac" in sys.modules and "oracle/_ref" in (sys.modules["ref_ac"].__file__ or "")
