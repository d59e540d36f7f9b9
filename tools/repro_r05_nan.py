"""Scratch repro (round 5): reference tests/test_classifier.py::test_prediction gave NaN scores on the product under the Hub stand-in."""
import os as _os; _os.environ.setdefault("AC_TEST_HOOKS", "1")  # (the process-wide switches used below are test hooks)
import logging, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
logging.basicConfig(level=logging.WARNING)
import numpy as np, torch
from oracle import hub_standin
hub_standin.install()
from adaptive_classifier import AdaptiveClassifier, _native as nv
texts = ["This is amazing", "Terrible experience", "Just okay", "Love it", "Hate it"]
labels = ["positive", "negative", "neutral", "positive", "negative"]
def poison():
    """Fill torch's caching allocator with NaN-patterned free blocks of many sizes: later torch.empty() workspaces then hold
    NaNs instead of the zeros a fresh process gets from the driver."""
    keep = []
    for sz in [256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26, 1 << 28]:
        for _ in range(8 if sz < (1 << 24) else 2):
            keep.append(torch.full((sz // 4,), float("nan"), device="cuda"))
    torch.cuda.synchronize()
    del keep
if "--poison" in sys.argv:
    poison()
clf = AdaptiveClassifier("bert-base-uncased")
if "--poison" in sys.argv:
    poison()
emb = torch.stack(clf._get_embeddings(texts))
print("emb finite", bool(torch.isfinite(emb).all()), emb.norm(dim=1))
clf.add_examples(texts, labels)
print("train info", clf.last_train_info)
fp = clf.adaptive_head.flat_params()
print("head finite", bool(torch.isfinite(fp).all()), "nan count", int((~torch.isfinite(fp)).sum()), "of", fp.numel())
print("protos finite", {l: bool(torch.isfinite(p).all()) for l, p in clf.memory.prototypes.items()})
e1 = clf._embed_device(["This is fantastic"])
print("query emb finite", bool(torch.isfinite(e1).all()), "one_launch", clf.model.last_one_launch)
e2 = clf._embed_device(["This is fantastic"], force_layered=True)
print("layered emb finite", bool(torch.isfinite(e2).all()), float((e1 - e2).abs().max()))
print("predict", clf.predict("This is fantastic"))
S, Cid, P = clf._device_stage(e2, 3)
print("S", S, "Cid", Cid, "P", P)
# head training alone, persistent vs stepwise, on the same data
for mask in (3, 2):
    nv.lib().ac_set_persistent_kernels(mask)
    c2 = AdaptiveClassifier("bert-base-uncased", encoder=clf.model, tokenizer=clf.tokenizer)
    c2.add_examples(texts, labels)
    fp = c2.adaptive_head.flat_params()
    print("persistent mask", mask, c2.last_train_info, "head finite", bool(torch.isfinite(fp).all()))
nv.lib().ac_set_persistent_kernels(3)
