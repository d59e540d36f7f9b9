"""Scratch probe: BASELINE configs[3]-style continuous-learning loop (pre-computed 768-d embeddings fed in chunks
of 32, 4 classes, cap 1000/class), then a 5th class; reports examples/s and training steps/s."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from adaptive_classifier import AdaptiveClassifier
from adaptive_classifier.encoder import HipBertEncoder
from helpers import small_bert
from oracle import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
dev = torch.device("cuda:0")
from transformers import BertConfig, BertModel
torch.manual_seed(0)
enc = HipBertEncoder(BertModel(BertConfig(num_hidden_layers=1), add_pooling_layer=False).eval(), device=dev)   # 768-d, unused
clf = AdaptiveClassifier("x", device="cuda:0", encoder=enc, tokenizer=None)
C, D = 4, 768
cent = synth.synth_unit_rows(C + 1, D, 3); noise = synth.synth_unit_rows(n + 64, D, 4)
def emb(i, c):
    v = cent[c] + 0.5 * noise[i]; return torch.from_numpy((v / np.linalg.norm(v)).astype(np.float32))
E = [emb(i, i % C) for i in range(n)]
t0 = time.perf_counter(); steps = 0; tmem = 0.0
_orig = clf.memory.add_examples_batch
def _timed(*a, **k):
    global tmem
    t = time.perf_counter(); r = _orig(*a, **k); torch.cuda.synchronize(); tmem += time.perf_counter() - t; return r
clf.memory.add_examples_batch = _timed
_T = {}
def _wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize(); _T[name] = _T.get(name, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
for nm in ("_run_epochs", "_train_adaptive_head"): _wrap(clf, nm)
_wrap(clf.memory, "_rebuild_index")
from adaptive_classifier import training as _tr
_wrap(_tr.HeadTrainer, "fused_epoch")
for s in range(0, n, 32):
    idx = range(s, min(n, s + 32))
    clf.add_embeddings([f"t{i}" for i in idx], [E[i] for i in idx], [f"c{i % C}" for i in idx])
    steps += clf.last_train_info["steps"]
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"memory bookkeeping: {tmem:.3f} s = {tmem/n*1e3:.3f} ms per example")
print("phase seconds (synchronised):", {k: round(v, 3) for k, v in _T.items()})
print(f"{n} examples in chunks of 32: {dt:.2f} s = {n/dt:.0f} examples/s; {steps} training steps = {steps/dt:.0f} steps/s overall; "
      f"stored {clf.get_memory_stats()['total_examples']}")
np.random.seed(0)
t0 = time.perf_counter()
clf.add_embeddings([f"n{i}" for i in range(32)], [emb(n + i, C) for i in range(32)], ["znew"] * 32)
torch.cuda.synchronize()
print(f"new class (EWC path, as-wired): {time.perf_counter()-t0:.3f} s, {clf.last_train_info}")
