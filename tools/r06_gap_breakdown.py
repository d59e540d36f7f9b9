"""Round 6: the interpreter's 41 us between ac_predict_post's return and the next ac_bert_encode_cls_unpad, by perf_counter stamps at the
function boundaries on the way (wrappers add ~0.3 us each)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import numpy as np, torch
import bench
from adaptive_classifier import _native as nv
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
clf, hf = bench.make_classifier(dev, 0, 1)
ids, types, mask = bench.synthetic_tokens(dev, 0)
pc = time.perf_counter
T = {}
def mark(k): T.setdefault(k, []).append(pc())
L = nv.lib()
post, unpad = L.ac_predict_post, L.ac_bert_encode_cls_unpad
class Wrap:
    def __init__(self, lib): self._l = lib
    def __getattr__(self, n): return getattr(self._l, n)
    def ac_predict_post(self, *a):
        r = post(*a); mark("post_ret"); return r
    def ac_bert_encode_cls_unpad(self, *a):
        mark("unpad_in"); return unpad(*a)
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        mark(key + "_in"); r = f(*a, **k); mark(key + "_out"); return r
    setattr(obj, name, g)
for _ in range(5): bench.predict_step(clf, ids, types, mask)
nv._lib = Wrap(L)
wrap(clf, "_unpack", "unpack"); wrap(clf, "_post_launch", "postlaunch"); wrap(clf, "_finish_from_embeddings", "finish")
wrap(clf, "predict_tokens", "predict"); wrap(clf.model, "encode_cls", "encode"); wrap(clf.model, "_run_chunks", "chunks")
for _ in range(60): bench.predict_step(clf, ids, types, mask)
torch.cuda.synchronize()
seq = [("post_ret", 0), ("postlaunch_out", 0), ("unpack_in", 0), ("unpack_out", 0), ("finish_out", 0), ("predict_out", 0), ("predict_in", 1), ("encode_in", 1),
       ("chunks_in", 1), ("unpad_in", 1)]
A = {k: np.array(v) for k, v in T.items()}
n = min(len(v) for v in A.values()) - 1
prev = None
for k, sh in seq:
    t = A[k][sh:sh + n] if sh else A[k][:n]
    if prev is not None:
        print("%-16s -> %-16s %6.1f us" % (prev[0], k, np.median(t - prev[1]) * 1e6))
    prev = (k, t)
print("total %.1f us" % (np.median(A["unpad_in"][1:1 + n] - A["post_ret"][:n]) * 1e6))
