"""Round 6: the host's share of the gap between two timed predict() steps.  Timestamps (perf_counter) at the return of ac_predict_post
(the packed result is readable) and at the entry of the next step's ac_bert_encode_cls_unpad; how long that call then waits for the
packing kernel's report; and a cProfile of the interpreter between the two."""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import numpy as np
import torch
import bench
from adaptive_classifier import _native as nv
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
clf, hf = bench.make_classifier(dev, 0, 1)
ids, types, mask = bench.synthetic_tokens(dev, 0)
L = nv.lib()
post, unpad = L.ac_predict_post, L.ac_bert_encode_cls_unpad
T = {"post_ret": [], "unpad_in": [], "unpad_out": [], "wait": []}
pc = time.perf_counter
class Wrap:
    def __init__(self, lib): self._l = lib
    def __getattr__(self, n): return getattr(self._l, n)
    def ac_predict_post(self, *a):
        r = post(*a); T["post_ret"].append(pc()); return r
    def ac_bert_encode_cls_unpad(self, *a):
        T["unpad_in"].append(pc()); r = unpad(*a); T["unpad_out"].append(pc()); T["wait"].append(L.ac_bert_unpad_last_wait_ns()); return r
for _ in range(5): bench.predict_step(clf, ids, types, mask)
nv._lib = Wrap(L)
n = 60
t0 = pc()
for _ in range(n): bench.predict_step(clf, ids, types, mask)
torch.cuda.synchronize()
wall = (pc() - t0) / n * 1e3
gap = np.array(T["unpad_in"][1:]) - np.array(T["post_ret"][:-1])
print("ms per step %.4f   host gap post-return -> unpad-entry: median %.1f us (min %.1f)   unpad call: %.1f us, of which waiting for the report %.1f us"
      % (wall, np.median(gap) * 1e6, gap.min() * 1e6, np.median(np.array(T["unpad_out"]) - np.array(T["unpad_in"])) * 1e6, np.median(T["wait"]) / 1e3))
nv._lib = L
pr = cProfile.Profile()
pr.enable()
for _ in range(200): bench.predict_step(clf, ids, types, mask)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
