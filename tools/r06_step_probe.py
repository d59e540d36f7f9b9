"""Round 6: where a timed predict() step (bench configs[1]) spends its wall time on the host side.  Phases of predict_tokens, by
perf_counter: encoder enqueue | kNN + head enqueue | blend enqueue | wait for the device + the packed D2H | unpack into the reference's lists;
next to the device span of the same step (HIP events around the first and last launch).  Under rocprofv3 --kernel-trace the same run gives
the launch sequence of one step (tools/r06_step_seq.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import numpy as np
import torch
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
clf, hf = bench.make_classifier(dev, 0, 1)
ids, types, mask = bench.synthetic_tokens(dev, 0)
K = bench.KNN_K
for _ in range(5): bench.predict_step(clf, ids, types, mask)
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for _ in range(n): bench.predict_step(clf, ids, types, mask)
torch.cuda.synchronize()
whole = (time.perf_counter() - t0) / n * 1e3
ph = []
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
pc = time.perf_counter
for _ in range(n):
    a = pc(); ev[0].record()
    emb = clf._encode_tokens(ids, types, mask, verify=False)
    b = pc()
    S, I, P = clf._device_stage(emb, K)
    c = pc()
    out, layout = clf._blend_device(S, I, P, K, False)
    ev[1].record()
    d = pc()
    host = out.cpu().numpy()
    e = pc()
    res = clf._unpack(host, layout, K, False)
    f = pc()
    ph.append([b - a, c - b, d - c, e - d, f - e, f - a, ev[0].elapsed_time(ev[1]) * 1e-3])
ph = np.median(np.array(ph), axis=0) * 1e3
names = ["encoder_enqueue", "knn_head_enqueue", "blend_enqueue", "wait_and_d2h", "unpack", "step_wall", "device_span"]
print("ms per step (predict_tokens, %d steps): %.4f" % (n, whole))
print("phases_ms " + "  ".join("%s %.4f" % (k, v) for k, v in zip(names, ph)), flush=True)
