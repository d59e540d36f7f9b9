"""Scratch probe: ac_memory_add_prune kernel time alone vs the host work around it."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import numpy as np, torch
from adaptive_classifier import _native as nv
dev = torch.device("cuda:0"); lib = nv.lib()
n0, k, cap, D = 1000, 8, 1000, 768
rows = torch.nn.functional.normalize(torch.randn(n0 + k, D), dim=1)
d_rows = rows.to(dev); d_sum = rows[:n0].double().sum(0).to(dev)
d_alive = torch.empty(n0 + k, dtype=torch.uint8, device=dev); d_dist = torch.empty(n0 + k, dtype=torch.float64, device=dev)
d_drop = torch.empty(k, dtype=torch.int32, device=dev)
import ctypes
NJ = 4
arr = (nv.ac_prune_job * NJ)()
sums = [d_sum.clone() for _ in range(NJ)]
for i in range(NJ):
    arr[i].rows = d_rows.data_ptr(); arr[i].ld = D; arr[i].n_old, arr[i].n_new, arr[i].cap = n0, k, cap
    arr[i].sum = sums[i].data_ptr(); arr[i].alive = d_alive.data_ptr(); arr[i].dist = d_dist.data_ptr(); arr[i].dropped = d_drop.data_ptr()
d_jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
def run():
    nv.check(lib.ac_memory_add_prune(ctypes.cast(arr, ctypes.c_void_p), nv.ptr(d_jobs), NJ, D, nv.stream_ptr(dev)), "prune")
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"kernel, 4 class jobs in one launch (n0={n0}, k={k}, D={D}): {e0.elapsed_time(e1)/20*1e3:.0f} us per call = {e0.elapsed_time(e1)/20/k*1e3:.0f} us per step")
t0 = time.perf_counter()
for _ in range(20):
    x = rows.to(dev); torch.cuda.synchronize()
print(f"H2D of the class matrix: {(time.perf_counter()-t0)/20*1e3:.2f} ms")
t0 = time.perf_counter()
for _ in range(20):
    c = torch.cat([rows[:n0], rows[n0:]]); o = np.argsort(np.random.rand(n0)); g = c[torch.from_numpy(o)]
print(f"host cat + gather: {(time.perf_counter()-t0)/20*1e3:.2f} ms")
lst = [object() for _ in range(n0 + k)]
t0 = time.perf_counter()
for _ in range(20):
    l2 = [lst[i] for i in o]
print(f"list rebuild: {(time.perf_counter()-t0)/20*1e3:.2f} ms")
