"""Scratch probe: encoder as one 256-batch vs two 128-batches on two streams (tail/ramp overlap)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import _native as nv
from adaptive_classifier.encoder import HipBertEncoder
from transformers import BertConfig, BertModel
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = HipBertEncoder(BertModel(BertConfig(), add_pooling_layer=False).eval(), device=dev)
B, S = 256, 32
ids = torch.randint(1000, 30000, (B, S), device=dev); ids[:, 0] = 101
mask = torch.ones_like(ids)
def run_parts(nparts, streams):
    h = B // nparts
    outs = []
    for p in range(nparts):
        st = streams[p]
        with torch.cuda.stream(st):
            need = enc.workspace_bytes(h, S)
            ws = wss[p]
            out = outbuf[p * h:(p + 1) * h]
            i = ids[p * h:(p + 1) * h]; m = mask[p * h:(p + 1) * h]
            nv.check(nv.lib().ac_bert_encode_cls(ctypes.byref(enc.ccfg), ctypes.byref(enc.weights), nv.ptr(i), None, nv.ptr(m),
                                                 h, S, nv.ptr(out), out.stride(0), nv.ptr(ws), ws.numel(), ctypes.c_void_p(st.cuda_stream)), "enc")
outbuf = torch.empty((B, 768), device=dev)
wss = [torch.empty(enc.workspace_bytes(B, S), dtype=torch.uint8, device=dev) for _ in range(4)]
main = torch.cuda.current_stream()
for nparts in (1, 2, 4, 1, 2, 4):
    streams = [main] if nparts == 1 else [torch.cuda.Stream() for _ in range(nparts)]
    for _ in range(3): run_parts(nparts, streams)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps): run_parts(nparts, streams)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"parts={nparts}: {dt*1e3:.2f} ms per 256-batch", flush=True)
ref = outbuf.clone()
run_parts(1, [main]); torch.cuda.synchronize()
print("max diff 1 vs split:", (outbuf - ref).abs().max().item())
