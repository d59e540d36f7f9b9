"""AC_GEMM_F16X2 (opt-in fp16x2 arithmetic of the encoder's token-row GEMMs): per-shape sweep of the ring configurations and
the encoder end to end next to bf16x3.

usage: f16x2_probe.py [--large] [--no-sweep]
  1. for the four GEMM shapes of a layer at the bench's packed row count: time every fp16x2 configuration (ac_linear_f16x2 behind
     ac_gemm_set_pipe_table_f16) and the bf16x3 default on the same shape;
  2. encode the bench batch under bf16x3 and fp16x2 (interleaved rounds), with the built-in table and with the sweep's winners;
     prints the max abs difference of the embeddings between the two arithmetics.
"""
import os as _os; _os.environ.setdefault("AC_TEST_HOOKS", "1")  # (the process-wide switches used below are test hooks)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import _native as nv
from adaptive_classifier.encoder import HipBertEncoder
from transformers import BertConfig, BertModel

args = sys.argv[1:]
large = "--large" in args
dev = torch.device("cuda:0")
lib = nv.lib()
torch.manual_seed(0)
cfg = BertConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096) if large else BertConfig()
H, I = cfg.hidden_size, cfg.intermediate_size
B, S = (1024, 32) if large else (256, 32)
g = torch.Generator().manual_seed(1234)
ids = torch.randint(1000, 30000, (B, S), generator=g); ids[:, 0] = 101
lens = torch.randint(8, S + 1, (B,), generator=g); lens[0] = S
mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
T = int(mask.sum())
CFGS = [222232, 124262, 224242, 234232, 322432, 244232, 244242, 234242, 224262, 124282]


def timed(fn, n=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


best = {}
if "--no-sweep" not in args:
    print(f"--- per-shape sweep, M = {T} token rows ({'bert-large' if large else 'bert-base'}) ---")
    for name, N, K, act, res, planes_out in (("QKV", 3 * H, H, 0, False, False), ("AO", H, H, 0, True, False),
                                             ("FFN1", I, H, 2, False, True), ("FFN2", H, I, 0, True, False)):
        A = torch.randn(T, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        bias = torch.randn(N, device=dev)
        R = torch.randn(T, N, device=dev) if res else None
        C = torch.empty(T, N, device=dev)
        Ap = torch.empty(2 * T * K, dtype=torch.int16, device=dev)
        Wp = torch.empty(2 * N * K, dtype=torch.int16, device=dev)
        Cp = torch.empty(3 * T * N, dtype=torch.int16, device=dev)
        nv.check(lib.ac_split_f16x2(nv.ptr(A), K, T, K, 6, nv.ptr(Ap), nv.stream_ptr(dev)), "split")
        nv.check(lib.ac_split_f16x2(nv.ptr(W), K, N, K, 10, nv.ptr(Wp), nv.stream_ptr(dev)), "split")
        A3 = torch.empty(3 * T * K, dtype=torch.int16, device=dev)
        W3 = torch.empty(3 * N * K, dtype=torch.int16, device=dev)
        nv.check(lib.ac_split_bf16x3(nv.ptr(A), K, T, K, nv.ptr(A3), nv.stream_ptr(dev)), "split3")
        nv.check(lib.ac_split_bf16x3(nv.ptr(W), K, N, K, nv.ptr(W3), nv.stream_ptr(dev)), "split3")

        def run16():
            nv.check(lib.ac_linear_f16x2(nv.ptr(Ap), nv.ptr(Wp), nv.ptr(bias), nv.ptr(R) if res else None, N,
                                         None if planes_out else nv.ptr(C), N, nv.ptr(Cp) if planes_out else None, T, N, K, act,
                                         nv.stream_ptr(dev)), "f16x2")

        def run3():
            nv.check(lib.ac_linear_bf16x3(nv.ptr(A), K, nv.ptr(A3), nv.ptr(W), K, nv.ptr(W3), nv.ptr(bias),
                                          nv.ptr(R) if res else None, N, None if planes_out else nv.ptr(C), N,
                                          nv.ptr(Cp) if planes_out else None, T, N, K, act, nv.stream_ptr(dev)), "bf16x3")
        lib.ac_gemm_set_arith(1)
        t3 = timed(run3)
        flop = 2.0 * T * N * K
        row3 = []
        for c in (222232, 124262, 224242, 234232, 322432, 244232):          # the bf16x3 configurations on the same shape
            nv.check(lib.ac_gemm_set_pipe_table(f"{N}x{K}={c}".encode()), "table")
            row3.append(f"{c} {timed(run3):6.1f}")
        lib.ac_gemm_set_pipe_table(None)
        row = [f"{name:5s} {T}x{N}x{K}: bf16x3 default {t3:7.1f} us ({6 * flop / t3 / 1e6 / 2500:.2f} of the bf16 pipe) [" + " ".join(row3) + "] | fp16x2:"]
        lib.ac_gemm_set_pipe_table_f16(None)
        tb = timed(run16)
        row.append(f"builtin {tb:6.1f}")
        res_t = {}
        for c in CFGS:
            nv.check(lib.ac_gemm_set_pipe_table_f16(f"{N}x{K}={c}".encode()), "table")
            res_t[c] = timed(run16)
            row.append(f"{c} {res_t[c]:6.1f}")
        c = min(res_t, key=res_t.get)
        best[(N, K)] = c
        row.append(f"| best {c} {res_t[c]:.1f} us = {3 * flop / res_t[c] / 1e6 / 2500:.2f} of the fp16 pipe, {t3 / res_t[c]:.2f}x bf16x3")
        print(" ".join(row), flush=True)
    lib.ac_gemm_set_pipe_table_f16(None)

hf = BertModel(cfg, add_pooling_layer=False).eval()
enc = HipBertEncoder(hf, device=dev).enable_f16x2()
idsd = (ids * mask).to(dev); maskd = mask.to(dev); types = torch.zeros_like(idsd)
tables = [("bf16x3", 1, None), ("f16x2 builtin", 2, None)]
if best:
    # the fused-LayerNorm launches need the 128 x 128 tile; the table only steers the N-wide GEMMs when fusion is off
    tables.append(("f16x2 swept", 2, ";".join(f"{N}x{K}={c}" for (N, K), c in best.items())))
    tables.append(("f16x2 swept QKV/FFN1 only", 2, ";".join(f"{N}x{K}={c}" for (N, K), c in best.items() if N != H)))
times = {n: [] for n, _, _ in tables}
outs = {}
for rnd in range(3):
    for name, mode, spec in tables:
        lib.ac_gemm_set_arith(mode)
        nv.check(lib.ac_gemm_set_pipe_table_f16(spec.encode() if spec else None), "table")
        times[name].append(timed(lambda: enc.encode_cls(idsd, types, maskd, verify=False), n=10, warm=2) / 1e3)
        outs[name] = enc.encode_cls(idsd, types, maskd).clone()
print(f"--- encoder, {B} ragged texts = {enc.last_tokens} token rows ---")
for name, _, spec in tables:
    t = sorted(times[name])
    d = (outs[name] - outs["bf16x3"]).abs().max().item()
    print(f"{name:28s} med {t[len(t) // 2]:.3f} ms  min {t[0]:.3f} ms   max |emb - bf16x3 emb| {d:.2e}   overflows {enc.f16x2_overflows}   [{spec}]")
lib.ac_gemm_set_arith(1)
