"""A/B (round 4, VERDICT r03 item 5 i): the packed 256-text batch as ONE launch sequence against TWO token-balanced halves on
two HIP streams, so that one half's epilogues / ring fills / LayerNorm exchanges run under the other half's k-loops.
Tile tables for the halves: the built-in per-shape choice at half the rows, and the full batch's tiles forced on the halves.
Interleaved rounds, median encode time; results must agree."""
import os as _os; _os.environ.setdefault("AC_TEST_HOOKS", "1")  # (the process-wide switches used below are test hooks)
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import torch
from adaptive_classifier import _native as nv
from adaptive_classifier.encoder import HipBertEncoder
from transformers import BertConfig, BertModel
dev = torch.device("cuda:0")
torch.manual_seed(0)
hf = BertModel(BertConfig(), add_pooling_layer=False).eval()
encs = [HipBertEncoder(hf, device=dev) for _ in range(2)]
B, S = 256, 32
g = torch.Generator().manual_seed(1234)
ids = torch.randint(1000, 30000, (B, S), generator=g); ids[:, 0] = 101
lens = torch.randint(8, S + 1, (B,), generator=g); lens[0] = S
mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
ids = (ids * mask)
# token-balanced halves: texts sorted by length, dealt alternately
order = torch.argsort(lens, stable=True)
halves = [order[0::2], order[1::2]]
print("tokens:", int(lens.sum()), "halves:", [int(lens[h].sum()) for h in halves])
ids, mask = ids.to(dev), mask.to(dev)
hid = [(ids[h.to(dev)].contiguous(), mask[h.to(dev)].contiguous()) for h in halves]
inv = torch.empty(B, dtype=torch.long); inv[torch.cat(halves)] = torch.arange(B)
inv = inv.to(dev)
streams = [torch.cuda.Stream(dev) for _ in range(2)]


def one():
    return encs[0].encode_cls(ids, None, mask, verify=False)


def two():
    outs = []
    cur = torch.cuda.current_stream(dev)
    for h in range(2):
        streams[h].wait_stream(cur)
        with torch.cuda.stream(streams[h]):
            outs.append(encs[h].encode_cls(hid[h][0], None, hid[h][1], verify=False))
    for h in range(2):
        cur.wait_stream(streams[h])
    return torch.cat(outs)[inv]


FULL = "2304x768=322432;768x768=124262;3072x768=244232;768x3072=124262"
variants = [("one stream, built-in tiles", one, None), ("two streams, built-in tiles per half", two, None),
            ("two streams, the full batch's tiles", two, FULL), ("two streams, no LayerNorm fusion", two, "nofuse")]
ref = one(); torch.cuda.synchronize()
times = {n: [] for n, _, _ in variants}
for rnd in range(5):
    for name, f, table in variants:
        nv.check(nv.lib().ac_gemm_set_pipe_table(table.encode() if table and table != "nofuse" else None), "table")
        nv.check(nv.lib().ac_gemm_set_ln_fusion(0 if table == "nofuse" else 1), "ln")
        for _ in range(3): out = f()
        torch.cuda.synchronize()
        assert (out - ref).abs().max().item() < 1e-5, (name, (out - ref).abs().max().item())
        t0 = time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / 20 * 1e3)
nv.check(nv.lib().ac_gemm_set_pipe_table(None), "table"); nv.check(nv.lib().ac_gemm_set_ln_fusion(1), "ln")
for name, _, _ in variants:
    t = sorted(times[name])
    print(f"{name:42s} median {t[len(t) // 2]:.3f} ms   min {t[0]:.3f} ms")
