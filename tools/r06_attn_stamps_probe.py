"""Where does the attention epilogue of the QKV GEMM spend its time?  Per-workgroup shader-clock stamps of gemm_pipe_nt<EPI_QKV_ATTN>
(measurement build tools/ab/libacamd_stamps.so = the tree compiled with -DAC_QKV_ATTN_STAMPS; AC_GEMM_STAMP_EPI=8 keeps the other
GEMMs from writing into the same buffer).  Slots: 0 start, 1 ring filled, 2 k-loop done, 3 end; per staging pass p: 4+3p staged,
5+3p wave 0's own sequences done, 6+3p every wave done."""
import os, sys
os.environ.setdefault("AC_TEST_HOOKS", "1"); os.environ["AC_GEMM_STAMP_EPI"] = "8"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("AC_LIBACAMD_PATH", os.path.join(ROOT, "tools", "ab", "libacamd_stamps.so"))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import numpy as np, torch
from adaptive_classifier import _native as nv
from adaptive_classifier.encoder import HipBertEncoder
from transformers import BertConfig, BertModel
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = HipBertEncoder(BertModel(BertConfig(num_hidden_layers=2), add_pooling_layer=False).eval(), device=dev)   # ONE fused launch per forward
B, S = 256, 32
g = torch.Generator().manual_seed(1234)
ids = torch.randint(1000, 30000, (B, S), generator=g); ids[:, 0] = 101
lens = torch.randint(8, S + 1, (B,), generator=g); lens[0] = S
mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
ids = (ids * mask).to(dev); mask = mask.to(dev); types = torch.zeros_like(ids)
buf = torch.zeros(512 * 16, dtype=torch.int64, device=dev)
for _ in range(3): enc.encode_cls(ids, types, mask, verify=False)
for mode in ("exchange", "boundary"):
    if mode == "boundary": os.environ["AC_QKV_ATTN_EXCHANGE"] = "0"
    nv.check(nv.lib().ac_gemm_debug_stamps(nv.ptr(buf), 512), "stamps")
    buf.zero_()
    enc.encode_cls(ids, types, mask, verify=False); torch.cuda.synchronize()
    nv.check(nv.lib().ac_gemm_debug_stamps(None, 0), "stamps")
    st = buf.cpu().numpy().reshape(512, 16).astype(np.int64)
    st = st[st[:, 0] != 0]
    d = lambda a, b: np.median(st[:, b] - st[:, a])
    print(f"[{mode}] workgroups {len(st)}: fill {d(0,1):.0f}  k-loop {d(1,2):.0f}  epilogue {d(2,3):.0f} cycles (median)")
    print("   pass 0: staged +%.0f  wave0 attention +%.0f  all waves +%.0f | pass 1: staged +%.0f  wave0 attention +%.0f  all waves +%.0f | tail +%.0f"
          % (d(2, 4), d(4, 5), d(5, 6), d(6, 7), d(7, 8), d(8, 9), d(9, 3)))
    ep = st[:, 3] - st[:, 2]
    print("   epilogue cycles: min %d  p25 %d  median %d  p75 %d  max %d" % (ep.min(), np.percentile(ep, 25), np.median(ep), np.percentile(ep, 75), ep.max()))
