"""Per-workgroup shader-clock stamps (start, ring filled, k-loop done, end) of each of the four GEMMs of an encoder layer AS THEY RUN IN
THE NETWORK (bench batch, 5141 rows): one forward per GEMM with AC_GEMM_STAMP_EPI / AC_GEMM_STAMP_K selecting it.  A 2-layer model:
layer 0 is the full-width layer, layer 1 the CLS-only one (its GEMMs have other shapes / kernels and do not match the filters)."""
import os, sys
os.environ.setdefault("AC_TEST_HOOKS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import numpy as np, torch
from adaptive_classifier import _native as nv
from adaptive_classifier.encoder import HipBertEncoder
from transformers import BertConfig, BertModel
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = HipBertEncoder(BertModel(BertConfig(num_hidden_layers=2), add_pooling_layer=False).eval(), device=dev)
B, S = 256, 32
g = torch.Generator().manual_seed(1234)
ids = torch.randint(1000, 30000, (B, S), generator=g); ids[:, 0] = 101
lens = torch.randint(8, S + 1, (B,), generator=g); lens[0] = S
mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64)
ids = (ids * mask).to(dev); mask = mask.to(dev); types = torch.zeros_like(ids)
for _ in range(3): enc.encode_cls(ids, types, mask, verify=False)
buf = torch.zeros(512 * 16, dtype=torch.int64, device=dev)
# (epilogue class, K, stamps per workgroup, name)
for epi, K, stride, name in ((8, 768, 16, "QKV + attention epilogue (256 x 192)"), (7, 768, 4, "attention output + LayerNorm (128 x 128)"),
                             (2, 768, 4, "FFN1 + GELU -> planes (256 x 256)"), (7, 3072, 4, "FFN2 + LayerNorm (128 x 128)")):
    os.environ["AC_GEMM_STAMP_EPI"], os.environ["AC_GEMM_STAMP_K"] = str(epi), str(K)
    nv.check(nv.lib().ac_gemm_debug_stamps(nv.ptr(buf), 512), "stamps"); buf.zero_()
    enc.encode_cls(ids, types, mask, verify=False); torch.cuda.synchronize()
    nv.check(nv.lib().ac_gemm_debug_stamps(None, 0), "stamps")
    st = buf.cpu().numpy().reshape(-1, stride).astype(np.int64)
    st = st[st[:, 0] != 0]
    t0 = st[:, 0].min()
    med = lambda a, b: np.median(st[:, b] - st[:, a])
    span = st[:, 3].max() - t0
    print(f"{name:44s} workgroups {len(st):3d}: fill {med(0,1):6.0f}  k-loop {med(1,2):7.0f}  epilogue {med(2,3):6.0f} (max {np.max(st[:,3]-st[:,2]):6d})  "
          f"first start -> last end {span:7d} cycles; k-loop share {med(1,2)/span:.2f}; start skew {np.max(st[:,0]-t0):5d}")
