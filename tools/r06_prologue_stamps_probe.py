"""Round 6: where pack_prologue_kernel (one workgroup) spends its ~18 us: s_memrealtime stamps (100 MHz) at its phase boundaries, from a
measurement build (-DAC_PROLOGUE_STAMPS, tools/ab/libacamd_pkstamps.so via AC_LIBACAMD_PATH)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "adaptive-classifier_amd")]
import numpy as np, torch
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
clf, hf = bench.make_classifier(dev, 0, 1)
ids, types, mask = bench.synthetic_tokens(dev, 0)
enc = clf.model
rows = []
for _ in range(20):
    enc.encode_cls(ids, types, mask, verify=False)
    torch.cuda.synchronize()
    total = enc.workspace_bytes(ids.shape[0], ids.shape[1])
    st = enc._ws[total - 256 + 64: total - 256 + 64 + 56].cpu().numpy().view(np.uint64).astype(np.int64)
    rows.append(np.diff(st) * 10.0 / 1e3)          # us
names = ["init+zeroing", "mask pass", "lens/scan", "cu+report", "tok_src fill", "tile table"]
med = np.median(np.array(rows[5:]), axis=0)
print("pack_prologue_kernel phases (us, median of 15): " + "  ".join("%s %.2f" % (n, v) for n, v in zip(names, med)) + "   sum %.2f" % med.sum())
