"""Small native row-wise ops of the hot path (softmax over head logits, L2 row normalisation)."""
import torch

from . import _native as nv


def softmax_rows(logits: torch.Tensor) -> torch.Tensor:
    """F.softmax(logits, dim=1) on device (classifier.py:435,1345) via ac_softmax_rows."""
    nv.require_gpu()
    x = logits.detach().to(torch.float32).contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        nv.check(nv.lib().ac_softmax_rows(nv.ptr(x), x.shape[0], x.shape[1], nv.ptr(out), nv.stream_ptr(x.device)),
                 "ac_softmax_rows")
    return out


def l2_normalize_rows(x: torch.Tensor) -> torch.Tensor:
    """F.normalize(x, p=2, dim=1) on device (classifier.py:1450) via ac_l2_normalize_rows."""
    nv.require_gpu()
    x = x.detach().to(torch.float32)          # the kernel reads fp32 rows (Example.embedding may be any float dtype)
    if x.stride(-1) != 1:
        x = x.contiguous()
    out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        nv.check(nv.lib().ac_l2_normalize_rows(nv.ptr(x), x.stride(0), x.shape[0], x.shape[1], nv.ptr(out),
                                               out.stride(0), nv.stream_ptr(x.device)), "ac_l2_normalize_rows")
    return out


def sigmoid(x: torch.Tensor) -> torch.Tensor:
    """torch.sigmoid on device (multilabel.py:43) via ac_sigmoid."""
    nv.require_gpu()
    x = x.detach().contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        nv.check(nv.lib().ac_sigmoid(nv.ptr(x), x.numel(), nv.ptr(out), nv.stream_ptr(x.device)), "ac_sigmoid")
    return out
