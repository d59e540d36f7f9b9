"""Elastic Weight Consolidation -- the `EWC` class contract of the reference
(/root/reference/src/adaptive_classifier/ewc.py:7-116, pinned by tests/test_ewc.py:34-84,128-153).

    EWC(model, dataset, device='cpu', ewc_lambda=100.0)
      .old_params   name -> clone of each trainable parameter            (ewc.py:30-34)
      .fisher_info  name -> sum over batches of grad(nll(sampled y))^2 / #batches   (ewc.py:51-94)
      .ewc_loss(batch_size=None) = lambda * sum F (p - p*)^2 [/ batch_size]          (ewc.py:96-116)

For an AdaptiveHead that lives on a GPU the Fisher pass and the penalty run on the HIP kernels
(`ac_head_fwd_bwd_ce` + `ac_fisher_accumulate`, `ac_ewc_loss`) over the head's flat parameter block
and additionally expose `.fisher_flat` / `.old_flat` for the fused `ac_ewc_adamw_step`.
Any other nn.Module (the reference's tests use nn.Linear) takes the generic autograd route, which
is the reference algorithm itself: torch autograd *is* the reference implementation there.
Batching uses torch's DataLoader(batch_size=32, shuffle=True) exactly like ewc.py:60-64, so the
global RNG is consumed identically.
"""
import ctypes
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native as nv
from .models import _NativeMLP
from .ops import sigmoid, softmax_rows
from .training import LOSS_CE, LOSS_CE_SIGMOID


class EWC:
    def __init__(self, model: nn.Module, dataset, device: str = "cpu", ewc_lambda: float = 100.0):
        self.model = model
        self.device = device
        self.ewc_lambda = ewc_lambda
        self._native = (isinstance(model, _NativeMLP) and model.native_dims() is not None
                        and next(model.parameters()).is_cuda)
        if self._native:
            flat = model.flat_params()               # parameters become views of this block
            self.old_flat = flat.detach().clone()
        self.old_params = {n: p.data.clone() for n, p in model.named_parameters() if p.requires_grad}
        self.fisher_info = self._compute_fisher(dataset)

    # ------------------------------------------------------------------------------------------
    def _compute_fisher(self, dataset) -> Dict[str, torch.Tensor]:
        self.model.eval()                            # ewc.py:57 (dropout off)
        loader = torch.utils.data.DataLoader(dataset, batch_size=32, shuffle=True)
        if self._native:
            return self._compute_fisher_native(loader)
        fisher = {n: torch.zeros_like(p) for n, p in self.model.named_parameters() if p.requires_grad}
        for batch_embeddings, _ in loader:
            self.model.zero_grad()
            outputs = self.model(batch_embeddings.to(self.device))
            probs = F.softmax(outputs, dim=1)
            log_probs = F.log_softmax(outputs, dim=1)
            sampled = torch.multinomial(probs, 1).squeeze(-1)
            F.nll_loss(log_probs, sampled).backward()
            for n, p in self.model.named_parameters():
                if p.grad is not None:
                    fisher[n] += p.grad.data ** 2 / len(loader)
        return fisher

    def _compute_fisher_native(self, loader, sampled_labels=None):
        from .training import HeadTrainer
        head = self.model
        tr = HeadTrainer(head)
        flat = head.flat_params()
        self.fisher_flat = torch.zeros_like(flat)
        inv = 1.0 / len(loader)
        for bi, (batch_embeddings, _) in enumerate(loader):
            X = batch_embeddings.to(flat.device)
            # a sigmoid-output head (multi-label) feeds its probabilities to softmax/nll (ewc.py:74-84)
            sig = hasattr(head, "num_classes") and type(head).__name__ == "MultiLabelAdaptiveHead"
            if sampled_labels is None:
                out = head.forward_native(X)
                probs = softmax_rows(sigmoid(out) if sig else out)
                y = torch.multinomial(probs, 1).squeeze(-1)     # ewc.py:81
            else:
                y = sampled_labels[bi]
            tr.forward_backward_loss(X, y=y, loss_kind=LOSS_CE_SIGMOID if sig else LOSS_CE, dropout_p=0.0)  # eval mode
            with torch.cuda.device(flat.device):
                nv.check(nv.lib().ac_fisher_accumulate(nv.ptr(tr.grads), inv, nv.ptr(self.fisher_flat),
                                                       flat.numel(), nv.stream_ptr(flat.device)),
                         "ac_fisher_accumulate")
        # name -> view dict, same keys/shapes as the reference's
        fisher, off = {}, 0
        for n, p in head.named_parameters():
            fisher[n] = self.fisher_flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        return fisher

    # ------------------------------------------------------------------------------------------
    def ewc_loss(self, batch_size: Optional[int] = None) -> torch.Tensor:
        if self._native and not torch.is_grad_enabled():
            return self.ewc_loss_native(batch_size)
        loss = 0
        for n, p in self.model.named_parameters():
            if p.requires_grad:
                loss += (self.fisher_info[n] * (p - self.old_params[n]) ** 2).sum()
        if batch_size is not None:
            loss = loss / batch_size
        return self.ewc_lambda * loss

    def ewc_loss_native(self, batch_size: Optional[int] = None) -> torch.Tensor:
        flat = self.model.flat_params()
        lam = self.ewc_lambda / batch_size if batch_size is not None else self.ewc_lambda
        out = torch.zeros((), dtype=torch.float32, device=flat.device)
        scratch = torch.empty(8192, dtype=torch.uint8, device=flat.device)
        with torch.cuda.device(flat.device):
            nv.check(nv.lib().ac_ewc_loss(nv.ptr(flat), nv.ptr(self.fisher_flat), nv.ptr(self.old_flat),
                                          flat.numel(), lam, nv.ptr(out), nv.ptr(scratch),
                                          nv.stream_ptr(flat.device)), "ac_ewc_loss")
        return out
