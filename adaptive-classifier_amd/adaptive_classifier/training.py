"""Native training step for AdaptiveHead: the inner loop of classifier.py:1483-1507
(_train_adaptive_head) and :327-353 (_train_new_classes) on MI355X.

One step = `ac_head_fwd_bwd_ce` (train-mode forward with the dropout masks, mean CE, backward into
a flat gradient block) + `ac_ewc_adamw_step` (EWC gradient + clip_grad_norm_(1.0) + AdamW over the
flat block).  Optimizer state (m, v) lives here like torch.optim.AdamW keeps it -- fresh for every
add_examples() call, as in the reference (a new AdamW is built at classifier.py:308,1464).
"""
import ctypes

import torch

from . import _native as nv

REDUCE_SCRATCH_BYTES = 8192
LOSS_CE, LOSS_BCE_SIGMOID, LOSS_CE_SIGMOID = 0, 1, 2      # AC_LOSS_* of include/acamd.h
LOSS_STEPWISE = 0x100                                     # AC_LOSS_STEPWISE: per-call opt-out of the persistent epoch kernel


class HeadTrainer:
    def __init__(self, head, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=1.0):
        nv.require_gpu()
        self.head = head
        self.dims = head.native_dims()
        if self.dims is None:
            raise nv.NativeError("HeadTrainer needs an AdaptiveHead with two hidden layers")
        self.flat = head.flat_params()
        if not self.flat.is_cuda:
            raise nv.NativeError("HeadTrainer: the head must live on a GPU (no CPU fallback)")
        dev = self.flat.device
        self.device = dev
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.grads = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.t = 0
        self.scratch = torch.zeros(REDUCE_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.out = torch.zeros(2, dtype=torch.float32, device=dev)      # [ewc penalty, grad norm]
        self.out3 = torch.zeros(3, dtype=torch.float32, device=dev)     # fused step: [ce, ewc penalty, grad norm]
        self.loss_accum = torch.zeros(1, dtype=torch.float32, device=dev)   # epoch running sum (device)
        self._ws = None
        self._snap = None
        self._snap_t = 0

    def _workspace(self, B):
        need = ctypes.c_size_t(0)
        nv.check(nv.lib().ac_head_workspace(ctypes.byref(self.dims), B, ctypes.byref(need)), "ac_head_workspace")
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        return self._ws

    def dropout_masks(self, B, p=0.1, generator=None):
        """Bernoulli(1-p) keep masks for the two hidden layers, uint8 [B,H1], [B,H2]."""
        m1 = (torch.rand((B, self.dims.H1), device=self.device, generator=generator) >= p).to(torch.uint8)
        m2 = (torch.rand((B, self.dims.H2), device=self.device, generator=generator) >= p).to(torch.uint8)
        return m1, m2

    def forward_backward(self, X, y, mask1=None, mask2=None, dropout_p=0.1):
        """Fills self.grads and self.loss (device scalars; no host sync)."""
        X = X.to(device=self.device, dtype=torch.float32)
        if X.stride(-1) != 1:
            X = X.contiguous()
        y = y.to(device=self.device, dtype=torch.int64).contiguous()
        B = X.shape[0]
        ws = self._workspace(B)
        with torch.cuda.device(self.device):
            nv.check(nv.lib().ac_head_fwd_bwd_ce(
                ctypes.byref(self.dims), nv.ptr(self.flat), nv.ptr(X), X.stride(0), nv.ptr(y),
                nv.ptr(mask1), nv.ptr(mask2), dropout_p, B, nv.ptr(self.loss), nv.ptr(self.grads),
                nv.ptr(ws), ws.numel(), nv.stream_ptr(self.device)), "ac_head_fwd_bwd_ce")
        return self.loss

    def optimizer_step(self, fisher=None, old_params=None, lambda_over_B=0.0):
        """clip_grad_norm_ + AdamW (+ EWC gradient when fisher/old_params are given)."""
        self.t += 1
        with torch.cuda.device(self.device):
            nv.check(nv.lib().ac_ewc_adamw_step(
                nv.ptr(self.flat), nv.ptr(self.grads), nv.ptr(self.m), nv.ptr(self.v),
                nv.ptr(fisher), nv.ptr(old_params), self.flat.numel(), lambda_over_B, self.max_grad_norm,
                self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.t,
                nv.ptr(self.out), nv.ptr(self.scratch), nv.stream_ptr(self.device)), "ac_ewc_adamw_step")
        return self.out

    def step(self, X, y, mask1=None, mask2=None, dropout_p=0.1, fisher=None, old_params=None, lambda_over_B=0.0):
        self.forward_backward(X, y, mask1, mask2, dropout_p)
        self.optimizer_step(fisher, old_params, lambda_over_B)
        return self.loss, self.out

    def forward_backward_loss(self, X, y=None, targets=None, loss_kind=LOSS_CE, mask1=None, mask2=None,
                              dropout_p=0.1):
        """forward_backward with a selectable loss (LOSS_CE / LOSS_BCE_SIGMOID / LOSS_CE_SIGMOID)."""
        X = X.to(device=self.device, dtype=torch.float32)
        if X.stride(-1) != 1:
            X = X.contiguous()
        B = X.shape[0]
        y = None if y is None else y.to(device=self.device, dtype=torch.int64).contiguous()
        targets = None if targets is None else targets.to(device=self.device, dtype=torch.float32).contiguous()
        ws = self._workspace(B)
        with torch.cuda.device(self.device):
            nv.check(nv.lib().ac_head_fwd_bwd_loss(
                ctypes.byref(self.dims), nv.ptr(self.flat), nv.ptr(X), X.stride(0), nv.ptr(y), nv.ptr(targets),
                0 if targets is None else targets.stride(0), loss_kind, nv.ptr(mask1), nv.ptr(mask2), dropout_p, B,
                nv.ptr(self.loss), nv.ptr(self.grads), nv.ptr(ws), ws.numel(), nv.stream_ptr(self.device)),
                "ac_head_fwd_bwd_loss")
        return self.loss

    def fused_step(self, X_all, y_all, index=None, dropout_p=0.1, seed=0, fisher=None, old_params=None,
                   lambda_over_B=0.0, loss_kind=LOSS_CE, targets_all=None):
        """One call = gather batch (rows `index` of X_all / y_all) + train-mode forward with in-kernel
        counter-based dropout + CE + backward + EWC/clip/AdamW (`ac_head_train_step`).  No torch kernels,
        no host sync; the step's CE + penalty is added to self.loss_accum on device."""
        B = int(index.numel()) if index is not None else X_all.shape[0]
        ws = self._workspace(B)
        self.t += 1
        with torch.cuda.device(self.device):
            nv.check(nv.lib().ac_head_train_step(
                ctypes.byref(self.dims), nv.ptr(self.flat), nv.ptr(self.m), nv.ptr(self.v), nv.ptr(self.grads),
                nv.ptr(X_all), X_all.stride(0), nv.ptr(y_all), nv.ptr(targets_all),
                0 if targets_all is None else targets_all.stride(0), loss_kind, nv.ptr(index), B, dropout_p, seed,
                nv.ptr(fisher), nv.ptr(old_params), lambda_over_B, self.max_grad_norm, self.lr, self.betas[0],
                self.betas[1], self.eps, self.weight_decay, self.t, nv.ptr(self.out3), nv.ptr(self.loss_accum),
                nv.ptr(ws), ws.numel(), nv.stream_ptr(self.device)), "ac_head_train_step")
        return self.out3

    def fused_epoch(self, X_all, y_all, order, batch, dropout_p=0.1, seed0=0, fisher=None, old_params=None,
                    lambda_B=0.0, loss_kind=LOSS_CE, targets_all=None, stepwise=False):
        """One call = one epoch of fused steps over consecutive `batch`-row slices of `order`
        (`ac_head_train_epoch`): step i uses dropout seed seed0 + i; EWC weight lambda_B / rows_i.
        order=None: X_all / y_all / targets_all are already in epoch order (batches = consecutive row slices, no
        per-step gather).  stepwise=True: this call uses the step-by-step launches (AC_LOSS_STEPWISE), whatever the
        process-wide persistent-kernel switch says.  Returns the number of steps taken."""
        n_total = int(order.numel()) if order is not None else int(X_all.shape[0])
        ws = self._workspace(min(batch, max(n_total, 1)))
        done = ctypes.c_int(0)
        # the persistent epoch kernel publishes the output layer step by step; should one of its grid barriers give up (a GPU
        # shared with another compute process), the epoch is void and `restore_epoch()` puts the parameters back
        if self._snap is None:
            self._snap = torch.empty((3,) + tuple(self.flat.shape), dtype=self.flat.dtype, device=self.flat.device)
        self._snap[0].copy_(self.flat); self._snap[1].copy_(self.m); self._snap[2].copy_(self.v)
        self._snap_t = self.t
        with torch.cuda.device(self.device):
            nv.check(nv.lib().ac_head_train_epoch(
                ctypes.byref(self.dims), nv.ptr(self.flat), nv.ptr(self.m), nv.ptr(self.v), nv.ptr(self.grads),
                nv.ptr(X_all), X_all.stride(0), nv.ptr(y_all), nv.ptr(targets_all),
                0 if targets_all is None else targets_all.stride(0), loss_kind | (LOSS_STEPWISE if stepwise else 0),
                nv.ptr(order), n_total, batch, dropout_p, seed0, nv.ptr(fisher), nv.ptr(old_params), lambda_B, self.max_grad_norm, self.lr,
                self.betas[0], self.betas[1], self.eps, self.weight_decay, self.t + 1, nv.ptr(self.out3),
                nv.ptr(self.loss_accum), nv.ptr(ws), ws.numel(), ctypes.byref(done),
                nv.stream_ptr(self.device)), "ac_head_train_epoch")
        self.t += done.value
        return done.value

    def restore_epoch(self):
        """Undo the last fused_epoch: parameters, both Adam moments and the step counter.  (A persistent epoch writes the
        moments back when it completes -- also when its loss is a genuine NaN -- and a barrier that gives up at the last
        step leaves them half written, so they are part of the snapshot.)"""
        self.flat.copy_(self._snap[0]); self.m.copy_(self._snap[1]); self.v.copy_(self._snap[2])
        self.t = self._snap_t
