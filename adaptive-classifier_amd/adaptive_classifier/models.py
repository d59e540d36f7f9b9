"""Example / AdaptiveHead / ModelConfig -- the public surface of the reference's models.py
(/root/reference/src/adaptive_classifier/models.py:10-196), with the head's arithmetic on MI355X.

AdaptiveHead keeps the exact nn.Module anatomy the reference exposes and its callers rely on
(classifier.py:1530, tests/test_new_class_accuracy_preservation.py:290-298):
  .model = nn.Sequential(Linear, ReLU, Dropout(0.1), ..., Linear), Linear at [0], [3], [-1];
  state_dict keys model.{0,3,6}.{weight,bias}; deterministic seed-42 init (models.py:51-66).
Initialisation happens on the host with torch's CPU generator so weights are bit-identical to
the reference; the parameters are then packed into ONE flat fp32 device block (the layout of
include/acamd.h) of which the nn.Parameters are views, so that forward / backward / the fused
EWC-AdamW step are single native launches over that block.
"""
import ctypes
import logging
from dataclasses import dataclass
from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from . import _native as nv

logger = logging.getLogger(__name__)


@dataclass
class Example:
    """One stored training example (models.py:10-28)."""
    text: str
    label: str
    embedding: Optional[torch.Tensor] = None

    def to_dict(self) -> Dict[str, Any]:
        emb = None if self.embedding is None else self.embedding.tolist()
        return {"text": self.text, "label": self.label, "embedding": emb}

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> "Example":
        emb = data.get("embedding")
        return cls(text=data["text"], label=data["label"],
                   embedding=None if emb is None else torch.tensor(emb))


# key -> default, in the order ModelConfig.to_dict() emits them (models.py:100-196)
_CONFIG_DEFAULTS = [
    ("max_length", 512), ("batch_size", 32), ("learning_rate", 0.001), ("warmup_steps", 0),
    ("max_examples_per_class", 1000), ("prototype_update_frequency", 100), ("similarity_threshold", 0.6),
    ("ewc_lambda", 100.0), ("num_representative_examples", 5),
    ("epochs", 10), ("early_stopping_patience", 3), ("min_examples_per_class", 3),
    ("prototype_weight", 0.7), ("neural_weight", 0.3), ("min_confidence", 0.1),
    ("device_map", "auto"), ("quantization", None), ("gradient_checkpointing", False),
    ("enable_strategic_mode", False), ("cost_function_type", "separable"), ("strategic_lambda", 0.1),
    ("cost_coefficients", None), ("strategic_training_frequency", 10),
    ("strategic_blend_regular_weight", 0.6), ("strategic_blend_strategic_weight", 0.4),
    ("strategic_robust_proto_weight", 0.8), ("strategic_robust_head_weight", 0.2),
    ("strategic_prediction_proto_weight", 0.5), ("strategic_prediction_head_weight", 0.5),
]


class ModelConfig:
    """Flat attribute bag with the reference's keys and defaults; unknown keys in update() only warn."""

    def __init__(self, config: Optional[Dict[str, Any]] = None):
        self.config = config or {}
        for key, default in _CONFIG_DEFAULTS:
            if key == "cost_coefficients" and default is None:
                default = {}
            setattr(self, key, self.config.get(key, default))

    def update(self, **kwargs):
        for key, value in kwargs.items():
            if hasattr(self, key):
                setattr(self, key, value)
            else:
                logger.warning(f"Unknown configuration parameter: {key}")

    def to_dict(self) -> Dict[str, Any]:
        return {key: getattr(self, key) for key, _ in _CONFIG_DEFAULTS}


def _seeded_linear(fan_in, fan_out, kind):
    """nn.Linear initialised exactly like models.py:49-53 / :63-66 (global seed side effect kept)."""
    layer = nn.Linear(fan_in, fan_out)
    torch.manual_seed(42)
    if kind == "hidden":
        nn.init.kaiming_uniform_(layer.weight, mode="fan_in", nonlinearity="relu")
    else:
        nn.init.xavier_uniform_(layer.weight)
    nn.init.zeros_(layer.bias)
    return layer


class _NativeMLP(nn.Module):
    """Shared plumbing of the MLP heads: `.model` is an nn.Sequential whose Linear parameters are views of
    ONE flat fp32 device block (layout of include/acamd.h) consumed by the HIP kernels."""

    def _init_native(self):
        self._flat = None          # flat fp32 device block the parameters are views of
        self._ws = None

    # ---- flat-block management -------------------------------------------------------------
    def linears(self):
        return [m for m in self.model if isinstance(m, nn.Linear)]

    def native_dims(self):
        """ac_head_dims if this head has the two-hidden-layer shape the fused kernels cover."""
        lin = self.linears()
        if len(lin) != 3:
            return None
        return nv.ac_head_dims(lin[0].in_features, lin[0].out_features, lin[1].out_features, lin[2].out_features)

    def flat_params(self):
        """The parameters as one contiguous device tensor (re-packed if torch re-allocated them)."""
        params = [p for l in self.linears() for p in (l.weight, l.bias)]
        total = sum(p.numel() for p in params)
        dev = params[0].device
        ok = (self._flat is not None and self._flat.numel() == total and self._flat.device == dev)
        if ok:
            off = 0
            for p in params:
                if p.data_ptr() != self._flat.data_ptr() + 4 * off or not p.is_contiguous():
                    ok = False
                    break
                off += p.numel()
        if not ok:
            flat = torch.empty(total, dtype=torch.float32, device=dev)
            off = 0
            for p in params:
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view(p.shape)
                off += n
            self._flat = flat
        return self._flat

    def _workspace(self, B):
        dims = self.native_dims()
        need = ctypes.c_size_t(0)
        nv.check(nv.lib().ac_head_workspace(ctypes.byref(dims), B, ctypes.byref(need)), "ac_head_workspace")
        dev = self.model[0].weight.device
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != dev:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        return self._ws

    # ---- forward ---------------------------------------------------------------------------
    def forward_native(self, x: torch.Tensor) -> torch.Tensor:
        """Eval-mode LOGITS [B, C] through ac_head_forward (no autograd)."""
        dims = self.native_dims()
        if dims is None:
            raise nv.NativeError("native head forward needs hidden_dims of length 2")
        flat = self.flat_params()
        x = x.detach().to(device=flat.device, dtype=torch.float32)
        if x.dim() == 1:
            x = x.unsqueeze(0)
        if x.stride(1) != 1:
            x = x.contiguous()
        B = x.shape[0]
        out = torch.empty((B, dims.C), dtype=torch.float32, device=flat.device)
        ws = self._workspace(B)
        with torch.cuda.device(flat.device):
            nv.check(nv.lib().ac_head_forward(ctypes.byref(dims), nv.ptr(flat), nv.ptr(x), x.stride(0), B,
                                              nv.ptr(out), nv.ptr(ws), ws.numel(), nv.stream_ptr(flat.device)),
                     "ac_head_forward")
        return out


class AdaptiveHead(_NativeMLP):
    """MLP head: (Linear -> ReLU -> Dropout(0.1)) x n -> Linear (models.py:30-98)."""

    DROPOUT_P = 0.1

    def __init__(self, input_dim: int, num_classes: int, hidden_dims: Optional[list] = None):
        super().__init__()
        if hidden_dims is None:
            hidden_dims = [input_dim]
        layers, prev = [], input_dim
        for dim in hidden_dims:
            layers += [_seeded_linear(prev, dim, "hidden"), nn.ReLU(), nn.Dropout(self.DROPOUT_P)]
            prev = dim
        layers.append(_seeded_linear(prev, num_classes, "out"))
        self.model = nn.Sequential(*layers)
        self._init_native()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Always returns [B, C] (models.py:71-80).

        Inference (no autograd needed) on a GPU runs the HIP kernels.  When autograd is required
        (callers outside the hot path that differentiate through the head) the torch modules are
        used so gradients flow to the same parameters.
        """
        if x.dim() == 1:
            x = x.unsqueeze(0)
        if self.model[0].weight.is_cuda and not torch.is_grad_enabled() and self.native_dims() is not None:
            return self.forward_native(x)
        return self.model(x)

    def update_num_classes(self, num_classes: int):
        """Grow the output layer, keeping existing class rows (models.py:82-98)."""
        old = self.model[-1]
        if num_classes > old.weight.size(0):
            new = _seeded_linear(old.weight.size(1), num_classes, "out")
            with torch.no_grad():
                new.weight[: old.weight.size(0)] = old.weight.detach().cpu()
                new.bias[: old.weight.size(0)] = old.bias.detach().cpu()
            self.model[-1] = new.to(old.weight.device)
            self._flat = None
