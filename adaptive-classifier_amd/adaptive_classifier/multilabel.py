"""Multi-label extension -- mirrors /root/reference/src/adaptive_classifier/multilabel.py:
`MultiLabelAdaptiveHead` (:15-68, sigmoid output, default nn.Linear init) and
`MultiLabelAdaptiveClassifier` (:71-425: adaptive / per-label thresholds, min/max predictions, one stored
example per (text, label) pair, BCE training on multi-hot targets).

On MI355X the head's logits come from `ac_head_forward`, the sigmoid from `ac_sigmoid`, and training runs the
fused `ac_head_train_step` with AC_LOSS_BCE_SIGMOID (`nn.BCELoss` on sigmoid outputs, multilabel.py:361).
Reference quirk kept: adding NEW labels to an existing classifier goes through the base class's
`_train_new_classes` (classifier.py:166-180), i.e. CrossEntropyLoss applied to the sigmoid outputs
(AC_LOSS_CE_SIGMOID); only the from-scratch / same-label path uses BCE.
"""
import logging
from collections import defaultdict
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .classifier import AdaptiveClassifier
from .models import _NativeMLP
from .ops import l2_normalize_rows, sigmoid
from .training import LOSS_BCE_SIGMOID, LOSS_CE_SIGMOID

logger = logging.getLogger(__name__)


class MultiLabelAdaptiveHead(_NativeMLP):
    """(Linear -> ReLU -> Dropout(0.1)) x n -> Linear -> sigmoid (multilabel.py:15-43)."""

    DROPOUT_P = 0.1

    def __init__(self, input_dim: int, num_classes: int, hidden_dims: List[int] = None):
        super().__init__()
        if hidden_dims is None:
            hidden_dims = [input_dim // 2]
        layers, prev = [], input_dim
        for dim in hidden_dims:
            layers += [nn.Linear(prev, dim), nn.ReLU(), nn.Dropout(self.DROPOUT_P)]
            prev = dim
        layers.append(nn.Linear(prev, num_classes))
        self.model = nn.Sequential(*layers)
        self.num_classes = num_classes
        self._init_native()

    def forward(self, x):
        if self.model[0].weight.is_cuda and not torch.is_grad_enabled() and self.native_dims() is not None:
            return sigmoid(self.forward_native(x))
        return torch.sigmoid(self.model(x))

    def update_num_classes(self, new_num_classes: int):
        """Grow the output layer; old rows kept, new rows xavier-initialised (multilabel.py:45-68)."""
        if new_num_classes <= self.num_classes:
            return
        final = self.model[-1]
        new = nn.Linear(final.in_features, new_num_classes)
        with torch.no_grad():
            new.weight[: self.num_classes] = final.weight.detach().cpu()
            new.bias[: self.num_classes] = final.bias.detach().cpu()
            nn.init.xavier_uniform_(new.weight[self.num_classes:])
            nn.init.zeros_(new.bias[self.num_classes:])
        self.model[-1] = new.to(final.weight.device)
        self.num_classes = new_num_classes
        self._flat = None


class MultiLabelAdaptiveClassifier(AdaptiveClassifier):
    """Multi-label AdaptiveClassifier: several labels per text, threshold-based decisions."""

    LOSS_KIND = LOSS_CE_SIGMOID          # what the inherited new-class loop computes on a sigmoid head

    def __init__(self, model_name: str, device: Optional[str] = None, config: Optional[Dict[str, Any]] = None,
                 seed: int = 42, default_threshold: float = 0.5, min_predictions: int = 1,
                 max_predictions: Optional[int] = None, *, use_onnx="auto", trust_remote_code: bool = False, encoder=None,
                 tokenizer=None):
        # (use_onnx / trust_remote_code, keyword-only: the reference's constructor lacks them, which is why ITS inherited `load()`
        #  -- `cls(..., use_onnx=...)`, classifier.py:708-716 -- dies with a TypeError for this class; here load() works)
        super().__init__(model_name, device, config, seed, use_onnx=use_onnx, trust_remote_code=trust_remote_code,
                         encoder=encoder, tokenizer=tokenizer)
        self.default_threshold = default_threshold
        self.min_predictions = min_predictions
        self.max_predictions = max_predictions
        self.label_thresholds = {}
        self.adaptive_head = None

    def _initialize_adaptive_head(self):
        hidden_dims = [self.embedding_dim, self.embedding_dim // 2]
        self.adaptive_head = MultiLabelAdaptiveHead(self.embedding_dim, len(self.label_to_id),
                                                    hidden_dims=hidden_dims).to(self.device)

    def _get_adaptive_threshold(self, num_labels: int) -> float:
        """multilabel.py:112-130."""
        for limit, factor in ((2, 1.0), (5, 0.8), (10, 0.6), (20, 0.4)):
            if num_labels <= limit:
                return self.default_threshold * factor
        return self.default_threshold * 0.2

    # ------------------------------------------------------------------------------ prediction
    def _head_outputs(self, emb: torch.Tensor) -> torch.Tensor:
        """The reference's inherited predict_batch / _predict_regular call `self.adaptive_head(x)` -- for this head
        the SIGMOID outputs (multilabel.py:43) -- and then apply F.softmax to them (classifier.py:1342-1345, :432-435):
        the blend sees softmax(sigmoid(z)), not softmax(z)."""
        return sigmoid(self.adaptive_head.forward_native(emb))

    def _head_probabilities(self, emb: torch.Tensor):
        """sigmoid(head(emb)) for a device batch -> numpy [b, C] (one D2H)."""
        self.adaptive_head.eval()
        with torch.no_grad():
            return sigmoid(self.adaptive_head.forward_native(emb)).cpu().numpy()

    def _decide(self, probs, threshold, max_labels):
        """Threshold / min / max logic of predict_multilabel (multilabel.py:167-226) for one probability row."""
        C = len(self.id_to_label)
        preds = []
        for i in range(min(len(probs), C)):
            label = self.id_to_label[i]
            p = float(probs[i])
            if p >= self.label_thresholds.get(label, threshold):
                preds.append((label, p))
        preds.sort(key=lambda x: x[1], reverse=True)
        if max_labels and len(preds) > max_labels:
            preds = preds[:max_labels]
        if len(preds) < self.min_predictions:
            order = sorted(range(C), key=lambda i: -float(probs[i]))[: min(self.min_predictions, C)]   # torch.topk
            extra = [(self.id_to_label[i], float(probs[i])) for i in order
                     if not any(p[0] == self.id_to_label[i] for p in preds)]
            preds.extend(extra[: self.min_predictions - len(preds)])
            preds.sort(key=lambda x: x[1], reverse=True)
        return preds

    def predict_multilabel(self, text: str, threshold: Optional[float] = None,
                           max_labels: Optional[int] = None) -> List[Tuple[str, float]]:
        if not text:
            raise ValueError("Empty input text")
        num_labels = len(self.label_to_id)
        if num_labels == 0:
            return []
        if threshold is None:
            threshold = self._get_adaptive_threshold(num_labels)
        max_labels = max_labels or self.max_predictions
        emb = self._embed_device([text])
        if self.adaptive_head is not None:
            return self._decide(self._head_probabilities(emb)[0], threshold, max_labels)
        protos = self.memory.get_nearest_prototypes(emb[0].cpu(), k=min(num_labels, max_labels) if max_labels else num_labels)
        return [(l, s) for l, s in protos if s >= threshold]

    def predict_multilabel_batch(self, texts: List[str], threshold: Optional[float] = None,
                                 max_labels: Optional[int] = None) -> List[List[Tuple[str, float]]]:
        """Batched form (one encoder call, one head call); same decisions as predict_multilabel per text."""
        if not texts:
            raise ValueError("Empty input batch")
        if threshold is None:
            threshold = self._get_adaptive_threshold(len(self.label_to_id))
        max_labels = max_labels or self.max_predictions
        probs = self._head_probabilities(self._embed_device(texts))
        return [self._decide(p, threshold, max_labels) for p in probs]

    def predict(self, text: str, k: int = 5) -> List[Tuple[str, float]]:
        preds = self.predict_multilabel(text, max_labels=k)
        return preds[:k] if preds else super().predict(text, k)

    # ------------------------------------------------------------------------------ training
    def add_examples(self, texts: List[str], labels: List[List[str]]):
        if not texts or not labels:
            raise ValueError("Empty input lists")
        if len(texts) != len(labels):
            raise ValueError("Mismatched text and label lists")
        flat_t, flat_l = [], []
        for text, text_labels in zip(texts, labels):
            for label in text_labels or []:          # one stored example per (text, label) pair (:259-266)
                flat_t.append(text)
                flat_l.append(label)
        if flat_t:
            super().add_examples(flat_t, flat_l)
        self._update_label_thresholds()

    def _update_label_thresholds(self):
        """Rare labels get lower thresholds, common ones higher (multilabel.py:275-305)."""
        if not self.memory.examples:
            return
        counts = {label: len(ex) for label, ex in self.memory.examples.items()}
        total = sum(counts.values())
        for label, count in counts.items():
            f = count / total
            factor = 0.3 if f < 0.05 else 0.5 if f < 0.1 else 1.2 if f > 0.3 else 1.0
            self.label_thresholds[label] = self.default_threshold * factor

    def _multi_hot_dataset(self):
        """Unique texts -> (embedding, multi-hot target) in the reference's iteration order (:322-349)."""
        C = len(self.label_to_id)
        text_to_labels = defaultdict(set)
        first_emb = {}
        for label, examples in self.memory.examples.items():
            for ex in examples:
                text_to_labels[ex.text].add(label)
        for text, labels in text_to_labels.items():
            for label in labels:
                hit = next((ex.embedding for ex in self.memory.examples[label] if ex.text == text), None)
                if hit is not None:
                    first_emb[text] = hit
                    break
        embs, targets = [], []
        for text, labels in text_to_labels.items():
            if text in first_emb:
                embs.append(first_emb[text])
                t = torch.zeros(C)
                for label in labels:
                    if label in self.label_to_id:
                        t[self.label_to_id[label]] = 1.0
                targets.append(t)
        return embs, targets

    def _train_adaptive_head(self, epochs: int = 10):
        """BCE training on multi-hot targets (multilabel.py:309-413): AdamW(1e-3, wd .01), clip 1.0,
        <= 10 epochs, early stop patience 3, no LR scheduler."""
        if not self.memory.examples:
            return
        embs, targets = self._multi_hot_dataset()
        if not embs:
            return
        X = l2_normalize_rows(torch.stack(embs).to(self.device))
        T = torch.stack(targets).to(self.device)
        self._run_epochs(X, None, batch_size=min(32, X.shape[0]), epochs=epochs, use_scheduler=False,
                         loss_kind=LOSS_BCE_SIGMOID, targets=T)

    def get_label_statistics(self) -> Dict[str, Any]:
        stats = super().get_example_statistics()
        stats.update({"label_thresholds": dict(self.label_thresholds),
                      "adaptive_threshold": self._get_adaptive_threshold(len(self.label_to_id)),
                      "default_threshold": self.default_threshold, "min_predictions": self.min_predictions,
                      "max_predictions": self.max_predictions})
        return stats
