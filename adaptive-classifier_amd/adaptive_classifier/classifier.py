"""AdaptiveClassifier -- the reference's public API for the predict()/add_examples() hot path
(/root/reference/src/adaptive_classifier/classifier.py), re-plumbed onto the MI355X kernels.

Same constructor, method names, argument meaning, return types and error behaviour as the
reference for the path in scope:
  add_examples :132-200 | predict :392-413 -> _predict_regular :415-480 | predict_batch :1308-1388 |
  _get_embeddings :1249-1282 | _initialize_adaptive_head :1238-1247 | _train_adaptive_head :1428-1522 |
  _train_new_classes :202-367 | get_memory_stats :1230 | get_example_statistics :1284 | clear_memory :1390 |
  merge_classifiers :1402-1426 | _update_adaptive_head :1524-1531 |
  save / load (format of :524-628, :764-915: config.json, examples.json, model.safetensors).
Out of scope here and rejected loudly: ONNX runtime/export (:59-81,1031-1104), strategic mode
(:482-522,1594-1823), Hub upload / model card (:917-1183).

What changed underneath:
  * the encoder is `HipBertEncoder` / `HipModernBertEncoder` (one native call -> unit-norm CLS rows that stay in HBM);
  * kNN + exp/softmax scoring run on device for the whole batch (`PrototypeMemory.search_batch`);
  * the head forward is one native call for the whole batch; the blend arithmetic keeps the
    reference's Python-float (fp64) semantics but is vectorised;
  * training steps are `HeadTrainer.step` (fused fwd/bwd + EWC/clip/AdamW), losses stay on device and
    are read once per epoch (the reference syncs every step, :1507).
Each method documents where it intentionally deviates.
"""
import copy
import ctypes
import json
import logging
import math
import os
from pathlib import Path
from typing import Any, Dict, List, Optional, Set, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

from . import _native as nv
try:                                    # host-side C helper (csrc/host/hostfast.c, built by the same Makefile); optional: the
    from . import _hostfast             # Python form of the same list building stays below
except ImportError:                     # pragma: no cover
    _hostfast = None
from .ewc import EWC
from .memory import PrototypeMemory
from .models import AdaptiveHead, Example, ModelConfig
from .ops import l2_normalize_rows, softmax_rows
from .training import HeadTrainer

logger = logging.getLogger(__name__)


def select_representative_examples(examples: List[Example], k: int = 5) -> List[Example]:
    """classifier.py:1533-1573: the examples closest to the k centroids of sklearn's KMeans(n_clusters=k,
    random_state=42, n_init=10) over the L2-normalised embeddings (host side, run once per class by save(); the same
    library call as the reference, so the same examples are chosen -- one per centroid, repeats kept as it keeps them)."""
    if len(examples) <= k:
        return examples
    from sklearn.cluster import KMeans
    emb = F.normalize(torch.stack([torch.as_tensor(ex.embedding).detach().cpu().float() for ex in examples]), p=2, dim=1)
    km = KMeans(n_clusters=k, random_state=42, n_init=10)
    km.fit(emb.numpy())
    chosen = [int(torch.argmin(torch.norm(emb - c, dim=1)).item()) for c in torch.tensor(km.cluster_centers_)]
    return [examples[i] for i in chosen]


class AdaptiveClassifier:
    """A classifier that can adapt to new classes and examples (hot path on MI355X)."""

    def __init__(self, model_name: str, device: Optional[str] = None, config: Optional[Dict[str, Any]] = None,
                 seed: int = 42, use_onnx: Optional[Union[bool, str]] = "auto", trust_remote_code: bool = False,
                 *, encoder=None, tokenizer=None):
        torch.manual_seed(seed)
        self._seed = int(seed)
        self.config = ModelConfig(config)
        self.device = device or ("cuda" if torch.cuda.is_available() else "cpu")
        if not str(self.device).startswith("cuda"):
            raise nv.NativeError("AdaptiveClassifier (MI355X build) needs a GPU device; there is no CPU path. "
                                 "Use the reference package for CPU inference.")
        # use_onnx is accepted for signature compatibility; on a GPU device the reference resolves
        # "auto" to False (classifier.py:123-125) and that is the only supported value here.
        self.use_onnx = False
        if use_onnx is True:
            logger.warning("use_onnx=True ignored: ONNX Runtime is a CPU optimisation outside this build's scope")
        self.model_name = model_name
        if encoder is None:
            from transformers import AutoModel, AutoTokenizer
            from .encoder import make_encoder
            hf = AutoModel.from_pretrained(model_name, trust_remote_code=trust_remote_code)
            encoder = make_encoder(hf.eval(), device=self.device)
            encoder.config._name_or_path = model_name
            if tokenizer is None:
                tokenizer = AutoTokenizer.from_pretrained(model_name, trust_remote_code=trust_remote_code)
        self.model = encoder
        # config["gemm_arith"]: "f32" | "bf16x3" | "f16x2" (opt-in, include/acamd.h) | absent = the process-wide default.
        # PER OBJECT: it travels with every encoder call this classifier makes (encode_cls(arith=...) -> ac_bert_config.
        # gemm_arith_opt); no process-wide switch is touched, other classifiers -- even on the same encoder -- keep theirs.
        self._gemm_arith = None
        if (config or {}).get("gemm_arith") is not None:
            from .encoder import arith_mode
            self._gemm_arith = arith_mode(config["gemm_arith"])                # ValueError on an unknown name
            if self._gemm_arith == nv.AC_GEMM_F16X2 and hasattr(encoder, "enable_f16x2"):
                encoder.enable_f16x2()                                           # (builds the fp16 weight planes once; range-checked)
        if tokenizer is not None and (config or {}).get("device_tokenizer", True):
            # BERT WordPiece vocabularies are tokenised on the device (ac_wordpiece_encode); anything else stays as given
            from .tokenizer import maybe_device_tokenizer
            tokenizer = maybe_device_tokenizer(tokenizer, self.device)
        self.tokenizer = tokenizer
        self.embedding_dim = self.model.config.hidden_size
        self.memory = PrototypeMemory(self.embedding_dim, config=self.config, device=self.device)
        self.adaptive_head = None
        self.label_to_id = {}
        self.id_to_label = {}
        self.train_steps = 0
        self.training_history = {}
        if self.config.enable_strategic_mode:
            raise NotImplementedError("strategic mode (classifier.py:1594+) is outside the MI355X hot-path build")
        self.strategic_mode = False
        # "as_wired": reproduce the reference, whose EWC term in _train_new_classes is identically 0
        # (SURVEY fact 3).  "intended": penalise the live head against the pre-expansion head.
        self.ewc_mode = (config or {}).get("ewc_mode", "as_wired")
        # Where the head's dropout masks come from while training.  "device" (default): counter-based, generated inside the
        # training kernels from (classifier seed, add_examples call, step) -- no mask tensors, a whole epoch in one launch.
        # "torch_cpu": REPLAY of the reference's CPU run -- the two masks of every step are drawn on the host exactly as
        # nn.Dropout does there (`torch.empty(B, H).bernoulli_(1 - p)` on torch's global CPU generator, layer order), every other
        # draw the reference makes from that generator is made too (the Fisher pass of its as-wired EWC: ewc.py:60-64, :81), and
        # the steps run one by one through the explicit-mask kernels.  Same seeds => the reference's own training trajectory
        # (tests/test_e2e_reference_gpu.py differences it: epochs, per-epoch loss, predictions); ~10x slower per step.
        self.dropout_source = (config or {}).get("dropout_source", "device")
        if self.dropout_source not in ("device", "torch_cpu"):
            raise ValueError("config['dropout_source'] must be 'device' or 'torch_cpu', got %r" % (self.dropout_source,))
        self.last_train_info = {}
        self.train_log = []             # one entry per head training run: {"kind", "epoch_losses", "steps"} (diagnostic)

    # ------------------------------------------------------------------------------ embeddings
    def _tokenize(self, texts: List[str]):
        return self.tokenizer(texts, max_length=self.config.max_length, truncation=True, padding=True,
                              return_tensors="pt")

    def _encoder_options(self):
        """Which of encode_cls's keyword options the encoder object accepts (decided once per encoder object from its
        signature, not by catching TypeError around the call: a user-supplied encoder may take none of them)."""
        cached = getattr(self, "_enc_opts", None)
        if cached is None or cached[0] is not self.model:
            import inspect
            try:
                params = inspect.signature(self.model.encode_cls).parameters
                anykw = any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())
                opts = {name for name in ("verify", "force_layered", "arith") if anykw or name in params}
            except (TypeError, ValueError):
                opts = set()
            cached = self._enc_opts = (self.model, opts)
        return cached[1]

    def _encode_tokens(self, input_ids, token_type_ids=None, attention_mask=None, verify: bool = True,
                       force_layered: bool = False) -> torch.Tensor:
        """[b, D] unit-norm CLS embeddings on the device from token tensors (classifier.py:1267-1275 without the D2H).
        verify / force_layered: see HipBertEncoder.encode_cls -- by default a forward one of whose bounded waits gave up
        (NaN rows) is detected and repeated inside the encoder, for every caller."""
        opts = self._encoder_options()
        kw = {}
        if "verify" in opts:
            kw["verify"] = verify
        if "force_layered" in opts:
            kw["force_layered"] = force_layered
        if "arith" in opts and getattr(self, "_gemm_arith", None) is not None:
            kw["arith"] = self._gemm_arith
        return self.model.encode_cls(input_ids, token_type_ids, attention_mask, **kw)

    def _embed_device(self, texts: List[str], verify: bool = True, force_layered: bool = False) -> torch.Tensor:
        inputs = self._tokenize(texts)
        return self._encode_tokens(inputs["input_ids"], inputs.get("token_type_ids"), inputs.get("attention_mask"),
                                   verify=verify, force_layered=force_layered)

    def _get_embeddings(self, texts: List[str]) -> List[torch.Tensor]:
        """Reference signature: list of CPU [D] tensors (classifier.py:1282).  The rows are checked on the host before
        anyone can store them: the native encoders repeat a forward that gave up themselves (verify=True), so a NaN here
        means non-finite weights or a user-supplied encoder's failure -- raised, never handed to the memory."""
        emb = self._embed_device(texts).cpu()
        if not bool(torch.isfinite(emb).all()):
            raise nv.NativeError("encoder returned non-finite embeddings (%d of %d rows); nothing was stored"
                                 % (int((~torch.isfinite(emb).all(dim=1)).sum()), emb.shape[0]))
        return [e for e in emb]

    # ------------------------------------------------------------------------------ add_examples
    def add_examples(self, texts: List[str], labels: List[str]):
        if not texts or not labels:
            raise ValueError("Empty input lists")
        if len(texts) != len(labels):
            raise ValueError("Mismatched text and label lists")
        embeddings = self._get_embeddings(texts)    # (before the label maps are touched: an encoder failure leaves no trace)
        has_existing_classes = len(self.label_to_id) > 0
        new_classes = set(labels) - set(self.label_to_id.keys())
        for label in sorted(new_classes):           # alphabetical ids (classifier.py:147-150)
            idx = len(self.label_to_id)
            self.label_to_id[label] = idx
            self.id_to_label[idx] = label
        self.add_embeddings(texts, embeddings, labels, _maps_done=True, _new_classes=new_classes,
                            _has_existing=has_existing_classes)

    def add_embeddings(self, texts: List[str], embeddings, labels: List[str], _maps_done=False, _new_classes=None,
                       _has_existing=None):
        """add_examples() after the encoder call: memory update + head training (classifier.py:155-200).
        Public so that pre-computed embeddings can be fed (BASELINE configs[3])."""
        if not _maps_done:
            if not texts or not labels:
                raise ValueError("Empty input lists")
            if len(texts) != len(labels):
                raise ValueError("Mismatched text and label lists")
            _has_existing = len(self.label_to_id) > 0
            _new_classes = set(labels) - set(self.label_to_id.keys())
            for label in sorted(_new_classes):
                idx = len(self.label_to_id)
                self.label_to_id[label] = idx
                self.id_to_label[idx] = label
        is_adding_new_classes = len(_new_classes) > 0
        # one NaN row would poison its class's prototype and fp64 running sum for good (memory.py:149-150 averages everything
        # stored): refuse before anything is mutated (label maps of a refused call are rolled back)
        try:
            bad = (~torch.isfinite(torch.stack([torch.as_tensor(e) for e in embeddings])).all(dim=1)).nonzero().flatten().tolist()
        except (RuntimeError, TypeError, ValueError):      # ragged or missing embeddings: the memory's own validation reports those
            bad = [i for i, e in enumerate(embeddings) if e is not None and not bool(torch.isfinite(torch.as_tensor(e)).all())]
        if bad:
            for label in _new_classes:                    # (the new labels took the highest ids: popping them restores the maps)
                idx = self.label_to_id.pop(label, None)
                if idx is not None:
                    self.id_to_label.pop(idx, None)
            raise ValueError("non-finite embedding for example(s) %s: nothing was added" % bad[:8])
        # memory update for the whole call: sequential add_example semantics, the per-class prune loop on the device
        self.memory.add_examples_batch([Example(t, l, e) for t, e, l in zip(texts, embeddings, labels)], list(labels))
        for label in labels:
            self.training_history[label] = self.training_history.get(label, 0) + 1
        if is_adding_new_classes and _has_existing:
            old_head = copy.deepcopy(self.adaptive_head) if self.adaptive_head is not None else None
            self.adaptive_head.update_num_classes(len(self.label_to_id))
            self.adaptive_head = self.adaptive_head.to(self.device)
            self._train_new_classes(old_head, _new_classes)
        else:
            if self.adaptive_head is None:
                self._initialize_adaptive_head()
            elif is_adding_new_classes:
                self.adaptive_head.update_num_classes(len(self.label_to_id))
                self.adaptive_head = self.adaptive_head.to(self.device)
            self._train_adaptive_head()
        self.memory._rebuild_index()                 # classifier.py:200

    def _initialize_adaptive_head(self):
        hidden_dims = [self.embedding_dim, self.embedding_dim // 2]
        self.adaptive_head = AdaptiveHead(self.embedding_dim, len(self.label_to_id), hidden_dims=hidden_dims).to(self.device)

    # ------------------------------------------------------------------------------ training loops
    @staticmethod
    def _index_loader(n, batch_size):
        """Batches of example indices in exactly the order the reference's
        DataLoader(shuffle=True, generator=torch.Generator().manual_seed(42)) yields (classifier.py:1454-1459).
        Kept as the executable definition of that order (tests compare `_EpochOrder` against it)."""
        ds = torch.utils.data.TensorDataset(torch.arange(n))
        return torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=True,
                                           generator=torch.Generator().manual_seed(42))

    class _EpochOrder:
        """The index order of successive epochs of that DataLoader without iterating one (which costs ~150 us of
        host time per 32-example batch -- more than the training step itself).  It consumes the generator exactly
        as torch does per epoch: `_BaseDataLoaderIter.__init__` draws the base seed, `RandomSampler.__iter__`
        draws `randperm(n)` for the epoch and, when the iterator is exhausted, one more `randperm(n)` for the
        empty `num_samples % n` tail.  tests/test_host_logic.py pins this against a real DataLoader."""

        def __init__(self, n, seed=42):
            self.n = n
            self.g = torch.Generator().manual_seed(seed)

        def next_epoch(self):
            torch.empty((), dtype=torch.int64).random_(generator=self.g)
            order = torch.randperm(self.n, generator=self.g)
            torch.randperm(self.n, generator=self.g)
            return order

    LOSS_KIND = 0            # AC_LOSS_CE; the multi-label subclass trains new classes with CE on sigmoid outputs

    def _run_epochs(self, X, y, batch_size, epochs, use_scheduler, ewc=None, lambda_B=None, loss_kind=None,
                    targets=None):
        """Shared epoch loop: native steps, device-side loss accumulation, one sync per epoch."""
        head = self.adaptive_head
        head.train()
        trainer = HeadTrainer(head, lr=0.001, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=1.0)
        sched = None
        if use_scheduler:        # ReduceLROnPlateau(mode=min, factor .5, patience 2) on a stand-in optimizer
            dummy = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=trainer.lr)
            sched = torch.optim.lr_scheduler.ReduceLROnPlateau(dummy, mode="min", factor=0.5, patience=2)
        n_rows = X.shape[0]
        epoch_order = self._EpochOrder(n_rows)
        steps_per_epoch = (n_rows + batch_size - 1) // batch_size          # len(loader), drop_last=False
        best_loss, patience, patience_counter = float("inf"), 3, 0
        steps = 0
        X = X.to(device=self.device, dtype=torch.float32).contiguous()     # the kernels read fp32 rows
        y = None if y is None else y.to(device=self.device, dtype=torch.int64).contiguous()
        if targets is not None:
            targets = targets.to(device=self.device, dtype=torch.float32).contiguous()
        loss_kind = self.LOSS_KIND if loss_kind is None else loss_kind
        # dropout masks are generated in-kernel from (classifier seed, add_examples call, step): counter-based and
        # reproducible per classifier seed, independent of later torch.manual_seed calls (AdaptiveHead.__init__ itself
        # reseeds torch's global generator to 42); the reference draws them from that global generator (not replayable)
        base_seed = (self._seed * 0x9E3779B97F4A7C15 + self.train_steps * 1000003) & 0x7FFFFFFFFFFFFFFF
        replay = getattr(self, "dropout_source", "device") == "torch_cpu"
        epoch_losses = []
        for epoch in range(epochs):
            trainer.loss_accum.zero_()
            if replay:
                done = self._replay_epoch(trainer, X, y, targets, epoch_order.next_epoch(), batch_size, ewc, lambda_B, loss_kind)
                avg_loss = float(trainer.loss_accum.item()) / steps_per_epoch
                steps += done
                epoch_losses.append(avg_loss)
                if sched is not None:
                    sched.step(avg_loss)
                    trainer.lr = dummy.param_groups[0]["lr"]
                if avg_loss < best_loss:
                    best_loss, patience_counter = avg_loss, 0
                else:
                    patience_counter += 1
                    if patience_counter >= patience:
                        break
                continue
            # one H2D of the epoch's batch order (same order as the reference's seeded DataLoader); every batch
            # but the last has batch_size rows (drop_last=False): one native call runs the whole epoch
            order = epoch_order.next_epoch().to(X.device)
            nb0 = min(batch_size, n_rows)
            # rows laid out in epoch order once (one device gather per epoch instead of one gather launch per step)
            Xe = X.index_select(0, order)
            ye = None if y is None else y.index_select(0, order)
            te = None if targets is None else targets.index_select(0, order)
            def run_epoch(stepwise=False):
                return trainer.fused_epoch(Xe, ye, None, nb0, AdaptiveHead.DROPOUT_P, base_seed + steps,
                                           fisher=None if ewc is None else ewc.fisher_flat,
                                           old_params=None if ewc is None else ewc.old_flat,
                                           lambda_B=0.0 if ewc is None else lambda_B, loss_kind=loss_kind,
                                           targets_all=te, stepwise=stepwise)
            done = run_epoch()
            avg_loss = float(trainer.loss_accum.item()) / steps_per_epoch    # the only host sync of the epoch
            if not math.isfinite(avg_loss) and (nv.lib().ac_set_persistent_kernels(-1) & 1):
                # a NaN epoch loss with the persistent kernel on: either the run diverged or a grid barrier gave up (device
                # shared with another process).  Put parameters and moments back and repeat the epoch launch by launch, once
                # (per-call flag: no process-wide switch is touched).
                logger.warning("training epoch returned NaN through the persistent kernel; repeating it with the step-by-step launches")
                trainer.restore_epoch()
                trainer.loss_accum.zero_()
                done = run_epoch(stepwise=True)
                avg_loss = float(trainer.loss_accum.item()) / steps_per_epoch
            steps += done
            epoch_losses.append(avg_loss)
            if sched is not None:
                sched.step(avg_loss)
                trainer.lr = dummy.param_groups[0]["lr"]
            if avg_loss < best_loss:
                best_loss, patience_counter = avg_loss, 0
            else:
                patience_counter += 1
                if patience_counter >= patience:
                    logger.debug(f"Early stopping at epoch {epoch + 1}")
                    break
        self.last_train_info = {"steps": steps, "epochs": epoch + 1, "final_loss": avg_loss}
        self.train_log.append({"epoch_losses": epoch_losses, "steps": steps, "rows": int(n_rows)})
        if not math.isfinite(avg_loss):
            # (the reference would hand out NaN scores silently after a diverged training run; say so once, where it happened)
            logger.warning("head training ended with a non-finite loss after %d steps: the head's probabilities will be NaN "
                           "until the next add_examples() retrains it (non-finite embeddings are refused earlier, so this is a "
                           "diverged run or a defect)", steps)
        self.train_steps += 1

    def _replay_epoch(self, trainer, X, y, targets, order, batch_size, ewc, lambda_B, loss_kind):
        """One epoch of config['dropout_source'] == 'torch_cpu': the reference's step loop (classifier.py:1483-1507, :327-353)
        with ITS dropout masks.  nn.Dropout on a CPU tensor is `noise = empty_like(x).bernoulli_(1 - p)` on the global generator
        (aten/src/ATen/native/Dropout.cpp, the non-fused path), first for the [B, H1] activation, then for [B, H2]; the same two
        calls here, in the same order, leave the generator where the reference's step leaves it.  The step itself is the
        explicit-mask kernel pair (ac_head_fwd_bwd_ce / _loss + ac_ewc_adamw_step) whose single steps tests/golden/head_step.json
        pins against the reference."""
        from .training import LOSS_CE
        n = X.shape[0]
        p = AdaptiveHead.DROPOUT_P
        H1, H2 = trainer.dims.H1, trainer.dims.H2
        done = 0
        for i0 in range(0, n, batch_size):
            idx = order[i0:i0 + batch_size].to(X.device)
            B = int(idx.numel())
            m1 = torch.empty(B, H1).bernoulli_(1 - p).to(torch.uint8).to(X.device)
            m2 = torch.empty(B, H2).bernoulli_(1 - p).to(torch.uint8).to(X.device)
            Xb = X.index_select(0, idx)
            yb = None if y is None else y.index_select(0, idx)
            tb = None if targets is None else targets.index_select(0, idx)
            if loss_kind == LOSS_CE and tb is None:
                trainer.forward_backward(Xb, yb, m1, m2, p)
            else:
                trainer.forward_backward_loss(Xb, y=yb, targets=tb, loss_kind=loss_kind, mask1=m1, mask2=m2, dropout_p=p)
            out = trainer.optimizer_step(None if ewc is None else ewc.fisher_flat, None if ewc is None else ewc.old_flat,
                                         0.0 if ewc is None else lambda_B / B)
            trainer.loss_accum += trainer.loss                       # (device side; one host sync per epoch as elsewhere)
            if ewc is not None:
                trainer.loss_accum += out[0:1]                       # the penalty is part of the reference's logged loss (:338-341)
            done += 1
        return done

    @staticmethod
    def _replay_fisher_rng(n_old, c_old):
        """config['dropout_source'] == 'torch_cpu' in as-wired mode: the reference builds EWC(old_head, old_dataset) here
        (classifier.py:273-300) although its penalty is identically zero, and that constructor's Fisher pass DRAWS from
        torch's global generator: a DataLoader(batch_size=32, shuffle=True) without a generator of its own (base seed + the
        sampler's seed, ewc.py:60-64) and one torch.multinomial(probs [B, C_old], 1) per batch (:81 -- an exponential_ over B x C
        elements whatever the probabilities are).  The same objects and calls on a dummy dataset of the same length: the
        generator ends where the reference's does, nothing else is computed."""
        ds = torch.utils.data.TensorDataset(torch.zeros(n_old, 1), torch.zeros(n_old, dtype=torch.long))
        for xb, _ in torch.utils.data.DataLoader(ds, batch_size=32, shuffle=True):
            torch.multinomial(torch.full((xb.shape[0], c_old), 1.0 / c_old), 1)

    def _train_adaptive_head(self, epochs: int = 10):
        """classifier.py:1428-1522: retrain on everything stored, sorted by (label, text)."""
        if not self.memory.examples:
            return
        # The training matrix is assembled ON THE DEVICE from the memory's device-resident class matrices: per call only
        # the row permutation (sorted labels, then sorted texts) crosses PCIe, not the stored embeddings.
        dev = torch.device(self.device)
        mats, perm, labs, off = [], [], [], 0
        for label in sorted(self.memory.examples.keys()):
            exs = self.memory.examples[label]
            n = len(exs)
            if not n:
                continue
            order = sorted(range(n), key=lambda i: exs[i].text)                  # stable, like sorted(examples, key=text)
            mats.append(self.memory.device_class_matrix(label, dev)[0][:n])
            perm.extend(off + i for i in order)
            off += n
            labs.extend([self.label_to_id[label]] * n)
        allm = mats[0] if len(mats) == 1 else torch.cat(mats)
        X = l2_normalize_rows(allm[torch.tensor(perm, dtype=torch.long).to(dev)])                     # :1450
        y = torch.tensor(labs, dtype=torch.long, device=self.device)
        self._run_epochs(X, y, batch_size=min(32, X.shape[0]), epochs=epochs, use_scheduler=True)

    def _train_new_classes(self, old_head, new_classes: Set[str]):
        """classifier.py:202-367: resample stored examples (numpy global RNG, same call sequence),
        EWC(lambda=5.0) built on the pre-expansion head, <= 15 epochs of CE (+ EWC) steps."""
        if not self.memory.examples:
            return
        all_embeddings, all_labels = [], []
        examples_per_class = {label: len(ex) for label, ex in self.memory.examples.items()}
        min_examples = min(examples_per_class.values())
        num_classes = len(examples_per_class)
        target = max(5, min(10, min_examples * 2))
        if num_classes > 20:                                   # :224-246
            for label, examples in self.memory.examples.items():
                num_samples = min(len(examples), target * 2 if label in new_classes else target)
                indices = np.random.choice(len(examples), size=num_samples, replace=num_samples > len(examples))
                for i in indices:
                    all_embeddings.append(examples[i].embedding)
                    all_labels.append(self.label_to_id[label])
        else:                                                  # :247-271
            for label, examples in self.memory.examples.items():
                weight = 2.0 if label in new_classes else min_examples / examples_per_class[label]
                num_samples = max(min_examples, int(len(examples) * weight))
                indices = np.random.choice(len(examples), size=num_samples, replace=num_samples > len(examples))
                for i in indices:
                    all_embeddings.append(examples[i].embedding)
                    all_labels.append(self.label_to_id[label])
        X = torch.stack([e.to(torch.float32) for e in all_embeddings]).to(self.device)
        y = torch.tensor(all_labels, dtype=torch.long, device=self.device)

        ewc = None
        if old_head is not None and self.ewc_mode == "intended":
            old_X, old_y = [], []
            old_label_to_id = {label: idx for idx, label in enumerate(self.id_to_label.values())
                               if label not in new_classes}
            for label, examples in self.memory.examples.items():
                if label not in new_classes:
                    for example in examples[:5]:
                        old_X.append(example.embedding)
                        old_y.append(old_label_to_id[label])
            if old_X:
                ds = torch.utils.data.TensorDataset(torch.stack([e.to(torch.float32) for e in old_X]),
                                                    torch.tensor(old_y, dtype=torch.long))
                ewc = self._expand_ewc(EWC(old_head, ds, device=self.device, ewc_lambda=5.0))
        # "as_wired": the reference's penalty is built on a frozen copy and is exactly 0 with no
        # gradient into the trained head, i.e. plain CE + AdamW -- which is what runs.
        if ewc is None and old_head is not None and getattr(self, "dropout_source", "device") == "torch_cpu":
            n_old = sum(min(5, len(ex)) for label, ex in self.memory.examples.items() if label not in new_classes)
            c_old = sum(1 for label in self.id_to_label.values() if label not in new_classes)
            if n_old:
                self._replay_fisher_rng(n_old, c_old)     # the generator draws of the reference's (ineffective) Fisher pass
        self._run_epochs(X, y, batch_size=32, epochs=15, use_scheduler=False, ewc=ewc, lambda_B=5.0)

    def _expand_ewc(self, ewc):
        """Lay Fisher / old params of the pre-expansion head out over the expanded head's flat block
        (new output rows get F = 0)."""
        head = self.adaptive_head
        flat = head.flat_params()
        fisher = torch.zeros_like(flat)
        old = flat.detach().clone()
        off_new, off_old = 0, 0
        old_head = ewc.model
        for p_new, p_old in zip(head.parameters(), old_head.parameters()):
            n_new, n_old = p_new.numel(), p_old.numel()
            fisher[off_new: off_new + n_old] = ewc.fisher_flat[off_old: off_old + n_old]
            old[off_new: off_new + n_old] = ewc.old_flat[off_old: off_old + n_old]
            off_new += n_new
            off_old += n_old
        ewc.fisher_flat, ewc.old_flat = fisher, old
        return ewc

    # ------------------------------------------------------------------------------ prediction
    def _device_stage(self, emb: torch.Tensor, k_proto: int):
        """Device stage shared by predict / predict_batch: kNN scores + head probabilities, left on the device.
        Returns (proto_scores [b,kp] f32, proto_class [b,kp] i64 with -1 for padding / unknown labels,
        head_probs [b,C] f32) -- each None when that source does not exist."""
        with torch.no_grad():
            S = Cid = probs = None
            if self.memory.index.ntotal > 0 or self.memory.updates_since_rebuild >= self.config.prototype_update_frequency:
                S, I, _ = self.memory.search_batch(emb, k_proto)
                Cid = self.memory.hit_class_ids(I, self.label_to_id)
            if self.adaptive_head is not None:
                self.adaptive_head.eval()
                probs = softmax_rows(self._head_outputs(emb))
        return S, Cid, probs

    def _head_outputs(self, emb: torch.Tensor) -> torch.Tensor:
        """What `self.adaptive_head(x)` returns in the reference before its F.softmax (classifier.py:432-435,
        :1342-1345): the logits for AdaptiveHead.  The multi-label subclass overrides this (sigmoid outputs)."""
        return self.adaptive_head.forward_native(emb)

    def _device_scores(self, emb: torch.Tensor, k_proto: int):
        """_device_stage copied to the host as numpy (the inputs of the numpy formula `_blend`)."""
        S, Cid, probs = self._device_stage(emb, k_proto)
        return (None if S is None else S.cpu().numpy(), None if S is None else Cid.cpu().numpy(),
                None if probs is None else probs.cpu().numpy())

    _BLEND_DEVICE_MAX_CLASSES = 2048

    def _blend_weights(self, regular: bool):
        """Per-class (prototype, head) weights as fp64 device vectors: 0.7 / 0.3 for predict_batch (:1359-1384);
        by training history for predict (:447-480: < 10 examples -> 0.3 / 0.7)."""
        C = len(self.id_to_label)
        if regular:
            hist = [self.training_history.get(self.id_to_label[c], 0) for c in range(C)]
            key = ("r", tuple(h < 10 for h in hist))
        else:
            key = ("b", C)
        cached = getattr(self, "_blend_w_cache", None)
        if cached is None or cached[0] != key:
            if regular:
                wp = [0.3 if h < 10 else 0.7 for h in hist]
                wh = [0.7 if h < 10 else 0.3 for h in hist]
            else:
                wp, wh = [0.7] * C, [0.3] * C
            w = torch.tensor([wp, wh], dtype=torch.float64, device=self.device)
            self._blend_w_cache = cached = (key, w)
        return cached[1]

    def _blend_device(self, S, Cid, P, k: int, regular: bool):
        """ac_blend_topk over one device stage -> (packed device buffer, layout).  No host synchronisation.
        packed: n[b] i32 | class[b,kk] i32 | score[b,kk] f64 (8-byte aligned)."""
        C = len(self.id_to_label)
        b = (S if S is not None else P).shape[0]
        kp = 0 if S is None else S.shape[1]
        kk = max(1, min(k, C))
        w = self._blend_weights(regular)
        ncls = C if regular else min(k, C)
        off_cls = 4 * b
        off_val = (off_cls + 4 * b * kk + 7) // 8 * 8
        out = torch.empty(off_val + 8 * b * kk, dtype=torch.uint8, device=self.device)
        base = out.data_ptr()
        if S is not None:
            S, Cid = S.contiguous(), Cid.contiguous()
        if P is not None:
            P = P.contiguous()
        with torch.cuda.device(out.device):
            nv.check(nv.lib().ac_blend_topk(None if S is None else S.data_ptr(), None if S is None else Cid.data_ptr(), kp,
                                            None if P is None else P.data_ptr(), C, w[0].data_ptr(), w[1].data_ptr(),
                                            ncls, kk, b, base, base + off_cls, base + off_val,
                                            nv.stream_ptr(out.device)), "ac_blend_topk")
        return out, (b, kk, off_cls, off_val, C)

    _last_nan = False

    def _unpack(self, host, layout, k: int, check: bool = True):
        """The packed result of _blend_device (already on the host, numpy uint8) -> list of (label, score) per query.
        NaN scores: noted in self._last_nan; check=True additionally asks the encoder whether it gave up and raises if so."""
        b, kk, off_cls, off_val, C = layout
        kcap = min(kk, k) if k >= 0 else 0
        if _hostfast is not None:                                # one C pass (csrc/host/hostfast.c): same lists, ~5x faster
            names = getattr(self, "_label_tuple", None)
            if names is None or names[0] is not self.id_to_label or len(names[1]) != C or (C <= 64 and names[2] != tuple(self.id_to_label.values())):
                names = self._label_tuple = (self.id_to_label, tuple(self.id_to_label[c] for c in range(C)), tuple(self.id_to_label.values()))
            res, self._last_nan = _hostfast.unpack(names[1], host, b, kk, off_cls, off_val, kcap)
            if self._last_nan and check:
                self._raise_if_encoder_gave_up()
            return res
        n = host[:off_cls].view(np.int32)
        cls = host[off_cls:off_cls + 4 * b * kk].view(np.int32)
        val = host[off_val:].view(np.float64)
        self._last_nan = bool(np.isnan(val).any())
        if self._last_nan and check:
            self._raise_if_encoder_gave_up()
        names = getattr(self, "_label_array", None)              # (kept for big label sets only: small ones are rebuilt per call)
        if C <= 64 or names is None or names[0] is not self.id_to_label or len(names[1]) != C:
            names = self._label_array = (self.id_to_label, np.array([self.id_to_label[c] for c in range(C)], dtype=object))
        # one pass over the flat arrays (a tuple per hit is what the reference returns); rows are slices of it
        pairs = list(zip(names[1][np.clip(cls, 0, C - 1)].tolist(), val.tolist()))
        if int(n.min()) >= kcap:
            return [pairs[i:i + kcap] for i in range(0, b * kk, kk)]
        n = n.tolist()
        return [pairs[q * kk:q * kk + min(n[q], kcap)] for q in range(b)]

    def _finish(self, S, Cid, P, k: int, regular: bool, b: int = 0, check: bool = True):
        """Blend + normalise + top-k of one device stage -> the reference's list of (label, score) per query.
        On the device (ac_blend_topk, one packed D2H) for up to 2048 classes; the numpy formula beyond."""
        C = len(self.id_to_label)
        self._last_nan = False
        if S is None and P is None:
            return [[] for _ in range(b)]
        if C < 1 or C > self._BLEND_DEVICE_MAX_CLASSES:
            Sh, Ph = None if S is None else S.cpu().numpy(), None if P is None else P.cpu().numpy()
            self._last_nan = bool((Sh is not None and np.isnan(Sh).any()) or (Ph is not None and np.isnan(Ph).any()))
            if self._last_nan and check:
                self._raise_if_encoder_gave_up()
            return self._blend(Sh, None if S is None else Cid.cpu().numpy(), Ph, k, regular)
        out, layout = self._blend_device(S, Cid, P, k, regular)
        return self._unpack(out.cpu().numpy(), layout, k, check)

    def _raise_if_encoder_gave_up(self):
        """NaN scores from embeddings this object did not compute itself (predict_embeddings): if the encoder's fused-LayerNorm
        GEMM epilogues timed out (a device that cannot hold one workgroup per CU at once, e.g. under a CU mask; include/acamd.h
        ac_bert_ln_fusion_status) say so loudly and switch the fusion off.  (predict / predict_batch / predict_tokens own their
        encoder call and repeat it instead: _predict_with_retry.)"""
        gave_up = getattr(self.model, "ln_fusion_aborted", None)
        if gave_up is not None and gave_up():
            self._ln_fusion_off()
            raise nv.NativeError("encoder: the fused LayerNorm epilogue gave up waiting for the tiles of a row panel "
                                 "(device shared or CU-masked?); the fusion is now off for this encoder -- repeat the call")

    def _ln_fusion_off(self):
        """LayerNorm fusion off for this classifier's encoder -- per object, always: the process-wide switches of the ABI are test
        hooks (include/acamd.h) and are never touched by the product.  (An encoder object without the option is not one of the
        native encoders and runs no such kernel.)"""
        off = getattr(self.model, "disable_ln_fusion", None)
        if off is not None:
            off()

    def predict(self, text: str, k: int = 5) -> List[Tuple[str, float]]:
        if not text:
            raise ValueError("Empty input text")
        return self._predict_regular(text, k)

    def _predict_with_retry(self, encode, finish):
        """The predict paths' contract with the encoder's two bounded waits (HipBertEncoder.encode_cls): run the chain with
        verify=False -- no host synchronisation between the encoder and the result copy, the GPU queue stays full -- and look
        at the scores that come to the host anyway.  NaN scores are then traced to their cause and the chain is repeated
        without the kernel that gave up; the caller never sees the NaNs and no exception is raised.
        encode(verify, force_layered) -> embeddings; finish(emb) -> (result, has_nan)."""
        res, nan = finish(encode(False, False))
        if not nan:
            return res
        if getattr(self.model, "last_one_launch", False):
            # NaN scores after the one-launch encoder: one of its grid barriers gave up (a device shared with another compute
            # process poisons the embedding with NaNs rather than hanging).  Repeat through the layer-by-layer kernels.
            logger.warning("single-query encoder returned NaN through the persistent kernel; repeating layer by layer")
            res, nan = finish(encode(True, True))
        else:
            gave_up = getattr(self.model, "ln_fusion_aborted", None)
            if gave_up is not None and gave_up():
                logger.warning("encoder: a fused LayerNorm epilogue gave up (device shared or CU-masked?); LayerNorm fusion is "
                               "now off for this encoder and the batch is encoded again")
                self._ln_fusion_off()
                res, nan = finish(encode(True, False))
            f16 = getattr(self.model, "f16x2_active", None)
            if nan and f16 is not None and self._f16x2_active():
                # opt-in fp16x2 arithmetic: an activation beyond fp16's range turns its rows into NaN; back to bf16x3
                logger.warning("encoder: non-finite result under fp16x2 arithmetic; this classifier goes back to bf16x3 and the "
                               "batch is encoded again")
                self.model.f16x2_overflows += 1
                if getattr(self, "_gemm_arith", None) == nv.AC_GEMM_F16X2:
                    self._gemm_arith = nv.AC_GEMM_BF16X3      # per object: other classifiers on this encoder keep their fp16x2
                elif hasattr(self.model, "set_arith"):
                    self.model.set_arith(nv.AC_GEMM_BF16X3)   # the choice was the encoder's (or the process's): the encoder's now
                else:
                    self.model.disable_f16x2()                # (an encoder object with the older interface)
                res, nan = finish(encode(True, False))
        return res                      # (still NaN: non-finite inputs or weights -- the caller's data, returned as computed)

    def _f16x2_active(self) -> bool:
        f16 = getattr(self.model, "f16x2_active", None)
        if f16 is None:
            return False
        try:
            return bool(f16(getattr(self, "_gemm_arith", None)))
        except TypeError:                       # (an encoder object with the older zero-argument form)
            return bool(f16())

    def _predict_regular(self, text: str, k: int = 5) -> List[Tuple[str, float]]:
        """classifier.py:415-480: prototype scores over ALL classes, head probs over ALL classes,
        history-keyed weights (0.3/0.7 vs 0.7/0.3), stable sort, normalise, top-k.
        (Replaying this ~90-kernel single-query chain as one captured HIP graph was measured and dropped: 1.013 vs
        1.014 ms -- the chain is paced by the GPU's dependent-dispatch interval, not by host enqueue)"""
        def encode(verify, force_layered):
            return self._embed_device([text], verify=verify, force_layered=force_layered)

        def finish(emb):
            max_classes = len(self.id_to_label) if self.id_to_label else k
            res = self._finish_from_embeddings(emb, k, regular=True, check=False, k_proto=max_classes)[0]
            return res, any(s != s for _, s in res)
        return self._predict_with_retry(encode, finish)

    def predict_batch(self, texts: List[str], k: int = 5, batch_size: Optional[int] = None) -> List[List[Tuple[str, float]]]:
        """classifier.py:1308-1388: top-k prototypes + top-k head classes, fixed 0.7/0.3 weights.
        `batch_size` is the UPPER bound on the texts per encoder call, as in the reference (which defaults it to 32 to bound CPU
        memory).  Left at None, the device default applies: config["min_device_batch"] (256, what BASELINE configs[1] is quoted
        on) -- 32 texts are ~600 token rows, a launch-bound encoder call.  A text's result does not depend on which texts share
        its batch beyond fp32 rounding of the encoder's sums (padding-free forward, exact search; the reference's own batched
        matmuls are batch-dependent in the same way)."""
        if not texts:
            raise ValueError("Empty input batch")
        out = []
        batch_size = max(int(self.config.config.get("min_device_batch", 256)), 1) if batch_size is None else max(int(batch_size), 1)
        for i in range(0, len(texts), batch_size):
            batch = texts[i:i + batch_size]
            out.extend(self._predict_batch_core(lambda verify, force_layered: self._embed_device(
                batch, verify=verify, force_layered=force_layered), k))
        return out

    def predict_tokens(self, input_ids, token_type_ids=None, attention_mask=None, k: int = 5) -> List[List[Tuple[str, float]]]:
        """One predict_batch() batch after the tokenizer (classifier.py:1267-1384): encoder -> device kNN + head -> blend
        -> one packed D2H -> the reference's lists.  Public for pre-tokenised input (the benchmark's form).  An encoder
        forward whose bounded wait gave up is repeated transparently (_predict_with_retry)."""
        return self._predict_batch_core(lambda verify, force_layered: self._encode_tokens(
            input_ids, token_type_ids, attention_mask, verify=verify, force_layered=force_layered), k)

    def _predict_batch_core(self, encode, k):
        def finish(emb):
            res = self._finish_from_embeddings(emb, k, regular=False, check=False)
            return res, self._last_nan
        return self._predict_with_retry(encode, finish)

    # ---- the tail of a batch in one launch + no copy (include/acamd.h ac_predict_post) ------------------------------------------
    _POST_MAX_HITS = 1024

    def _post_buffer(self, nbytes: int):
        """This classifier's host-mapped result buffer (ac_host_alloc), grown as needed: (address, numpy uint8 view)."""
        buf = getattr(self, "_post_buf", None)
        if buf is None or buf[2] < nbytes:
            if buf is not None:
                nv.lib().ac_host_free(buf[0])
            size = max(1 << 16, 2 * nbytes)
            p = ctypes.c_void_p()
            nv.check(nv.lib().ac_host_alloc(size, ctypes.byref(p)), "ac_host_alloc")
            view = np.frombuffer((ctypes.c_ubyte * size).from_address(p.value), dtype=np.uint8)
            buf = self._post_buf = (p, view, size)
        return buf[0], buf[1]

    def __del__(self):
        buf = getattr(self, "_post_buf", None)
        if buf is not None:
            try:
                self._post_buf = None
                nv.lib().ac_host_free(buf[0])
            except Exception:       # interpreter shutdown
                pass

    def _finish_from_embeddings(self, emb: torch.Tensor, k: int, regular: bool, check: bool = True, k_proto: Optional[int] = None):
        """Embeddings -> the reference's list of (label, score) per query.  One native call after the search and the head's
        forward (ac_predict_post: prototype scores, hit classes, F.softmax, blend, top-k, the packed result written straight
        into host-mapped memory and waited for without a stream synchronisation); the separate kernels + a D2H copy
        (_device_stage / _finish) beyond its limits (2048 classes, 1024 hits per query) or with AC_PREDICT_POST=0."""
        C = len(self.id_to_label)
        b = emb.shape[0]
        k_proto = k if k_proto is None else k_proto
        mode = os.environ.get("AC_PREDICT_POST", "1")
        if (mode == "0" or C < 1 or C > self._BLEND_DEVICE_MAX_CLASSES or b == 0
                or k_proto > self._POST_MAX_HITS or not emb.is_cuda):
            S, Cid, P = self._device_stage(emb, k_proto)
            return self._finish(S, Cid, P, k, regular=regular, b=b, check=check)
        self._last_nan = False
        with torch.no_grad():
            D = I = head = None
            cmap = (None, 0, None, 0)
            if self.memory.index.ntotal > 0 or self.memory.updates_since_rebuild >= self.config.prototype_update_frequency:
                D, I = self.memory.search_raw(emb, k_proto)
                D, I = D.to(torch.float32).contiguous(), I.to(torch.int64).contiguous()
                cmap = self.memory.class_map(self.label_to_id, emb.device)
            # (the head's three launches on a side stream under the search's seven: overlapped in the trace, 0.04 ms SLOWER per step --
            #  two cross-stream waits cost more than 25 us of small kernels; profiles/r06/head_side_stream.txt)
            if self.adaptive_head is not None:
                self.adaptive_head.eval()
                head = self._head_outputs(emb).contiguous()
        if D is None and head is None:
            return [[] for _ in range(b)]
        stage, view, need, layout = self._post_launch(D, I, cmap, head, k, regular, host=mode != "2")
        if mode == "2":                 # (A/B: the fused kernel with an ordinary D2H copy of its result)
            return self._unpack(stage[:need].cpu().numpy(), layout, k, check)
        return self._unpack(view[:need], layout, k, check)

    def _post_launch(self, D, I, cmap, head, k: int, regular: bool, host: bool = True):
        """ac_predict_post over one batch's search result (D, I: may be None), class map and head outputs (may be None).
        host=True: returns when the packed result is readable in this classifier's host-mapped buffer (no stream
        synchronisation); host=False: asynchronous, the result stays in the device buffer.
        -> (device buffer, host view, bytes, layout for _unpack)."""
        C = len(self.id_to_label)
        dev = (D if D is not None else head).device
        b = (D if D is not None else head).shape[0]
        kp = 0 if D is None else D.shape[1]
        kk = max(1, min(k, C))
        w = self._blend_weights(regular)
        ncls = C if regular else min(k, C)
        off_cls = 4 * b
        off_val = (off_cls + 4 * b * kk + 7) // 8 * 8
        need = (off_val + 8 * b * kk + 15) // 16 * 16
        addr, view = self._post_buffer(need) if host else (None, None)
        stage = getattr(self, "_post_stage", None)
        if stage is None or stage.numel() < need or stage.device != dev:
            stage = self._post_stage = torch.empty(max(1 << 16, 2 * need), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            nv.check(nv.lib().ac_predict_post(nv.ptr(D), nv.ptr(I), kp, nv.ptr(cmap[0]), cmap[1], nv.ptr(cmap[2]), cmap[3],
                                              nv.ptr(head), C, 1, w[0].data_ptr(), w[1].data_ptr(), ncls, kk, b, nv.ptr(stage), need,
                                              addr, nv.stream_ptr(dev)), "ac_predict_post")
        return stage, view, need, (b, kk, off_cls, off_val, C)

    def predict_embeddings(self, emb: torch.Tensor, k: int = 5) -> List[List[Tuple[str, float]]]:
        """predict_batch() after the encoder: device kNN + head, then the blend of :1359-1384."""
        return self._finish_from_embeddings(emb, k, regular=False)

    def _blend(self, S, Cid, P, k, regular):
        """The two score-combination formulas of the reference, evaluated in fp64 like its Python floats,
        vectorised over the batch (no per-query device access).

        regular=True  (predict, :447-480):        weights by training_history (<10 -> proto .3 / head .7, else
                                                  .7 / .3), head over ALL classes.
        regular=False (predict_batch, :1359-1384): proto .7 / head .3, head over its top min(k, C) classes only.
        Ordering: descending score, ties in insertion order (prototype hits in distance order first, then
        head classes in descending probability) -- Python's stable sort in the reference.
        S [b,kp] scores, Cid [b,kp] class id of each hit (-1 = none), P [b,C] head probabilities.
        """
        C = len(self.id_to_label)
        b = (S if S is not None else P).shape[0]
        if regular:
            hist = np.array([self.training_history.get(self.id_to_label[c], 0) for c in range(C)])
            wp = np.where(hist < 10, 0.3, 0.7)
            wh = np.where(hist < 10, 0.7, 0.3)
        else:
            wp = np.full(C, 0.7)
            wh = np.full(C, 0.3)
        combined = np.zeros((b, C), dtype=np.float64)
        BIG = 1 << 30
        ins = np.full((b, C), BIG, dtype=np.int64)          # insertion rank, for stable tie order
        kp = 0
        if S is not None:
            kp = S.shape[1]
            S64 = S.astype(np.float64)
            jidx = np.arange(kp, dtype=np.int64)[None, :]
            # one pass per class (C is small next to b * kp).  With one row per class each class occurs
            # once per query, so summing == the reference's dict assignment; with the generalised store
            # (M6, several rows per class) the hits of a class vote (sum).
            if C <= 64:
                for c in range(C):
                    hit = Cid == c
                    if hit.any():
                        combined[:, c] = np.where(hit, S64, 0.0).sum(axis=1) * wp[c]
                        ins[:, c] = np.where(hit, jidx, BIG).min(axis=1)       # first hit position
            else:                                            # many classes: scatter instead of a pass per class
                ok = Cid >= 0
                r = np.broadcast_to(np.arange(b)[:, None], Cid.shape)[ok]
                cc = Cid[ok]
                np.add.at(combined, (r, cc), S64[ok] * wp[cc])
                np.minimum.at(ins, (r, cc), np.broadcast_to(jidx, Cid.shape)[ok])
        present = ins < BIG
        if P is not None:
            P64 = P.astype(np.float64)
            ncls = C if regular else min(k, C)
            if ncls == C:
                in_top = np.ones((b, C), dtype=bool)
                rank = np.argsort(np.argsort(-P, axis=1, kind="stable"), axis=1, kind="stable")
            else:
                top = np.argsort(-P, axis=1, kind="stable")[:, :ncls]                 # torch.topk order
                in_top = np.zeros((b, C), dtype=bool)
                np.put_along_axis(in_top, top, True, axis=1)
                rank = np.full((b, C), BIG, dtype=np.int64)
                np.put_along_axis(rank, top, np.broadcast_to(np.arange(ncls), top.shape), axis=1)
            combined += np.where(in_top, P64 * wh[None, :], 0.0)
            newly = in_top & ~present
            ins = np.where(newly, kp + rank, ins)
            present |= in_top
        score = np.where(present, combined, -np.inf)
        order = np.lexsort((ins, -score), axis=1)                         # desc score, then insertion order
        total = np.where(present, combined, 0.0).sum(axis=1)
        denom = np.where(total > 0, total, 1.0)
        vals = np.take_along_axis(combined / denom[:, None], order, axis=1).tolist()
        names = np.array([self.id_to_label[c] for c in range(C)], dtype=object)
        labs = names[order].tolist()
        npres = np.minimum(present.sum(axis=1), k).tolist()
        return [list(zip(labs[q][:n], vals[q][:n])) for q, n in enumerate(npres)]

    # ------------------------------------------------------------------------------ misc API
    def get_memory_stats(self) -> Dict[str, Any]:
        return self.memory.get_stats()

    def get_example_statistics(self) -> Dict[str, Any]:
        stats = {
            "total_examples": sum(len(exs) for exs in self.memory.examples.values()),
            "examples_per_class": {label: len(exs) for label, exs in self.memory.examples.items()},
            "num_classes": len(self.label_to_id),
            "train_steps": self.train_steps,
            "memory_usage": {
                "prototypes": sum(p.nelement() * p.element_size() for p in self.memory.prototypes.values()),
                "examples": sum(sum(ex.embedding.nelement() * ex.embedding.element_size() for ex in exs)
                                for exs in self.memory.examples.values()),
            },
        }
        if self.adaptive_head is not None:
            stats["model_params"] = sum(p.nelement() for p in self.adaptive_head.parameters())
        return stats

    def clear_memory(self, labels: Optional[List[str]] = None):
        if labels is None:
            self.memory.clear()
        else:
            for label in labels:
                self.memory.examples.pop(label, None)
                self.memory.prototypes.pop(label, None)
                self.memory.drop_label(label)
            self.memory._rebuild_index()

    def merge_classifiers(self, other: "AdaptiveClassifier") -> "AdaptiveClassifier":
        """classifier.py:1402-1426: fold another classifier's labels and stored examples into this one, then re-initialise
        and retrain the head on everything stored.  Same order of effects as the reference: new labels get the next free
        ids in `other.label_to_id` order, `other`'s examples go through the memory's add_example semantics class by class
        (prune / prototype / rebuild counters; here one batched device pass per class, memory.add_examples_batch), the head
        is rebuilt only if this classifier already had one.  As in the reference, training_history is not merged and the
        index is whatever the lazy rebuild counter left (the next search brings it up to date)."""
        if self.embedding_dim != other.embedding_dim:
            raise ValueError("Classifiers have different embedding dimensions")
        next_idx = max(self.id_to_label.keys()) + 1              # (raises on an empty classifier, like the reference)
        for label in other.label_to_id:
            if label not in self.label_to_id:
                self.label_to_id[label] = next_idx
                self.id_to_label[next_idx] = label
                next_idx += 1
        for label, examples in list(other.memory.examples.items()):
            if examples:
                self.memory.add_examples_batch(list(examples), [label] * len(examples))
        if self.adaptive_head is not None:
            self._initialize_adaptive_head()
            self._train_adaptive_head()
        return self

    def _update_adaptive_head(self):
        """classifier.py:1524-1531: make the head cover every known label (create it, or grow its output layer)."""
        num_classes = len(self.label_to_id)
        if self.adaptive_head is None:
            self._initialize_adaptive_head()
        elif num_classes > self.adaptive_head.model[-1].out_features:
            self.adaptive_head.update_num_classes(num_classes)
            self.adaptive_head = self.adaptive_head.to(self.device)

    def to(self, device: str) -> "AdaptiveClassifier":
        if str(device) != str(self.device):
            raise nv.NativeError("this build binds encoder and memory to one GPU; construct a new classifier")
        return self

    # ------------------------------------------------------------------------------ persistence (N1)
    def select_representative_examples(self, examples: List[Example], k: int = 5) -> List[Example]:
        """classifier.py:1533-1573."""
        return select_representative_examples(examples, k)

    def save(self, save_dir: str, all_examples: bool = False, include_onnx: bool = False, quantize_onnx: bool = False):
        """On-disk layout of classifier.py:524-628 (config.json, examples.json, model.safetensors).
        Like the reference, examples.json holds the k-means representatives of every class
        (config.num_representative_examples, :560-566); `all_examples=True` (additive) writes every stored example.
        include_onnx / quantize_onnx (classifier.py:1185-1197) are accepted and ignored: ONNX export is outside this build."""
        from safetensors.torch import save_file
        d = Path(save_dir)
        d.mkdir(parents=True, exist_ok=True)
        cfg = {"model_name": self.model.config._name_or_path, "embedding_dim": self.embedding_dim,
               "label_to_id": self.label_to_id, "id_to_label": {str(k): v for k, v in self.id_to_label.items()},
               "train_steps": self.train_steps, "training_history": self.training_history,
               "config": self.config.to_dict(), "library_name": "adaptive-classifier"}
        (d / "config.json").write_text(json.dumps(cfg, indent=2, sort_keys=True))
        keep = (lambda exs: exs) if all_examples else \
            (lambda exs: select_representative_examples(exs, k=self.config.num_representative_examples))
        ex = {label: [e.to_dict() for e in keep(exs)] for label, exs in self.memory.examples.items()}
        (d / "examples.json").write_text(json.dumps(ex, indent=2, sort_keys=True))
        tensors = {f"prototype_{label}": p.contiguous() for label, p in self.memory.prototypes.items()}
        if self.adaptive_head is not None:
            for key, value in self.adaptive_head.state_dict().items():
                tensors[f"adaptive_head_{key}"] = value.detach().cpu().contiguous()
        save_file(tensors, str(d / "model.safetensors"))
        card = d / "README.md"
        if not card.exists():        # classifier.py:594-599 writes a generated model card here; a minimal one keeps the file set equal
            card.write_text("---\nlibrary_name: adaptive-classifier\n---\n\n# Adaptive classifier\n\nEncoder: `%s`; %d classes: %s; "
                            "%d stored examples.\nLoad with `AdaptiveClassifier.load(<this directory>)`.\n"
                            % (cfg["model_name"], len(self.label_to_id), ", ".join(sorted(self.label_to_id)),
                               sum(len(v) for v in self.memory.examples.values())))

    _save_pretrained = save

    @classmethod
    def load(cls, save_dir: str, device: Optional[str] = None, use_onnx: Optional[Union[bool, str]] = "auto",
             prefer_quantized: bool = True, trust_remote_code: bool = False, *, encoder=None, tokenizer=None):
        """classifier.py:1200-1213 / _from_pretrained :631-915: build a classifier for the saved model_name
        and restore its state.  `encoder` / `tokenizer` may be passed to skip the HF download (offline use)."""
        cfg = json.loads((Path(save_dir) / "config.json").read_text())
        clf = cls(cfg["model_name"], device=device, config=cfg.get("config"), use_onnx=use_onnx,
                  trust_remote_code=trust_remote_code, encoder=encoder, tokenizer=tokenizer)
        return clf.load_state(save_dir)

    _from_pretrained = load

    def save_pretrained(self, save_directory: str, **kwargs):
        """ModelHubMixin.save_pretrained for a local directory (classifier.py:524): the files save() writes."""
        self.save(str(save_directory))
        return str(save_directory)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, *, device: Optional[str] = None, revision: Optional[str] = None,
                        cache_dir: Optional[str] = None, local_files_only: bool = False, token=None, trust_remote_code: bool = False,
                        encoder=None, tokenizer=None, **kwargs):
        """The reference's usual entry point (README: `AdaptiveClassifier.from_pretrained("adaptive-classifier/llm-router")`,
        ModelHubMixin -> _from_pretrained, classifier.py:631-915): a local directory is loaded as it is; anything else is taken
        for a Hub repository id and fetched with huggingface_hub.snapshot_download (config.json, examples.json,
        model.safetensors -- the ONNX files of a repository are not needed), then loaded the same way.  Uploading
        (push_to_hub), model cards and ONNX export stay outside this build."""
        import os
        path = str(pretrained_model_name_or_path)
        if not os.path.isdir(path):
            from huggingface_hub import snapshot_download
            path = snapshot_download(repo_id=path, revision=revision, cache_dir=cache_dir, local_files_only=local_files_only,
                                     token=token, allow_patterns=["config.json", "examples.json", "*.safetensors"])
        return cls.load(path, device=device, trust_remote_code=trust_remote_code, encoder=encoder, tokenizer=tokenizer)

    def _outside_this_build(self, what, where):
        raise NotImplementedError(f"{what} ({where}) is outside the MI355X hot-path build (predict / add_examples); "
                                  "use the reference package for it")

    def predict_strategic(self, *args, **kwargs):
        self._outside_this_build("strategic prediction", "classifier.py:1594+")

    def predict_robust(self, *args, **kwargs):
        self._outside_this_build("robust prediction", "classifier.py:1594+")

    def evaluate_strategic_robustness(self, *args, **kwargs):
        self._outside_this_build("strategic evaluation", "classifier.py:1594+")

    def export_onnx(self, *args, **kwargs):
        self._outside_this_build("ONNX export", "classifier.py:1031-1104")

    def push_to_hub(self, *args, **kwargs):
        raise NotImplementedError("push_to_hub (classifier.py:917+, ModelHubMixin) is outside the MI355X hot-path build: "
                                  "save() the classifier and upload the directory with huggingface_hub")

    def load_state(self, save_dir: str):
        """Restore labels, examples, prototypes and head from a directory written by save(), by the
        reference's _save_pretrained (config.json + examples.json + model.safetensors, classifier.py:524-628)
        or in the reference's older layout (examples inline in config.json + tensors.safetensors, as in its
        scripts/adaptive_router fixture)."""
        from safetensors.torch import load_file
        d = Path(save_dir)
        cfg = json.loads((d / "config.json").read_text())
        self.label_to_id = dict(cfg["label_to_id"])
        self.id_to_label = {int(k): v for k, v in cfg["id_to_label"].items()}
        self.train_steps = cfg.get("train_steps", 0)
        tfile = d / "model.safetensors"
        if not tfile.exists():
            tfile = d / "tensors.safetensors"
        tensors = load_file(str(tfile))
        if (d / "examples.json").exists():
            saved = json.loads((d / "examples.json").read_text())
        else:
            saved = cfg.get("examples", {})
        self.memory.clear()
        saved_counts = {}
        for label, exs in saved.items():
            self.memory.examples[label] = [Example.from_dict(e) for e in exs]
            saved_counts[label] = len(exs)
        for label in self.label_to_id:
            key = f"prototype_{label}"
            if key in tensors:
                self.memory.prototypes[label] = tensors[key]
        self.memory._restore_from_save()
        self.training_history = cfg.get("training_history") or {l: 20 * n for l, n in saved_counts.items()}  # :909-913
        head_sd = {k[len("adaptive_head_"):]: v for k, v in tensors.items() if k.startswith("adaptive_head_")}
        if head_sd:
            self._initialize_adaptive_head()
            self.adaptive_head.load_state_dict(head_sd)
            self.adaptive_head = self.adaptive_head.to(self.device)
        return self
