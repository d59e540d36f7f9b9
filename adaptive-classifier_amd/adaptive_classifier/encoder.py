"""HipBertEncoder: the encoder call of the hot path on MI355X.

Replaces `self.model(**inputs).last_hidden_state[:, 0, :]` + `F.normalize`
(/root/reference/src/adaptive_classifier/classifier.py:1271-1275) for BERT-architecture, DistilBERT and RoBERTa-family
checkpoints (bert-base-uncased, bert-large, e5-large-v2, bert-tiny, distilbert-base-*, roberta-*, xlm-roberta-* /
multilingual-e5-*) with one native call,
`ac_bert_encode_cls`, that writes unit-norm CLS vectors straight into a device buffer usable as the
kNN query block (no device->host copy, cf. classifier.py:1282).

Weights are taken from a transformers `BertModel` (pretrained or randomly initialised); Q/K/V
projections are fused into one [3H, H] matrix.  `HipModernBertEncoder` covers ModernBERT
(answerdotai/ModernBERT-base: RoPE, alternating global / sliding-window attention, pre-norm, GeGLU) through
`ac_modernbert_encode_cls`; `make_encoder` picks by `config.model_type`.  Other architectures are not
covered by the HIP encoders and raise.
"""
import ctypes
import os

import torch

from . import _native as nv


MAX_TOKENS = 1 << 17          # tokens per native encoder call (~8 GB of activations + planes for bert-base)


class _Cfg:
    """The slice of a HF config the classifier reads (classifier.py:88,549)."""

    def __init__(self, hidden_size, name_or_path):
        self.hidden_size = hidden_size
        self._name_or_path = name_or_path


ROBERTA_TYPES = ("roberta", "xlm-roberta", "camembert")

SMALL_TOKENS = 32     # up to here ac_bert_encode_cls runs the whole forward as one persistent launch (bert_small.hip): no packing

ARITH_MODES = {"f32": nv.AC_GEMM_F32, "bf16x3": nv.AC_GEMM_BF16X3, "f16x2": nv.AC_GEMM_F16X2}


def arith_mode(value):
    """"f32" | "bf16x3" | "f16x2" | AC_GEMM_* | None (the process-wide default) -> AC_GEMM_* or None; anything else: ValueError."""
    if value is None:
        return None
    if isinstance(value, str):
        if value not in ARITH_MODES:
            raise ValueError(f"gemm_arith must be one of {sorted(ARITH_MODES)} (or None for the process default), got {value!r}")
        return ARITH_MODES[value]
    if int(value) not in ARITH_MODES.values():
        raise ValueError(f"gemm_arith: unknown mode {value!r}")
    return int(value)


class HipBertEncoder:
    def __init__(self, hf_bert, device=None, unpad=True):
        """unpad: leave the padding tokens out of the forward (ac_bert_pack + ac_bert_encode_cls_packed) whenever the
        attention mask is right-padded -- same CLS vectors from sum(len) token rows instead of b * S."""
        nv.require_gpu()
        self.unpad = bool(unpad)
        self.last_tokens = 0                  # token rows the last encode_cls call actually ran (roofline accounting)
        self.last_one_launch = False          # the last native call ran as the one persistent launch (bert_small.hip)
        self.ln_gave_up = 0                   # encode_cls calls repeated because a fused-LayerNorm exchange gave up
        # PER-OBJECT options (None = the process-wide default of include/acamd.h): they travel to the native calls inside
        # ac_bert_config (gemm_arith_opt / ln_fusion_opt) and hold for THIS encoder's calls only; encode_cls(arith=...) overrides
        # the arithmetic per call (how two classifiers sharing one encoder keep their own).
        self.arith = None                     # AC_GEMM_* or None
        self.ln_fusion = None                 # True / False / None; set to False by this encoder once an exchange gave up
        cfg = hf_bert.config
        mtype = getattr(cfg, "model_type", "bert")
        if mtype not in ("bert", "distilbert", "electra") + ROBERTA_TYPES:
            raise nv.NativeError(f"HipBertEncoder covers BERT / DistilBERT / RoBERTa-family / ELECTRA encoders, got {mtype!r}")
        if mtype == "electra" and getattr(cfg, "embedding_size", cfg.hidden_size) != cfg.hidden_size:
            raise nv.NativeError("HipBertEncoder: ELECTRA with embedding_size != hidden_size (embeddings_project) is not covered")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        sd = {k: v.detach() for k, v in hf_bert.state_dict().items()}
        pos_offset = 0
        if mtype in ("bert", "electra") or mtype in ROBERTA_TYPES:          # (ELECTRA's discriminator body is the BERT block, same names)
            # RoBERTa / XLM-RoBERTa / CamemBERT (modeling_roberta.py): the BERT block under the same parameter names; positions
            # count from padding_idx + 1 (create_position_ids_from_input_ids: cumsum over the non-pad tokens + padding_idx), which
            # for right-padded inputs -- the only kind the classifier's tokenizer call produces -- is a constant row offset into
            # the position table; padded positions never reach the CLS row.  One token type (ids are 0 or absent).
            if mtype in ROBERTA_TYPES:
                pos_offset = int(cfg.pad_token_id) + 1
            if getattr(cfg, "position_embedding_type", "absolute") not in (None, "absolute"):
                raise nv.NativeError("HipBertEncoder: only absolute position embeddings are supported")
            act = cfg.hidden_act
            H, L, A, I = cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.intermediate_size
            type_vocab, eps = cfg.type_vocab_size, float(cfg.layer_norm_eps)
            names = {"word": "embeddings.word_embeddings.weight", "pos": "embeddings.position_embeddings.weight",
                     "type": "embeddings.token_type_embeddings.weight", "eln": "embeddings.LayerNorm",
                     "layer": "encoder.layer.{}.", "q": "attention.self.query", "k": "attention.self.key",
                     "v": "attention.self.value", "ao": "attention.output.dense", "ln1": "attention.output.LayerNorm",
                     "ff1": "intermediate.dense", "ff2": "output.dense", "ln2": "output.LayerNorm"}
        else:
            # DistilBERT (modeling_distilbert.py): same block, no token-type embeddings -> a single zero row
            act = cfg.activation
            H, L, A, I = cfg.dim, cfg.n_layers, cfg.n_heads, cfg.hidden_dim
            type_vocab, eps = 1, 1e-12
            names = {"word": "embeddings.word_embeddings.weight", "pos": "embeddings.position_embeddings.weight",
                     "type": None, "eln": "embeddings.LayerNorm",
                     "layer": "transformer.layer.{}.", "q": "attention.q_lin", "k": "attention.k_lin",
                     "v": "attention.v_lin", "ao": "attention.out_lin", "ln1": "sa_layer_norm",
                     "ff1": "ffn.lin1", "ff2": "ffn.lin2", "ln2": "output_layer_norm"}
        if act not in ("gelu",):
            raise nv.NativeError(f"HipBertEncoder: activation {act!r} unsupported (erf-GELU only)")
        self.config = _Cfg(H, getattr(cfg, "_name_or_path", ""))
        self.training = False
        self.ccfg = nv.ac_bert_config(H, L, A, I, cfg.vocab_size, cfg.max_position_embeddings - pos_offset, type_vocab, eps)

        def dev(t):
            return t.to(device=self.device, dtype=torch.float32).contiguous()

        self._keep = []                      # owns the device tensors

        def own(t):
            t = dev(t)
            self._keep.append(t)
            return t

        per = {k: [] for k in ("qkv_w", "qkv_b", "ao_w", "ao_b", "ln1_g", "ln1_b", "ff1_w", "ff1_b",
                               "ff2_w", "ff2_b", "ln2_g", "ln2_b")}
        for l in range(L):
            p = names["layer"].format(l)
            per["qkv_w"].append(own(torch.cat([sd[p + names[x] + ".weight"] for x in ("q", "k", "v")], 0)))
            per["qkv_b"].append(own(torch.cat([sd[p + names[x] + ".bias"] for x in ("q", "k", "v")], 0)))
            for dst, src in (("ao", "ao"), ("ff1", "ff1"), ("ff2", "ff2")):
                per[dst + "_w"].append(own(sd[p + names[src] + ".weight"]))
                per[dst + "_b"].append(own(sd[p + names[src] + ".bias"]))
            for dst in ("ln1", "ln2"):
                per[dst + "_g"].append(own(sd[p + names[dst] + ".weight"]))
                per[dst + "_b"].append(own(sd[p + names[dst] + ".bias"]))
        self._arrays = {}
        w = nv.ac_bert_weights()
        w.word_emb = own(sd[names["word"]]).data_ptr()
        w.pos_emb = own(sd[names["pos"]][pos_offset:]).data_ptr()
        w.type_emb = own(sd[names["type"]] if names["type"] else torch.zeros(1, H)).data_ptr()
        w.emb_ln_g = own(sd[names["eln"] + ".weight"]).data_ptr()
        w.emb_ln_b = own(sd[names["eln"] + ".bias"]).data_ptr()
        self._has_types = names["type"] is not None
        for k, tensors in per.items():
            arr = (ctypes.c_void_p * L)(*[t.data_ptr() for t in tensors])
            self._arrays[k] = arr             # host array of device pointers; must outlive the calls
            setattr(w, k, ctypes.cast(arr, ctypes.c_void_p).value)
        # bf16x3 operand planes of the four weight matrices per layer (ac_split_bf16x3, once): with them the
        # token-row GEMMs stage pre-split operands instead of splitting inside every tile (AC_GEMM_BF16X3)
        self._planes = []
        for k in ("qkv_w", "ao_w", "ff1_w", "ff2_w"):
            planes = [_split_planes(t, self.device) for t in per[k]]
            self._planes.extend(planes)
            arr = (ctypes.c_void_p * L)(*[t.data_ptr() for t in planes])
            self._arrays[k + "3"] = arr
            setattr(w, k + "3", ctypes.cast(arr, ctypes.c_void_p).value)
        self.weights = w
        self._ws = None
        self.num_params = sum(t.numel() for t in self._keep)
        self._gemm_w = {k: per[k] for k in ("qkv_w", "ao_w", "ff1_w", "ff2_w")}
        self._planes_f16 = None               # fp16x2 weight planes (enable_f16x2)
        self.f16x2_overflows = 0              # encode_cls calls repeated in bf16x3 because an activation left the fp16 range
        if nv.lib().ac_gemm_get_arith() == nv.AC_GEMM_F16X2:     # AC_GEMM_ARITH=f16x2 in the environment (process-wide default)
            try:
                self.enable_f16x2()
            except nv.NativeError as e:                           # (an explicit enable_f16x2() call raises; a process-wide
                import logging                                    #  preference does not stop a model from loading)
                logging.getLogger(__name__).warning("%s -- this encoder stays in bf16x3", e)

    # -- opt-in fp16x2 arithmetic of the token-row GEMMs (include/acamd.h: AC_GEMM_F16X2) ------
    F16X2_MAX_WEIGHT = 63.9                   # |w| 2^10 must stay below fp16's 65504

    def enable_f16x2(self):
        """Build the fp16x2 planes of the four GEMM weights per layer (ac_split_f16x2, once) and hand them to the native
        encoder.  They are USED by calls whose arithmetic is AC_GEMM_F16X2 -- the call's `arith`, this encoder's `set_arith`, or
        the process-wide default (ac_gemm_set_arith(2) / AC_GEMM_ARITH=f16x2) -- with >= 192 token rows; see include/acamd.h for
        what the mode trades."""
        if self._planes_f16 is None:
            big = max(float(t.abs().max()) for ts in self._gemm_w.values() for t in ts)
            if not big < self.F16X2_MAX_WEIGHT:
                raise nv.NativeError(f"fp16x2 arithmetic: a GEMM weight of magnitude {big:g} leaves the fp16 range at scale 2^10 "
                                     f"(|w| < {self.F16X2_MAX_WEIGHT}); keep the default bf16x3 arithmetic for this model")
            self._planes_f16 = {k: [_split_planes_f16(t, self.device) for t in ts] for k, ts in self._gemm_w.items()}
        L = self.ccfg.layers
        for k, planes in self._planes_f16.items():
            arr = (ctypes.c_void_p * L)(*[t.data_ptr() for t in planes])
            self._arrays[k + "h"] = arr
            setattr(self.weights, k + "h", ctypes.cast(arr, ctypes.c_void_p).value)
        return self

    def disable_f16x2(self):
        """Back to bf16x3 for this encoder whatever the process-wide mode (the planes stay allocated)."""
        for k in ("qkv_wh", "ao_wh", "ff1_wh", "ff2_wh"):
            setattr(self.weights, k, None)
        return self

    def effective_arith(self, arith=None) -> int:
        """The arithmetic a call with per-call option `arith` runs in: the call's, else this encoder's, else the process's."""
        a = arith_mode(arith)
        if a is None:
            a = self.arith
        return nv.lib().ac_gemm_get_arith() if a is None else a

    def f16x2_active(self, arith=None) -> bool:
        return bool(self.weights.qkv_wh) and self.effective_arith(arith) == nv.AC_GEMM_F16X2

    def set_arith(self, arith):
        """This encoder's arithmetic for calls that do not name one ("f32" | "bf16x3" | "f16x2" | None = process default).
        "f16x2" builds the fp16 weight planes (raises NativeError for weights outside the fp16 range)."""
        a = arith_mode(arith)
        if a == nv.AC_GEMM_F16X2:
            self.enable_f16x2()
        self.arith = a
        return self

    def disable_ln_fusion(self):
        """LayerNorms as separate launches for THIS encoder from now on (what it does itself after an exchange gave up)."""
        self.ln_fusion = False
        return self

    def _call_cfg(self, arith=None):
        """ac_bert_config of one native call: the architecture + this call's options (0 = process default, else value + 1)."""
        a = arith_mode(arith)
        if a is None:
            a = self.arith
        c = self.ccfg
        return nv.ac_bert_config(c.hidden, c.layers, c.heads, c.intermediate, c.vocab, c.max_pos, c.type_vocab, c.ln_eps,
                                 0 if a is None else a + 1,
                                 0 if self.ln_fusion is None else (2 if self.ln_fusion else 1), 0)

    def _call_cfg_shared(self, arith=None):
        """_call_cfg for encode_cls's own use: one struct per (arithmetic, fusion) state, never handed out, never modified."""
        a = arith_mode(arith)
        if a is None:
            a = self.arith
        key = (a, self.ln_fusion)
        cached = self.__dict__.get("_cfg_cache")
        if cached is None or cached[0] != key:
            cached = self._cfg_cache = (key, self._call_cfg(arith))
        return cached[1]

    # -- the nn.Module-ish surface classifier.py touches (:1253-1255,1278-1279,1215) ----------
    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        self.training = False                 # inference-only encoder
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise nv.NativeError("HipBertEncoder is bound to its GPU; build a new one for another device")
        return self

    def _as_ids(self, t):
        """int64, contiguous, on this encoder's device (already so on the predict paths: three attribute reads, no dispatch)."""
        if t.dtype is torch.int64 and t.device == self.device and t.is_contiguous():
            return t
        return t.to(device=self.device, dtype=torch.int64).contiguous()

    def workspace_bytes(self, b, S):
        cache = self.__dict__.setdefault("_ws_bytes_cache", {})       # (a function of the architecture and (b, S) only: the
        got = cache.get((b, S))                                        #  predict path asks with the same pair at every step)
        if got is None:
            need = ctypes.c_size_t(0)
            nv.check(nv.lib().ac_bert_workspace(ctypes.byref(self.ccfg), b, S, ctypes.byref(need)), "ac_bert_workspace")
            if len(cache) > 256:
                cache.clear()
            got = cache[(b, S)] = need.value
        return got

    def encode_cls(self, input_ids, token_type_ids=None, attention_mask=None, out=None, verify=True, force_layered=False,
                   verify_small=None, arith=None):
        """int64 [b, S] ids (+ optional type ids / mask) -> unit-norm CLS embeddings [b, H] on device.
        arith: this call's GEMM arithmetic ("f32" | "bf16x3" | "f16x2"; None = this encoder's `arith`, else the process default) --
        a per-call option inside ac_bert_config, no process-wide state is touched.

        Two kernels of this forward wait for other workgroups with a BOUNDED wait and poison their rows with NaN when they
        give up (a device shared with another compute process, or under a CU mask): the one persistent launch that runs
        <= 32 token rows (bert_small.hip, launched without the cooperative residency check) and the fused-LayerNorm GEMM
        epilogues of the layer-by-layer path (gemm_pipe.hip).  verify=True (default): their verdicts are read after the
        call's last launch (one small D2H, i.e. ONE stream sync per call whatever the number of row chunks) and on a
        give-up the call is repeated without the kernel that gave up (layer by layer / LayerNorm fusion switched off for
        the process) -- every caller gets finite embeddings for finite weights.  Callers that look at their final result
        anyway (the classifier's predict paths) pass verify=False, keep the GPU queue full, and on NaN ask
        `ln_fusion_aborted()` / `last_one_launch` and call again.  verify_small: older name of `verify`."""
        if verify_small is not None:
            verify = verify_small
        ids = self._as_ids(input_ids)
        b, S = ids.shape
        if not self._has_types:
            token_type_ids = None           # DistilBERT has no segment embeddings (tokenizer may still emit ids)
        tt = None if token_type_ids is None else self._as_ids(token_type_ids)
        mk = None if attention_mask is None else self._as_ids(attention_mask)
        H = self.ccfg.hidden
        if out is None:
            out = torch.empty((b, H), dtype=torch.float32, device=self.device)
        # activations cost ~64 KB per token; very large batches go through in row chunks of <= MAX_TOKENS tokens
        cb = b if b * S <= MAX_TOKENS else max(1, MAX_TOKENS // S)
        need = self.workspace_bytes(min(b, cb), S)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            # a FRESH workspace starts with clean verdict words (acamd.h: the first 256 bytes): a <= 32-row call that ends up
            # layer by layer (bert-large: no one-launch kernel; a one-launch abort retried layered) reads them below without
            # having cleared them, and torch.empty memory is garbage -- a spurious "gave up" would switch the fusion off
            self._ws[:256].zero_()
        cfg = self._call_cfg_shared(arith)
        with torch.cuda.device(self.device):
            # the fused-LayerNorm verdict of this call starts clean (sticky over the chunks below; include/acamd.h); a call
            # that runs as the one persistent launch has no such epilogue (and is the latency path: no extra launch)
            layered = self._run_chunks(ids, tt, mk, b, S, cb, out, verify, force_layered, cfg, clear=b * S > SMALL_TOKENS or force_layered)
            if verify and layered and self.ln_fusion_aborted():
                import logging
                logging.getLogger(__name__).warning(
                    "encoder: a fused LayerNorm epilogue gave up waiting for the tiles of a row panel (device shared or "
                    "CU-masked?); LayerNorm fusion is now off for this encoder and the batch is encoded again")
                self.ln_gave_up += 1
                self.disable_ln_fusion()
                cfg = self._call_cfg_shared(arith)
                self._run_chunks(ids, tt, mk, b, S, cb, out, verify, force_layered, cfg, clear=True)
            if verify and layered and self.f16x2_active(arith) and not bool(torch.isfinite(out).all()):
                # an activation beyond fp16's range at scale 2^6 turned its rows into NaN (never into a wrong number)
                import logging
                logging.getLogger(__name__).warning(
                    "encoder: non-finite embeddings under fp16x2 arithmetic (an activation beyond +-1023?); the batch is "
                    "encoded again in bf16x3%s", "" if arith_mode(arith) is not None else
                    ", which is this encoder's arithmetic from now on")
                self.f16x2_overflows += 1
                # per object: the fp16 planes stay for other users of this encoder (classifiers that pass arith="f16x2");
                # only the owner of the choice changes -- the call's `arith` is the caller's (the classifier demotes itself),
                # an encoder-level or process-level f16x2 becomes an encoder-level bf16x3
                if arith_mode(arith) is None:
                    self.arith = nv.AC_GEMM_BF16X3
                cfg = self._call_cfg_shared(nv.AC_GEMM_BF16X3)
                self._run_chunks(ids, tt, mk, b, S, cb, out, verify, force_layered, cfg)
        return out

    # one call for packing + forward, no stream synchronisation (include/acamd.h ac_bert_encode_cls_unpad); AC_BERT_UNPAD_ONE_CALL=0
    # keeps the separate ac_bert_pack -> read-back -> ac_bert_encode_cls_packed form (A/B runs, the equivalence test)
    _UNPAD_MAX_SEQS = 8192

    def _run_chunks(self, ids, tt, mk, b, S, cb, out, verify, force_layered, cfg=None, clear=False):
        """The native calls of one encode_cls: row chunks of <= cb sequences.  Returns True when at least one chunk ran layer
        by layer (the path whose GEMM epilogues may carry the fused LayerNorm).  clear: the fused-LayerNorm verdict words of
        the workspace start clean with this call (sticky over its chunks)."""
        self.last_tokens = 0
        layered = False
        cfg = self.ccfg if cfg is None else cfg
        one_call = os.environ.get("AC_BERT_UNPAD_ONE_CALL", "1") != "0"
        if one_call and os.environ.get("AC_LIBACAMD_PATH"):      # (an older library under A/B, tools/: the entry may not exist)
            try:
                nv.lib().ac_bert_encode_cls_unpad
            except AttributeError:
                one_call = False
        for r0 in range(0, b, cb):
            r1 = min(b, r0 + cb)
            nb = r1 - r0
            mk_all_ones = False
            unpad = self.unpad and mk is not None and S > 1 and nb * S > SMALL_TOKENS
            if unpad and one_call and nb <= self._UNPAD_MAX_SEQS:
                total, path = ctypes.c_int(0), ctypes.c_int(0)
                whole = nb == b                 # (one chunk -- every predict batch: no tensor views, ~2 us of interpreter each)
                nv.check(nv.lib().ac_bert_encode_cls_unpad(
                    ctypes.byref(cfg), ctypes.byref(self.weights), nv.ptr(ids if whole else ids[r0:r1]),
                    nv.ptr(None if tt is None else (tt if whole else tt[r0:r1])), nv.ptr(mk if whole else mk[r0:r1]), nb, S,
                    nv.ptr(out if whole else out[r0:r1]), out.stride(0), nv.ptr(self._ws), self._ws.numel(),
                    1 if clear else 0, ctypes.byref(total), ctypes.byref(path), nv.stream_ptr(self.device)),
                    "ac_bert_encode_cls_unpad")
                clear = False
                self.last_one_launch = False
                layered = True
                self.last_tokens += total.value
                continue
            if clear:
                nv.check(nv.lib().ac_bert_ln_fusion_clear(nv.ptr(self._ws), self._ws.numel(), nv.stream_ptr(self.device)),
                         "ac_bert_ln_fusion_clear")
                clear = False
            if unpad:
                # padding-free path: pack on the device, read back {rows, prefix flag, longest} (one 16-byte D2H)
                cu = torch.empty(nb + 1, dtype=torch.int32, device=self.device)
                src = torch.empty(nb * S, dtype=torch.int32, device=self.device)
                info = torch.empty(4, dtype=torch.int32, device=self.device)
                nv.check(nv.lib().ac_bert_pack(nv.ptr(mk[r0:r1]), nb, S, nv.ptr(cu), nv.ptr(src), nv.ptr(info),
                                               nv.stream_ptr(self.device)), "ac_bert_pack")
                total, not_prefix, longest, _ = info.tolist()
                if not not_prefix and total < nb * S:
                    self.last_one_launch = False
                    layered = True
                    nv.check(nv.lib().ac_bert_encode_cls_packed(
                        ctypes.byref(cfg), ctypes.byref(self.weights), nv.ptr(ids[r0:r1]),
                        nv.ptr(None if tt is None else tt[r0:r1]), nb, S, nv.ptr(cu), nv.ptr(src), total, longest,
                        nv.ptr(out[r0:r1]), out.stride(0), nv.ptr(self._ws), self._ws.numel(),
                        nv.stream_ptr(self.device)), "ac_bert_encode_cls_packed")
                    self.last_tokens += total
                    continue
                if not not_prefix and total == nb * S:
                    mk_all_ones = True          # nothing to leave out: the unpacked forward needs no mask (and may fuse its attention)
            used = ctypes.c_int(0)

            def call(opts):
                nv.check(nv.lib().ac_bert_encode_cls_opts(
                    ctypes.byref(cfg), ctypes.byref(self.weights), nv.ptr(ids[r0:r1]),
                    nv.ptr(None if tt is None else tt[r0:r1]), nv.ptr(None if (mk is None or mk_all_ones) else mk[r0:r1]), nb, S,
                    nv.ptr(out[r0:r1]), out.stride(0), nv.ptr(self._ws), self._ws.numel(), opts, ctypes.byref(used),
                    nv.stream_ptr(self.device)), "ac_bert_encode_cls_opts")
            call(nv.AC_BERT_LAYERED if force_layered else 0)
            self.last_one_launch = bool(used.value)
            if used.value and verify:
                aborted = ctypes.c_int(0)
                nv.check(nv.lib().ac_bert_one_launch_status(ctypes.byref(self.ccfg), nb, S, nv.ptr(self._ws),
                                                            self._ws.numel(), ctypes.byref(aborted),
                                                            nv.stream_ptr(self.device)), "ac_bert_one_launch_status")
                if aborted.value:
                    import logging
                    logging.getLogger(__name__).warning(
                        "one-launch encoder: a grid barrier gave up (device shared?); repeating layer by layer")
                    call(nv.AC_BERT_LAYERED)
                    self.last_one_launch = False
            layered = layered or not self.last_one_launch
            self.last_tokens += nb * S
        return layered

    def ln_fusion_aborted(self) -> bool:
        """Verdict of the fused-LayerNorm GEMM epilogues over ALL the chunks of the last encode_cls() call (sticky word at the
        head of the workspace, cleared when a call starts; an 8-byte D2H: stream sync).  False after a call that ran as the
        one persistent launch only (that path has no such epilogue)."""
        if self._ws is None:
            return False
        aborted = ctypes.c_int(0)
        nv.check(nv.lib().ac_bert_ln_fusion_status(ctypes.byref(self.ccfg), 1, 1, nv.ptr(self._ws), self._ws.numel(),
                                                   ctypes.byref(aborted), nv.stream_ptr(self.device)), "ac_bert_ln_fusion_status")
        return bool(aborted.value)

    def flops(self, b, S, executed=True, tokens=None, sum_len_sq=None):
        """FLOPs of one forward (dense projections + attention), for roofline reports.
        executed=True counts what the kernels run (the last layer's output projection / FFN and its
        attention only touch the b CLS rows); executed=False is the full BertModel.forward count.
        tokens / sum_len_sq: the padding-free path's row count and sum of squared sequence lengths (attention)."""
        c = self.ccfg
        H, I, L = c.hidden, c.intermediate, c.layers
        T = b * S
        if tokens is not None and executed:
            per_tok = 2.0 * (4.0 * H * H + 2.0 * H * I)
            ssq = float(sum_len_sq if sum_len_sq is not None else tokens * tokens / max(b, 1))
            attn_layer = 4.0 * c.heads * ssq * (H // c.heads)
            qkv_last = (2.0 * tokens * 2.0 * H * H + 2.0 * b * H * H) if tokens >= 4 * b else 2.0 * tokens * 3.0 * H * H
            last = qkv_last + 2.0 * b * (H * H + 2.0 * H * I) + 4.0 * c.heads * tokens * (H // c.heads)
            return per_tok * tokens * (L - 1) + attn_layer * (L - 1) + last
        per_tok = 2.0 * (4.0 * H * H + 2.0 * H * I)
        attn_layer = 4.0 * b * c.heads * S * S * (H // c.heads)
        if not executed:
            return per_tok * T * L + attn_layer * L
        # (last layer: K and V of every token, Q of the b CLS rows only when the batch has >= 4 token rows per sequence)
        qkv_last = (2.0 * T * 2.0 * H * H + 2.0 * b * H * H) if T >= 4 * b else 2.0 * T * 3.0 * H * H
        last = qkv_last + 2.0 * b * (H * H + 2.0 * H * I) + attn_layer / S
        return per_tok * T * (L - 1) + attn_layer * (L - 1) + last


def _split_planes(t, device):
    """ac_split_bf16x3 of one [rows, K] fp32 weight (operand planes for AC_GEMM_BF16X3)."""
    rows, K = t.shape
    pl = torch.empty(3 * rows * K, dtype=torch.int16, device=device)
    with torch.cuda.device(device):           # the launch must target the weights' GPU, not the process's current one
        nv.check(nv.lib().ac_split_bf16x3(t.data_ptr(), K, rows, K, pl.data_ptr(), nv.stream_ptr(device)), "ac_split_bf16x3")
    return pl


def _split_planes_f16(t, device):
    """ac_split_f16x2 of one [rows, K] fp32 weight at scale 2^10 (operand planes for AC_GEMM_F16X2)."""
    rows, K = t.shape
    pl = torch.empty(2 * rows * K, dtype=torch.int16, device=device)
    with torch.cuda.device(device):
        nv.check(nv.lib().ac_split_f16x2(t.data_ptr(), K, rows, K, 10, pl.data_ptr(), nv.stream_ptr(device)), "ac_split_f16x2")
    return pl


class HipModernBertEncoder:
    """ModernBERT (transformers modeling_modernbert.py) behind the same surface as HipBertEncoder."""

    def __init__(self, hf_model, device=None, unpad=True):
        nv.require_gpu()
        self.unpad = bool(unpad)            # leave the padding tokens of ragged batches out of the forward
        self.last_tokens = 0
        cfg = hf_model.config
        if getattr(cfg, "model_type", "") != "modernbert":
            raise nv.NativeError(f"HipModernBertEncoder needs a ModernBERT model, got {getattr(cfg, 'model_type', None)!r}")
        if cfg.hidden_activation != "gelu":
            raise nv.NativeError(f"HipModernBertEncoder: activation {cfg.hidden_activation!r} unsupported (erf-GELU only)")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        H, L, A, I = cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.intermediate_size
        types = list(cfg.layer_types)
        # the period of the global layers: `global_attn_every_n_layers` where the config carries it (checkpoints' config.json do),
        # else read off `layer_types` (a default-constructed transformers >= 5 ModernBertConfig only has the list)
        every = getattr(cfg, "global_attn_every_n_layers", None)
        if every is None:
            glob = [l for l, t in enumerate(types) if t == "full_attention"]
            every = (glob[1] - glob[0]) if len(glob) > 1 else max(L, 1)
        every = int(every)
        if any((t == "full_attention") != (l % every == 0) for l, t in enumerate(types)):
            raise nv.NativeError("HipModernBertEncoder: layer_types must be global every global_attn_every_n_layers")
        self.config = _Cfg(H, getattr(cfg, "_name_or_path", ""))
        self.training = False
        self.ccfg = nv.ac_modernbert_config(H, L, A, I, cfg.vocab_size, cfg.max_position_embeddings, every,
                                            int(cfg.sliding_window), float(cfg.norm_eps), 0)
        self.arith = None               # this encoder's GEMM arithmetic (None = the process default); per call: encode_cls(arith=)
        sd = {k: v.detach() for k, v in hf_model.state_dict().items()}
        self._keep, self._arrays = [], {}

        def own(t):
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
            self._keep.append(t)
            return t

        w = nv.ac_modernbert_weights()
        w.tok_emb = own(sd["embeddings.tok_embeddings.weight"]).data_ptr()
        w.emb_norm_g = own(sd["embeddings.norm.weight"]).data_ptr()
        w.emb_norm_b = own(sd["embeddings.norm.bias"]).data_ptr() if "embeddings.norm.bias" in sd else None
        w.final_norm_g = own(sd["final_norm.weight"]).data_ptr()
        w.final_norm_b = own(sd["final_norm.bias"]).data_ptr() if "final_norm.bias" in sd else None
        w.zero_bias = own(torch.zeros(max(3 * H, 2 * I))).data_ptr()
        # RoPE tables exactly as ModernBertRotaryEmbedding builds them: fp32 inv_freq, fp32 outer product, cos / sin
        dh = H // A
        pos = torch.arange(cfg.max_position_embeddings, dtype=torch.float32)
        for kind, key in (("global", "full_attention"), ("local", "sliding_attention")):
            theta = float(cfg.rope_parameters[key]["rope_theta"])
            inv_freq = 1.0 / (theta ** (torch.arange(0, dh, 2, dtype=torch.int64).to(dtype=torch.float) / dh))
            freqs = (inv_freq[:, None].float() @ pos[None, :]).transpose(0, 1)          # [max_pos, dh/2]
            setattr(w, f"rope_cos_{kind}", own(freqs.cos()).data_ptr())
            setattr(w, f"rope_sin_{kind}", own(freqs.sin()).data_ptr())

        # Wi rows (and bias) interleaved in blocks of 32 inputs + their 32 gates: the GeGLU then fuses into the
        # GEMM epilogue (EPI_GEGLU32), because a wave's two 32-column tiles hold input_j / gate_j in the same lane
        inter = I % 32 == 0
        if inter:
            t = torch.arange(I // 32)[:, None] * 32 + torch.arange(32)[None, :]            # [I/32, 32] input rows
            perm = torch.cat([t, t + I], dim=1).reshape(-1)                               # in-block, then gate-block
        w.wi_interleaved32 = 1 if inter else 0

        def per_layer(field, name, optional=False):
            ts = []
            for l in range(L):
                k = f"layers.{l}.{name}"
                v = sd.get(k)
                if v is not None and inter and name in ("mlp.Wi.weight", "mlp.Wi.bias"):
                    v = v[perm]
                ts.append(own(v) if v is not None else None)
            if all(t is None for t in ts):
                if not optional:
                    raise nv.NativeError(f"HipModernBertEncoder: {name} missing from the checkpoint")
                setattr(w, field, None)
                return ts
            arr = (ctypes.c_void_p * L)(*[None if t is None else t.data_ptr() for t in ts])
            self._arrays[field] = arr
            setattr(w, field, ctypes.cast(arr, ctypes.c_void_p).value)
            return ts

        per_layer("attn_norm_g", "attn_norm.weight")            # layer 0: Identity -> NULL entry
        per_layer("attn_norm_b", "attn_norm.bias", optional=True)
        mats = {"wqkv": per_layer("wqkv", "attn.Wqkv.weight"), "wo": per_layer("wo", "attn.Wo.weight"),
                "wi": per_layer("wi", "mlp.Wi.weight"), "wo2": per_layer("wo2", "mlp.Wo.weight")}
        per_layer("wqkv_b", "attn.Wqkv.bias", optional=True)
        per_layer("wo_b", "attn.Wo.bias", optional=True)
        per_layer("mlp_norm_g", "mlp_norm.weight")
        per_layer("mlp_norm_b", "mlp_norm.bias", optional=True)
        per_layer("wi_b", "mlp.Wi.bias", optional=True)
        per_layer("wo2_b", "mlp.Wo.bias", optional=True)
        self._planes = []
        for field, key in (("wqkv3", "wqkv"), ("wo3", "wo"), ("wi3", "wi"), ("wo23", "wo2")):
            planes = [_split_planes(t, self.device) for t in mats[key]]
            self._planes.extend(planes)
            arr = (ctypes.c_void_p * L)(*[t.data_ptr() for t in planes])
            self._arrays[field] = arr
            setattr(w, field, ctypes.cast(arr, ctypes.c_void_p).value)
        self.weights = w
        self._ws = None
        self.num_params = sum(t.numel() for t in self._keep)

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        self.training = False
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise nv.NativeError("HipModernBertEncoder is bound to its GPU; build a new one for another device")
        return self

    def workspace_bytes(self, b, S):
        need = ctypes.c_size_t(0)
        nv.check(nv.lib().ac_modernbert_workspace(ctypes.byref(self.ccfg), b, S, ctypes.byref(need)),
                 "ac_modernbert_workspace")
        return need.value

    def _call_cfg(self, arith=None):
        """ac_modernbert_config of one native call: the architecture + this call's arithmetic (0 = process default, else value + 1)."""
        a = arith_mode(arith)
        if a is None:
            a = self.arith
        c = self.ccfg
        return nv.ac_modernbert_config(c.hidden, c.layers, c.heads, c.intermediate, c.vocab, c.max_pos, c.global_every,
                                       c.local_window, c.norm_eps, 0 if a is None else a + 1)

    def encode_cls(self, input_ids, token_type_ids=None, attention_mask=None, out=None, verify=True, force_layered=False,
                   verify_small=None, arith=None):
        """int64 [b, S] ids (+ optional mask; token types do not exist in ModernBERT) -> unit-norm CLS [b, H].
        arith: this call's GEMM arithmetic ("f32" | "bf16x3" | "f16x2"; None = this encoder's `arith`, else the process default),
        a per-call option inside ac_modernbert_config as for HipBertEncoder; "f16x2" runs as bf16x3 here (no fp16 weight planes
        are built for this family).
        (verify / force_layered: interface parity with HipBertEncoder; this encoder has neither a one-launch path nor
        LayerNorm-fused GEMM epilogues, i.e. no kernel that can give up.)"""
        ccfg = self._call_cfg(arith)
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        b, S = ids.shape
        mk = None if attention_mask is None else attention_mask.to(device=self.device, dtype=torch.int64).contiguous()
        H = self.ccfg.hidden
        if out is None:
            out = torch.empty((b, H), dtype=torch.float32, device=self.device)
        cb = b if b * S <= MAX_TOKENS else max(1, MAX_TOKENS // S)
        need = self.workspace_bytes(min(b, cb), S)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self.last_tokens = 0
        with torch.cuda.device(self.device):
            for r0 in range(0, b, cb):
                r1 = min(b, r0 + cb)
                nb = r1 - r0
                if self.unpad and mk is not None and S > 1:
                    # padding-free path (same packing kernels as the BERT encoder): RoPE positions stay per sequence
                    cu = torch.empty(nb + 1, dtype=torch.int32, device=self.device)
                    src = torch.empty(nb * S, dtype=torch.int32, device=self.device)
                    info = torch.empty(4, dtype=torch.int32, device=self.device)
                    nv.check(nv.lib().ac_bert_pack(nv.ptr(mk[r0:r1]), nb, S, nv.ptr(cu), nv.ptr(src), nv.ptr(info),
                                                   nv.stream_ptr(self.device)), "ac_bert_pack")
                    total, not_prefix, longest, _ = info.tolist()
                    if not not_prefix and total < nb * S:
                        nv.check(nv.lib().ac_modernbert_encode_cls_packed(
                            ctypes.byref(ccfg), ctypes.byref(self.weights), nv.ptr(ids[r0:r1]), nb, S, nv.ptr(cu),
                            nv.ptr(src), total, longest, nv.ptr(out[r0:r1]), out.stride(0), nv.ptr(self._ws),
                            self._ws.numel(), nv.stream_ptr(self.device)), "ac_modernbert_encode_cls_packed")
                        self.last_tokens += total
                        continue
                self.last_tokens += nb * S
                nv.check(nv.lib().ac_modernbert_encode_cls(
                    ctypes.byref(ccfg), ctypes.byref(self.weights), nv.ptr(ids[r0:r1]),
                    nv.ptr(None if mk is None else mk[r0:r1]), r1 - r0, S, nv.ptr(out[r0:r1]), out.stride(0),
                    nv.ptr(self._ws), self._ws.numel(), nv.stream_ptr(self.device)), "ac_modernbert_encode_cls")
        return out

    def flops(self, b, S, executed=True):
        c = self.ccfg
        H, I, L = c.hidden, c.intermediate, c.layers
        n_glob = sum(1 for l in range(L) if l % c.global_every == 0)
        keys_local = min(S, 2 * c.local_window + 1)
        attn = 4.0 * b * c.heads * S * (H // c.heads) * (n_glob * S + (L - n_glob) * keys_local)
        return 2.0 * b * S * L * (4.0 * H * H + 3.0 * H * I) + attn


def make_encoder(hf_model, device=None):
    """The native encoder for a transformers model: BERT / DistilBERT -> HipBertEncoder, ModernBERT ->
    HipModernBertEncoder; anything else raises (there is no eager fallback)."""
    mtype = getattr(hf_model.config, "model_type", "bert")
    if mtype == "modernbert":
        return HipModernBertEncoder(hf_model, device=device)
    return HipBertEncoder(hf_model, device=device)
