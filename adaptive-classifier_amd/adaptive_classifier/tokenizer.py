"""HipWordPieceTokenizer: the tokenizer call of the hot path on MI355X.

Replaces `self.tokenizer(texts, max_length=..., truncation=True, padding=True, return_tensors="pt")`
(/root/reference/src/adaptive_classifier/classifier.py:1259-1265) for BERT-family WordPiece vocabularies
(transformers BertTokenizer / DistilBertTokenizer ...: tokenizers BertNormalizer + BertPreTokenizer + WordPiece).
The ids are produced on the device by `ac_wordpiece_encode` and handed to the encoder without a host round trip:
host work per batch is one concatenation of the texts' bytes, one small H2D and one 4-byte D2H (the padded length,
which the encoder launch needs).

Exactness: identical ids / mask to the wrapped transformers tokenizer.  The kernel covers ASCII texts of up to 4096
bytes; texts with non-ASCII bytes (accent stripping / CJK spacing / Unicode punctuation live in the host library),
longer texts, and texts containing a literal special-token string ("[SEP]", "[MASK]" ...) are tokenised by the wrapped
tokenizer itself and spliced into the batch -- the reference's own tokenizer, not a re-implementation.
"""
import ctypes

import numpy as np
import torch

from . import _native as nv

MAX_DEVICE_BYTES = 4096


def _wordpiece_spec(tok):
    """(vocab dict, lower_case, specials) if `tok` is a BERT-style WordPiece tokenizer this module reproduces, else None."""
    be = getattr(tok, "backend_tokenizer", None) or getattr(tok, "_tokenizer", None)
    if be is None:
        return None
    try:
        import json
        cfg = json.loads(be.to_str())
    except Exception:
        return None
    model, norm, pre = cfg.get("model") or {}, cfg.get("normalizer") or {}, cfg.get("pre_tokenizer") or {}
    if model.get("type") != "WordPiece" or norm.get("type") != "BertNormalizer" or pre.get("type") != "BertPreTokenizer":
        return None
    if model.get("continuing_subword_prefix", "##") != "##" or int(model.get("max_input_chars_per_word", 100)) != 100:
        return None
    if not norm.get("clean_text", True):
        return None
    post = cfg.get("post_processor") or {}
    if post.get("type") not in ("TemplateProcessing", "BertProcessing"):
        return None
    vocab = model.get("vocab") or {}
    need = [tok.unk_token, tok.cls_token, tok.sep_token, tok.pad_token]
    if any(t is None or t not in vocab for t in need) or model.get("unk_token") != tok.unk_token:
        return None
    specials = sorted(set(str(t) for t in getattr(tok, "all_special_tokens", need)) |
                      set(a["content"] for a in cfg.get("added_tokens") or []))
    return vocab, bool(norm.get("lowercase", True)), specials


class HipWordPieceTokenizer:
    """Callable with the signature the classifier uses; returns device tensors."""

    def __init__(self, hf_tokenizer, device=None):
        nv.require_gpu()
        spec = _wordpiece_spec(hf_tokenizer)
        if spec is None:
            raise nv.NativeError("HipWordPieceTokenizer: not a BERT WordPiece tokenizer (BertNormalizer + BertPreTokenizer + WordPiece)")
        self.hf = hf_tokenizer
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        vocab, self.lower, self.specials = spec
        self._build_table(vocab)
        self.device_texts = self.host_texts = 0          # routing statistics

    # ---- vocabulary -> device hash table ---------------------------------------------------------------
    def _build_table(self, vocab):
        L = nv.lib()
        pieces = []
        for tokstr, tid in vocab.items():
            cont = tokstr.startswith("##") and len(tokstr) > 2
            raw = (tokstr[2:] if cont else tokstr).encode("utf-8")
            if len(raw) == 0 or len(raw) > 65535:
                continue
            pieces.append((raw, int(tid), cont))
        slots = 1
        while slots < 2 * len(pieces) + 16:
            slots <<= 1
        keys = np.zeros(slots, np.uint64)
        offs = np.zeros(slots, np.uint32)
        ids = np.zeros(slots, np.int32)
        lens = np.zeros(slots, np.uint16)
        blob = bytearray()
        mask = slots - 1
        max_piece = 1
        for raw, tid, cont in pieces:
            buf = (ctypes.c_uint8 * len(raw)).from_buffer_copy(raw)
            h = int(L.ac_wordpiece_hash(buf, len(raw), 1 if cont else 0))
            s = (h ^ (h >> 32)) & mask
            while keys[s] != 0:
                s = (s + 1) & mask
            keys[s], offs[s], ids[s], lens[s] = h, len(blob), tid | ((1 << 30) if cont else 0), len(raw)
            blob += raw
            max_piece = max(max_piece, len(raw))
        blob += b"\0" * 16
        dev = self.device
        self._t = [torch.from_numpy(a).to(dev) for a in (keys.view(np.int64), offs.view(np.int32), ids, lens.view(np.int16),
                                                          np.frombuffer(bytes(blob), np.uint8).copy())]
        hf = self.hf
        self.vocab = nv.ac_wordpiece_vocab(*[t.data_ptr() for t in self._t], slots, max_piece,
                                           vocab[hf.unk_token], vocab[hf.cls_token], vocab[hf.sep_token], vocab[hf.pad_token],
                                           1 if self.lower else 0)
        self.pad_id = vocab[hf.pad_token]

    # ---- the call --------------------------------------------------------------------------------------
    def _device_ok(self, text, raw):
        return len(raw) <= MAX_DEVICE_BYTES and raw.isascii() and not any(sp in text for sp in self.specials)

    def __call__(self, texts, max_length=512, truncation=True, padding=True, return_tensors="pt", **kw):
        if isinstance(texts, str):
            texts = [texts]
        if not truncation or not padding or return_tensors != "pt" or kw:
            raise nv.NativeError("HipWordPieceTokenizer supports the classifier's call only: truncation=True, padding=True, "
                                 "return_tensors='pt'")
        b = len(texts)
        M = int(max_length)
        dev = self.device
        # common case first: one pass over the joined batch decides that every text is plain ASCII, short enough and free
        # of special-token strings (a match across a text boundary only costs the slow path)
        joined = "".join(texts)
        if joined.isascii() and not any(sp in joined for sp in self.specials) and max(map(len, texts), default=0) <= MAX_DEVICE_BYTES:
            on_dev = [True] * b
            sizes = [len(t) for t in texts]
            blob = joined.encode("ascii")
        else:
            raws = [t.encode("utf-8") for t in texts]
            on_dev = [self._device_ok(t, r) for t, r in zip(texts, raws)]
            sizes = [len(r) if ok else 0 for r, ok in zip(raws, on_dev)]
            blob = b"".join(r for r, ok in zip(raws, on_dev) if ok)
        ids = torch.empty((b, M), dtype=torch.int64, device=dev)
        mask = torch.empty((b, M), dtype=torch.int64, device=dev)
        lens = torch.empty(b, dtype=torch.int32, device=dev)
        # offsets and text travel in ONE host-to-device copy: [int32 offs[b + 1] | pad to 16 | text bytes | 64 zero bytes]
        head = (4 * (b + 1) + 15) // 16 * 16
        buf = np.zeros(head + len(blob) + 64, np.uint8)
        buf[:4 * (b + 1)].view(np.int32)[1:] = np.cumsum(sizes, dtype=np.int64).astype(np.int32)
        buf[head:head + len(blob)] = np.frombuffer(blob, np.uint8)
        d_buf = torch.from_numpy(buf).to(dev)
        d_offs, d_text = d_buf[:4 * (b + 1)], d_buf[head:]
        with torch.cuda.device(dev):
            nv.check(nv.lib().ac_wordpiece_encode(nv.ptr(d_text), nv.ptr(d_offs), b, ctypes.byref(self.vocab), M, nv.ptr(ids),
                                                  nv.ptr(mask), nv.ptr(lens), nv.stream_ptr(dev)), "ac_wordpiece_encode")
        host_rows = [i for i, ok in enumerate(on_dev) if not ok]
        self.device_texts += b - len(host_rows)
        self.host_texts += len(host_rows)
        if host_rows:            # the wrapped tokenizer itself for what the kernel does not cover; spliced into the batch
            enc = self.hf([texts[i] for i in host_rows], max_length=M, truncation=True, padding="max_length", return_tensors="pt")
            rows = torch.tensor(host_rows, dtype=torch.int64, device=dev)
            ids[rows] = enc["input_ids"].to(dev)
            mask[rows] = enc["attention_mask"].to(dev)
            lens[rows] = enc["attention_mask"].sum(1).to(device=dev, dtype=torch.int32)
        S = int(lens.max().item()) if b else 0                    # the one host sync: the encoder launch needs S
        ids_s, mask_s = ids[:, :S].contiguous(), mask[:, :S].contiguous()
        return {"input_ids": ids_s, "token_type_ids": torch.zeros_like(ids_s), "attention_mask": mask_s}

    # pass-through of what callers may read from the wrapped tokenizer
    def __getattr__(self, name):
        return getattr(self.hf, name)


def maybe_device_tokenizer(hf_tokenizer, device):
    """HipWordPieceTokenizer around `hf_tokenizer` when it is a BERT WordPiece tokenizer, else the tokenizer unchanged."""
    if hf_tokenizer is None or isinstance(hf_tokenizer, HipWordPieceTokenizer):
        return hf_tokenizer
    try:
        if _wordpiece_spec(hf_tokenizer) is None:
            return hf_tokenizer
        return HipWordPieceTokenizer(hf_tokenizer, device)
    except Exception:
        return hf_tokenizer
