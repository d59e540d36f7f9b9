"""Shared test helpers: an offline tokenizer stub and a small random BERT."""
import zlib

import torch


class HashTokenizer:
    """Whitespace + crc32 'tokenizer' with the HF call signature the classifier uses
    (classifier.py:1259-1265): pads to the longest text of the batch, truncates to max_length."""

    def __init__(self, vocab=2000):
        self.vocab = vocab

    def __call__(self, texts, max_length=512, truncation=True, padding=True, return_tensors="pt"):
        toks = []
        for t in texts:
            ids = [101] + [1000 + zlib.crc32(w.encode()) % (self.vocab - 1000) for w in t.lower().split()] + [102]
            toks.append(ids[:max_length])
        S = max(len(t) for t in toks)
        ids = torch.zeros((len(toks), S), dtype=torch.int64)
        mask = torch.zeros((len(toks), S), dtype=torch.int64)
        for i, t in enumerate(toks):
            ids[i, : len(t)] = torch.tensor(t)
            mask[i, : len(t)] = 1
        return {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask}


def small_bert(hidden=128, layers=2, heads=2, inter=512, vocab=2000, seed=0):
    from oracle import bert_oracle
    return bert_oracle.make_bert(hidden, layers, heads, inter, vocab=vocab, seed=seed)
