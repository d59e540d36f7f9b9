"""Shared test helpers: an offline tokenizer stub and a small random BERT."""
import zlib

import numpy as np
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


def near_tie_store(n, D, seed, clusters=50, noise=2e-4):
    """Rows that cluster tightly around `clusters` unit centres (thousands of near-equal distances per query: the
    norm/dot fp32 forms reorder them), a third of them copies of other rows with three coordinates moved by one ulp
    (pairs whose exact distances differ by ~1e-9 relative: below what ANY fp32 summation resolves).  Returns (P, centres)."""
    from oracle import synth
    centres = synth.synth_unit_rows(clusters, D, seed)
    rng = np.random.default_rng(seed)
    P = centres[rng.integers(0, clusters, n)] + (rng.standard_normal((n, D)) * noise).astype(np.float32)
    P = np.ascontiguousarray(P, dtype=np.float32)
    src = rng.integers(0, n, n // 3)
    dst = rng.permutation(n)[: n // 3]
    P[dst] = P[src]
    for t in range(3):
        c = rng.integers(0, D, n // 3)
        P[dst, c] = np.nextafter(P[dst, c], np.float32(np.inf if t % 2 else -np.inf))
    return P, centres
