"""On-device WordPiece (ac_wordpiece_encode behind HipWordPieceTokenizer) against transformers' BertTokenizer with the
same vocabulary: identical input_ids / attention_mask / token_type_ids for the classifier's call
(max_length, truncation=True, padding=True; classifier.py:1259-1265).  No pretrained vocabulary is available offline,
so the vocabularies are synthetic (specials + characters + generated word pieces)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

WORDS = ("the of and to in is that for it as was with be by on not he this are or his from at which but have an had they "
         "you were their one all we can her has there been if more when will would who so no out up into do any what them "
         "great product works well love much best purchase ever made terrible waste money awful buy broke after day fine "
         "nothing special average okay neither good bad password reset login help support please account payment refund "
         "shipping order tracking number classifier adaptive prototype memory neural network embedding transformer attention "
         "playing played plays player unbelievable internationalization tokenization tokenizer wordpiece").split()


def _vocab(drop=()):
    toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    chars = [chr(c) for c in range(33, 127)]
    toks += [c for c in chars if c not in drop and not ("A" <= c <= "Z")]
    toks += ["##" + c for c in chars if ("##" + c) not in drop and (c.isalnum() and not c.isupper())]
    seen = set(toks)
    rng = np.random.default_rng(0)
    for w in WORDS:
        for piece in (w, w[: max(2, len(w) // 2)], "##" + w[len(w) // 2:], "##" + w[-3:], "##ing", "##ed", "##s", "##ly", "##tion"):
            if piece not in seen and len(piece.replace("##", "")) > 0:
                seen.add(piece); toks.append(piece)
    for _ in range(300):                                   # random multi-character pieces, incl. digits
        n = int(rng.integers(2, 7))
        s = "".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz0123456789"), n))
        for piece in (s, "##" + s):
            if piece not in seen:
                seen.add(piece); toks.append(piece)
    return {t: i for i, t in enumerate(toks)}


def _texts():
    rng = np.random.default_rng(1)
    out = ["", "   ", "hello", "Hello, World!  This is GREAT...", "a\tb\nc\rd", "ctrl\x01inside\x7fword \x00 nul",
           "x" * 100 + " " + "y" * 101 + " z", "don't stop-believing (really)!?", "price: $12.50 + 3% = #wow @home",
           "unbelievable internationalization tokenization", "UPPER lower MiXeD 123abc abc123", "q" * 300,
           "trailing space ", " leading", "multiple    spaces\t\ttabs", "[brackets] {braces} <angles> |pipes|",
           "café naïve résumé", "中文 mixed with english", "emoji \U0001f600 here",
           "has a [SEP] literal and a [MASK] too", "long " * 1200]
    for _ in range(60):
        n = int(rng.integers(1, 40))
        ws = []
        for _ in range(n):
            w = str(rng.choice(WORDS))
            r = rng.random()
            if r < 0.15: w = w.upper()
            elif r < 0.3: w = w.capitalize()
            elif r < 0.4: w = w + str(int(rng.integers(0, 1000)))
            elif r < 0.5: w = w + str(rng.choice(list(",.;:!?'\"-()")))
            elif r < 0.55: w = "".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz"), int(rng.integers(1, 15))))
            ws.append(w)
        out.append(str(rng.choice([" ", "  ", "\t", "\n"])).join(ws))
    return out


@pytest.mark.parametrize("lower,drop,max_length", [(True, (), 512), (True, ("z", "##q", "7", "'"), 64), (False, (), 24), (True, (), 8)])
def test_device_wordpiece_equals_transformers(lower, drop, max_length, cuda_dev):
    from transformers import BertTokenizer
    from adaptive_classifier.tokenizer import HipWordPieceTokenizer
    vocab = _vocab(drop)
    if not lower:                                           # cased vocabulary: add capitals so they are not all [UNK]
        for c in "ABCDEFGHIJKLMNOPQRSTUVWXYZ":
            vocab.setdefault(c, len(vocab)); vocab.setdefault("##" + c, len(vocab))
    hf = BertTokenizer(vocab=vocab, do_lower_case=lower)
    dev_tok = HipWordPieceTokenizer(hf, device=cuda_dev)
    texts = _texts()
    for lo in range(0, len(texts), 17):                     # several batches: different padded lengths
        batch = texts[lo:lo + 17]
        want = hf(batch, max_length=max_length, truncation=True, padding=True, return_tensors="pt")
        got = dev_tok(batch, max_length=max_length, truncation=True, padding=True, return_tensors="pt")
        for key in ("input_ids", "attention_mask", "token_type_ids"):
            g, w = got[key].cpu(), want[key]
            assert g.shape == w.shape, (key, g.shape, w.shape)
            bad = (g != w).any(dim=1).nonzero().flatten().tolist()
            assert not bad, (key, [(batch[i], g[i].tolist(), w[i].tolist()) for i in bad[:2]])
        assert got["input_ids"].is_cuda
    assert dev_tok.device_texts > 0 and dev_tok.host_texts >= 4     # non-ASCII / special-token / over-long texts took the host route


def test_classifier_uses_the_device_tokenizer(cuda_dev):
    """A classifier built with a transformers BertTokenizer tokenises on the device and predicts what the same classifier
    with the host tokenizer predicts."""
    from transformers import BertTokenizer
    from adaptive_classifier import AdaptiveClassifier
    from adaptive_classifier.encoder import HipBertEncoder
    from adaptive_classifier.tokenizer import HipWordPieceTokenizer
    from helpers import small_bert
    vocab = _vocab()
    hf = BertTokenizer(vocab=vocab, do_lower_case=True)
    enc = HipBertEncoder(small_bert(vocab=len(vocab) + 8), device=cuda_dev)
    a = AdaptiveClassifier("synthetic", device="cuda:0", encoder=enc, tokenizer=hf)
    b = AdaptiveClassifier("synthetic", device="cuda:0", config={"device_tokenizer": False}, encoder=enc, tokenizer=hf)
    assert isinstance(a.tokenizer, HipWordPieceTokenizer) and b.tokenizer is hf
    texts = ["great product works well", "love it so much", "terrible waste of money", "awful do not buy",
             "it is fine nothing special", "average product okay"]
    labels = ["pos", "pos", "neg", "neg", "neu", "neu"]
    for c in (a, b):
        c.add_examples(texts, labels)
    for t in ["really great product!", "AWFUL... do NOT buy", "okay I guess"]:
        pa, pb = a.predict(t, k=3), b.predict(t, k=3)
        assert [l for l, _ in pa] == [l for l, _ in pb]
        assert np.allclose([s for _, s in pa], [s for _, s in pb], atol=1e-6)
    oa, ob = a.predict_batch(texts, k=2), b.predict_batch(texts, k=2)
    assert [[l for l, _ in p] for p in oa] == [[l for l, _ in p] for p in ob]
