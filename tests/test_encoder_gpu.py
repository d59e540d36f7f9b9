"""Parity of the HIP BERT encoder (ac_bert_encode_cls) against transformers BertModel fp32 on CPU.
Target: max-abs error 1e-4 on the unit-norm CLS vector."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hidden,layers,heads,inter,b,S,ragged", [
    (128, 2, 2, 512, 5, 16, True),        # bert-tiny architecture
    (768, 12, 12, 3072, 8, 32, True),     # bert-base, BASELINE configs[0] shape (batch 8)
    (768, 2, 12, 3072, 3, 130, True),     # S > one key tile, odd sizes
    (1024, 3, 16, 4096, 4, 24, False),    # bert-large / e5-large-v2 width
])
def test_encoder_cls_matches_transformers(hidden, layers, heads, inter, b, S, ragged, cuda_dev):
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    vocab = 2000
    model = bert_oracle.make_bert(hidden, layers, heads, inter, vocab=vocab, seed=0)
    ids, types, mask = bert_oracle.synthetic_batch(b, S, vocab=vocab, seed=1234, ragged=ragged)
    types[:, S // 2:] = 1
    want = bert_oracle.encode_cls(model, ids, types, mask)
    enc = HipBertEncoder(model, device=cuda_dev)
    got = enc.encode_cls(ids, types, mask).cpu()
    err = (got - want).abs().max().item()
    cos = torch.nn.functional.cosine_similarity(got, want, dim=1).min().item()
    assert err < 1e-4, (err, cos)
    assert abs(got.norm(dim=1) - 1).max().item() < 1e-5


def test_encoder_no_mask_and_large_batch(cuda_dev):
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    model = bert_oracle.make_bert(768, 2, 12, 3072, vocab=3000, seed=1)
    ids, types, mask = bert_oracle.synthetic_batch(64, 32, vocab=3000, ragged=False)
    want = bert_oracle.encode_cls(model, ids, types, mask)
    enc = HipBertEncoder(model, device=cuda_dev)
    got = enc.encode_cls(ids).cpu()            # type ids / mask omitted = zeros / ones
    assert (got - want).abs().max().item() < 1e-4
