"""Parity of the HIP BERT encoder (ac_bert_encode_cls) against transformers BertModel fp32 on CPU.
Target: max-abs error 1e-4 on the unit-norm CLS vector."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# Attention regimes of the random-init oracle models (oracle/bert_oracle.sharpen_attention):
#   flat   -- transformers' own init (std 0.02): logits ~0.3, softmax within a few percent of uniform;
#   peaked -- query / key projections scaled so the mean max attention probability is >= 0.5 (S <= 64) / >= 0.3 (S = 512),
#             plus four LayerNorm outlier channels (gain x12), as trained checkpoints have.
# The same 1e-4 bar holds in both; the peaked regime is the one that drives the online-softmax max / rescale path and would
# expose a key-mask mistake.
REGIMES = ["flat", "peaked"]


def regime_kw(regime, hidden):
    if regime == "flat":
        return {}
    return {"qk_scale": 8.0 if hidden < 256 else 5.0, "ln_outlier": 12.0}


def check_peaked(regime, model, ids, mask, types=None, S=None):
    """print the attention statistics; in the peaked regime they must be far from uniform"""
    from oracle import bert_oracle
    mean_max, min_layer = bert_oracle.attention_peak_stats(model, ids, mask, types)
    S = S or ids.shape[1]
    print(f"[attention stats] regime={regime} S={S}: mean max probability {mean_max:.3f} (lowest layer {min_layer:.3f}), "
          f"uniform would be ~{1.0 / max(1.0, float(mask.sum(1).float().mean())):.3f}")
    if regime == "peaked":
        assert mean_max >= (0.3 if S >= 256 else 0.5), mean_max


@pytest.mark.parametrize("hidden,layers,heads,inter,b,S,ragged", [
    (128, 2, 2, 512, 5, 16, True),        # bert-tiny architecture
    (768, 12, 12, 3072, 8, 32, True),     # bert-base, BASELINE configs[0] shape (batch 8)
    (768, 2, 12, 3072, 3, 130, True),     # S > one key tile, odd sizes
    (1024, 3, 16, 4096, 4, 24, False),    # bert-large / e5-large-v2 width
    (768, 2, 12, 3072, 2, 512, True),     # the reference's max_length (classifier.py:1262), 16 key tiles
    (768, 1, 12, 3072, 3, 33, True),      # one key past a tile boundary
    (1024, 2, 16, 4096, 9, 30, True),     # bert-large width with T = 270 rows: pre-split operand planes, ragged tiles
    (1024, 24, 16, 4096, 16, 32, True),   # FULL-DEPTH bert-large = e5-large-v2 architecture (BASELINE configs[4]): 24 layers
    (384, 6, 12, 1536, 24, 32, True),     # all-MiniLM-L6-v2 architecture: 12 heads of 32 dims (round 4), packed path
    (384, 2, 12, 1536, 2, 130, True),     # head dim 32 over several key tiles
    (384, 3, 12, 1536, 1, 12, False),     # head dim 32, a single short query (the one-launch kernel covers head dim 64 only: layered)
    (384, 2, 12, 1536, 40, 16, False),    # head dim 32, full rows: LayerNorm-fused GEMM epilogues with three column tiles
])
@pytest.mark.parametrize("regime", REGIMES)
def test_encoder_cls_matches_transformers(hidden, layers, heads, inter, b, S, ragged, regime, cuda_dev):
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    vocab = 2000
    model = bert_oracle.make_bert(hidden, layers, heads, inter, vocab=vocab, seed=0, **regime_kw(regime, hidden))
    ids, types, mask = bert_oracle.synthetic_batch(b, S, vocab=vocab, seed=1234, ragged=ragged)
    types[:, S // 2:] = 1
    if layers <= 12:
        check_peaked(regime, model, ids, mask, types)
    want = bert_oracle.encode_cls(model, ids, types, mask)
    enc = HipBertEncoder(model, device=cuda_dev)
    got = enc.encode_cls(ids, types, mask).cpu()
    err = (got - want).abs().max().item()
    cos = torch.nn.functional.cosine_similarity(got, want, dim=1).min().item()
    assert err < 1e-4, (err, cos)
    assert abs(got.norm(dim=1) - 1).max().item() < 1e-5


@pytest.mark.parametrize("hidden,layers,heads,inter,b,S,ragged", [
    (768, 12, 12, 3072, 1, 11, False),    # a single short query: the predict(text) case, bert-base
    (768, 12, 12, 3072, 1, 32, False),    # 32 token rows = both MFMA row tiles
    (768, 3, 12, 3072, 2, 16, True),      # two sequences in one launch: attention stays inside each, ragged key masks
    (768, 2, 12, 3072, 3, 9, True),       # 27 rows, odd sizes
    (384, 6, 6, 1536, 1, 20, False),      # MiniLM-L6 width (H = 384: three k-blocks per wave)
    (128, 2, 2, 512, 4, 8, True),         # bert-tiny
])
@pytest.mark.parametrize("regime", REGIMES)
def test_small_batch_one_launch_path_matches_transformers(hidden, layers, heads, inter, b, S, ragged, regime, cuda_dev):
    """b * S <= 32 token rows: ac_bert_encode_cls runs every layer inside ONE persistent launch (bert_small.hip, strict
    fp32 MFMA).  Same 1e-4 bar against transformers fp32, and against the layer-by-layer kernels on the same texts
    (the batch replicated until it exceeds 32 rows takes that path)."""
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    vocab = 2000
    model = bert_oracle.make_bert(hidden, layers, heads, inter, vocab=vocab, seed=0, **regime_kw(regime, hidden))
    ids, types, mask = bert_oracle.synthetic_batch(b, S, vocab=vocab, seed=4321, ragged=ragged)
    types[:, S // 2:] = 1
    if S >= 16:
        check_peaked(regime, model, ids, mask, types)
    want = bert_oracle.encode_cls(model, ids, types, mask)
    enc = HipBertEncoder(model, device=cuda_dev)
    got = enc.encode_cls(ids, types, mask).cpu()
    assert (got - want).abs().max().item() < 1e-4
    assert abs(got.norm(dim=1) - 1).max().item() < 1e-5
    rep = 32 // (b * S) + 1
    big = enc.encode_cls(ids.repeat(rep, 1), types.repeat(rep, 1), mask.repeat(rep, 1)).cpu()[:b]
    assert (got - big).abs().max().item() < 2e-5


def test_small_batch_path_without_types_and_mask_and_switch(cuda_dev):
    """ids only (type ids = zeros, mask = ones) through the one-launch path; ac_set_persistent_kernels switches it off / on and
    reports the previous mask."""
    from adaptive_classifier import _native as nv
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    model = bert_oracle.make_bert(768, 2, 12, 3072, vocab=3000, seed=2)
    ids, types, mask = bert_oracle.synthetic_batch(2, 13, vocab=3000, seed=9, ragged=False)
    want = bert_oracle.encode_cls(model, ids, torch.zeros_like(ids), torch.ones_like(ids))
    enc = HipBertEncoder(model, device=cuda_dev)
    got = enc.encode_cls(ids).cpu()
    assert (got - want).abs().max().item() < 1e-4
    prev = nv.lib().ac_set_persistent_kernels(-1)
    assert prev & 2
    try:
        assert nv.lib().ac_set_persistent_kernels(prev & ~2) == prev
        layered = enc.encode_cls(ids).cpu()
        assert nv.lib().ac_set_persistent_kernels(-1) == prev & ~2
    finally:
        nv.lib().ac_set_persistent_kernels(prev)
    assert (got - layered).abs().max().item() < 2e-5 and (layered - want).abs().max().item() < 1e-4


def test_encoder_no_mask_and_large_batch(cuda_dev):
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    model = bert_oracle.make_bert(768, 2, 12, 3072, vocab=3000, seed=1)
    ids, types, mask = bert_oracle.synthetic_batch(64, 32, vocab=3000, ragged=False)
    want = bert_oracle.encode_cls(model, ids, types, mask)
    enc = HipBertEncoder(model, device=cuda_dev)
    got = enc.encode_cls(ids).cpu()            # type ids / mask omitted = zeros / ones
    assert (got - want).abs().max().item() < 1e-4


@pytest.mark.parametrize("regime", REGIMES)
def test_distilbert_encoder_matches_transformers(regime, cuda_dev):
    """N4 (part): DistilBERT = the BERT block without token-type embeddings; same kernels, mapped weights."""
    import torch.nn.functional as F
    from transformers import DistilBertConfig, DistilBertModel
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    cfg = DistilBertConfig(vocab_size=2000, dim=768, n_layers=3, n_heads=12, hidden_dim=3072)
    try:
        cfg._attn_implementation = "eager"
    except Exception:
        pass
    torch.manual_seed(0)
    model = DistilBertModel(cfg).eval()
    if regime == "peaked":
        bert_oracle.sharpen_attention(model, **regime_kw(regime, 768))
    ids, types, mask = bert_oracle.synthetic_batch(6, 40, vocab=2000, seed=7, ragged=True)
    check_peaked(regime, model, ids, mask)
    with torch.no_grad():
        want = F.normalize(model(input_ids=ids, attention_mask=mask).last_hidden_state[:, 0, :], p=2, dim=1)
    enc = HipBertEncoder(model, device=cuda_dev)
    got = enc.encode_cls(ids, types, mask).cpu()          # token_type_ids ignored for DistilBERT
    assert (got - want).abs().max().item() < 1e-4


@pytest.mark.parametrize("hidden,layers,heads,inter,local,b,S", [
    (128, 4, 2, 192, 16, 7, 40),          # T = 280: pre-split operand planes; windows of +-8 inside S = 40
    (128, 4, 2, 192, 16, 2, 20),          # T = 40: small-M GEMM path, fp32 activations
    (768, 5, 12, 1152, 128, 2, 200),      # ModernBERT-base width; +-64 windows cut inside S = 200; 7 key tiles
    (768, 2, 12, 1152, 128, 3, 33),       # one key past a tile boundary
    (128, 4, 2, 192, 128, 1, 1024),       # long sequence: 32 key tiles, +-64 windows skip most of them in local layers
    (128, 3, 2, 192, 8, 200, 12),         # b >= 192: the CLS-only last layer (a local one here) runs on planes too
    (128, 3, 2, 192, 128, 1, 8192),       # ModernBERT's full context: 256 key tiles in the global layer
])
def test_modernbert_encoder_matches_transformers(hidden, layers, heads, inter, local, b, S, cuda_dev):
    """N4: ModernBERT (RoPE, alternating global / sliding-window attention, pre-norm, GeGLU, bias-free) against
    transformers ModernBertModel fp32 eager on CPU; SURVEY 8c bar: 1e-4 max-abs on the unit-norm CLS vector."""
    from adaptive_classifier.encoder import make_encoder, HipModernBertEncoder
    from oracle import bert_oracle
    model = bert_oracle.make_modernbert(hidden, layers, heads, inter, vocab=2000, max_pos=max(1024, S), local_attention=local,
                                        seed=3, init_scale=4.0)
    ids, _, mask = bert_oracle.synthetic_batch(b, S, vocab=2000, seed=77, ragged=True)
    ids[:, 0] = 1
    want = bert_oracle.encode_cls_modernbert(model, ids, mask)
    enc = make_encoder(model, device=cuda_dev)
    assert isinstance(enc, HipModernBertEncoder)
    got = enc.encode_cls(ids, None, mask).cpu()
    err = (got - want).abs().max().item()
    assert err < 1e-4, err
    assert abs(got.norm(dim=1) - 1).max().item() < 1e-5
    # the sliding window and RoPE matter in this configuration: a global-only / position-free variant is far off
    if S > 2 * (local // 2) + 1:
        model.config.sliding_window = 10 ** 6
        assert (bert_oracle.encode_cls_modernbert(model, ids, mask) - want).abs().max().item() > 1e-3


def test_large_batches_are_encoded_in_row_chunks(cuda_dev, monkeypatch):
    """encode_cls splits batches beyond MAX_TOKENS tokens into row chunks; the result does not depend on it beyond
    fp32 rounding (different row counts take different GEMM kernels)."""
    from adaptive_classifier import encoder as enc_mod
    from oracle import bert_oracle
    model = bert_oracle.make_bert(128, 2, 2, 512, vocab=2000, seed=0)
    ids, types, mask = bert_oracle.synthetic_batch(23, 16, vocab=2000, seed=5, ragged=True)
    enc = enc_mod.HipBertEncoder(model, device=cuda_dev)
    whole = enc.encode_cls(ids, types, mask).cpu()
    monkeypatch.setattr(enc_mod, "MAX_TOKENS", 5 * 16)          # 5 rows per native call, ragged last chunk
    parts = enc.encode_cls(ids, types, mask).cpu()
    assert (whole - parts).abs().max().item() < 2e-6
    mb = bert_oracle.make_modernbert(128, 2, 2, 192, vocab=2000, max_pos=64, local_attention=8, seed=1)
    enc2 = enc_mod.make_encoder(mb, device=cuda_dev)
    monkeypatch.setattr(enc_mod, "MAX_TOKENS", 1 << 17)
    whole = enc2.encode_cls(ids, None, mask).cpu()
    monkeypatch.setattr(enc_mod, "MAX_TOKENS", 7 * 16)
    assert (whole - enc2.encode_cls(ids, None, mask).cpu()).abs().max().item() < 2e-6


@pytest.mark.parametrize("regime", REGIMES)
def test_padding_free_path_equals_padded_path(regime, cuda_dev):
    """ac_bert_pack + ac_bert_encode_cls_packed (padding tokens left out of the forward) give the CLS vectors of the padded
    forward: same dot products in the same order, masked keys contribute exact zeros.  Masks that are not right-padded
    (left padding, holes, an empty row) keep the padded path."""
    from adaptive_classifier import _native as nv
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    model = bert_oracle.make_bert(768, 3, 12, 3072, vocab=3000, seed=2, **regime_kw(regime, 768))
    packed = HipBertEncoder(model, device=cuda_dev, unpad=True)
    padded = HipBertEncoder(model, device=cuda_dev, unpad=False)
    for (b, S, seed) in [(37, 48, 3), (256, 32, 1234), (5, 130, 9), (1, 16, 4)]:
        ids, types, mask = bert_oracle.synthetic_batch(b, S, vocab=3000, seed=seed, ragged=True)
        types[:, S // 3:] = 1
        if b == 37:
            check_peaked(regime, model, ids, mask, types)
        want = bert_oracle.encode_cls(model, ids, types, mask)
        # in-order k walk (the default): the same sums in the same order in both layouts; the rotated walk (experiment switch:
        # a tile's k order depends on its XCD, i.e. on the layout): equal to fp32 rounding, which the peaked model amplifies
        for krot, tol in ((0, 1e-6), (1, 2e-5 if regime == "peaked" else 2e-6)):
            nv.lib().ac_gemm_set_krot(krot)
            try:
                a = packed.encode_cls(ids, types, mask)
                assert packed.last_tokens == int(mask.sum()) or b == 1
                c = padded.encode_cls(ids, types, mask)
                assert padded.last_tokens == b * S
            finally:
                nv.lib().ac_gemm_set_krot(0)
            assert (a - c).abs().max().item() <= tol, (krot, b, S, (a - c).abs().max().item())
            assert (a.cpu() - want).abs().max().item() < 1e-4
    # not right-padded -> padded path, same answer as transformers
    ids, types, mask = bert_oracle.synthetic_batch(6, 24, vocab=3000, seed=5, ragged=True)
    mask = torch.flip(mask, dims=[1]); ids = torch.flip(ids, dims=[1])             # left padding
    mask[3, 10] = 0                                                                  # a hole
    a = packed.encode_cls(ids, types, mask)
    assert packed.last_tokens == 6 * 24
    # (left-padded CLS position differs from the reference's [:, 0] pooling only through which token sits at 0:
    #  the oracle pools position 0 too, so both see the same padded token there)
    want = bert_oracle.encode_cls(model, ids, types, mask)
    ok = mask[:, 0] != 0                                                             # rows whose position 0 is a real token
    assert (a.cpu()[ok] - want[ok]).abs().max().item() < 1e-4 if ok.any() else True


@pytest.mark.parametrize("hidden,layers,heads,inter,local,b,S", [
    (128, 4, 2, 192, 16, 9, 40),          # windows of +-8 cut inside the ragged sequences; T = 360 -> planes
    (768, 3, 12, 1152, 128, 5, 70),       # ModernBERT-base width, two key tiles
    (128, 3, 2, 192, 128, 3, 12),         # small-M GEMM path
])
def test_modernbert_padding_free_path_equals_padded_path(hidden, layers, heads, inter, local, b, S, cuda_dev):
    """ac_modernbert_encode_cls_packed (padding tokens left out; RoPE positions per sequence) gives the CLS vectors of the
    padded forward, and both match transformers within the 1e-4 bar."""
    from adaptive_classifier.encoder import HipModernBertEncoder
    from oracle import bert_oracle
    model = bert_oracle.make_modernbert(hidden, layers, heads, inter, vocab=2000, max_pos=1024, local_attention=local, seed=3,
                                        init_scale=4.0)
    ids, _, mask = bert_oracle.synthetic_batch(b, S, vocab=2000, seed=77, ragged=True)
    ids[:, 0] = 1
    want = bert_oracle.encode_cls_modernbert(model, ids, mask)
    packed = HipModernBertEncoder(model, device=cuda_dev, unpad=True)
    padded = HipModernBertEncoder(model, device=cuda_dev, unpad=False)
    got_p = packed.encode_cls(ids, None, mask).cpu()
    got_f = padded.encode_cls(ids, None, mask).cpu()
    assert packed.last_tokens == int(mask.sum()) < b * S == padded.last_tokens
    assert (got_p - got_f).abs().max().item() < 2e-6
    assert (got_p - want).abs().max().item() < 1e-4


@pytest.mark.parametrize("hidden,layers,heads,inter,b,S,ragged", [
    (768, 3, 12, 3072, 200, 32, True),     # ~4000 packed rows, 32 row panels x 6 column tiles: the bench's launch shape
    (768, 2, 12, 3072, 24, 16, False),     # 384 rows = exactly 3 panels, padded layout
    (128, 3, 2, 512, 40, 16, True),        # one column tile per panel (H = 128), ragged last panel
    (1024, 3, 16, 4096, 20, 24, True),     # 8 column tiles per panel (bert-large width)
])
@pytest.mark.parametrize("regime", REGIMES)
def test_layernorm_fused_into_the_gemm_epilogue(hidden, layers, heads, inter, b, S, ragged, regime, cuda_dev):
    """x = LayerNorm(x + A W^T + b) inside the attention-output / FFN2 GEMM epilogues (gemm_pipe.hip EPI_BIAS_RES_LN: the tiles
    of a row panel exchange (mean, M2) partials): the fused launches really run (counter), their result equals the separate
    LayerNorm launches to rounding and transformers to the 1e-4 bar -- also with LayerNorm outlier channels (peaked regime)."""
    from adaptive_classifier import _native as nv
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    vocab = 2000
    model = bert_oracle.make_bert(hidden, layers, heads, inter, vocab=vocab, seed=3, **regime_kw(regime, hidden))
    ids, types, mask = bert_oracle.synthetic_batch(b, S, vocab=vocab, seed=99, ragged=ragged)
    want = bert_oracle.encode_cls(model, ids, types, mask)
    enc = HipBertEncoder(model, device=cuda_dev)
    lib = nv.lib()
    try:
        nv.check(lib.ac_gemm_set_ln_fusion(0), "ac_gemm_set_ln_fusion")
        n0 = lib.ac_gemm_ln_fusion_launches()
        separate = enc.encode_cls(ids, types, mask).cpu()
        assert lib.ac_gemm_ln_fusion_launches() == n0
        nv.check(lib.ac_gemm_set_ln_fusion(1), "ac_gemm_set_ln_fusion")
        fused = enc.encode_cls(ids, types, mask).cpu()
        assert lib.ac_gemm_ln_fusion_launches() == n0 + 2 * (layers - 1)       # every layer but the CLS-only last one
        assert not enc.ln_fusion_aborted()
        again = enc.encode_cls(ids, types, mask).cpu()
    finally:
        nv.check(lib.ac_gemm_set_ln_fusion(1), "ac_gemm_set_ln_fusion")
    assert torch.equal(fused, again)                                            # the exchange order does not leak into the result
    assert (fused - separate).abs().max().item() < 5e-6, (fused - separate).abs().max().item()
    for got in (fused, separate):
        assert (got - want).abs().max().item() < 1e-4


def test_starved_layernorm_exchange_gives_up_loudly(cuda_dev):
    """The give-up path of the fused-LayerNorm epilogue for real: with the exchange starved (every tile waits for one arrival
    more than will come) the launches must END within their bounded wait; with verify=False the rows come out NaN and the
    verdict says so; with verify=True (default) the SAME call notices, switches the fusion off and returns finite, correct
    rows -- one place that guarantees finite embeddings."""
    import time
    from adaptive_classifier import _native as nv
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    model = bert_oracle.make_bert(128, 3, 2, 512, vocab=2000, seed=5)
    ids, types, mask = bert_oracle.synthetic_batch(24, 16, vocab=2000, seed=7, ragged=False)
    want = bert_oracle.encode_cls(model, ids, types, mask)
    enc = HipBertEncoder(model, device=cuda_dev)
    lib = nv.lib()
    try:
        nv.check(lib.ac_gemm_set_ln_fusion(2), "ac_gemm_set_ln_fusion")
        t0 = time.time()
        bad = enc.encode_cls(ids, types, mask, verify=False).cpu()
        assert time.time() - t0 < 30.0
        assert torch.isnan(bad).all()
        assert enc.ln_fusion_aborted()
        nv.check(lib.ac_gemm_set_ln_fusion(0), "ac_gemm_set_ln_fusion")
        good = enc.encode_cls(ids, types, mask, verify=False).cpu()
        assert not enc.ln_fusion_aborted()                       # the verdict of a call starts clean
        assert (good - want).abs().max().item() < 1e-4
        # the default contract: still starved, but the call repairs itself
        nv.check(lib.ac_gemm_set_ln_fusion(2), "ac_gemm_set_ln_fusion")
        n_gave_up, n_fused = enc.ln_gave_up, lib.ac_gemm_ln_fusion_launches()
        healed = enc.encode_cls(ids, types, mask).cpu()
        assert enc.ln_gave_up == n_gave_up + 1 and lib.ac_gemm_ln_fusion_launches() > n_fused
        assert torch.isfinite(healed).all() and (healed - want).abs().max().item() < 1e-4
        n_fused = lib.ac_gemm_ln_fusion_launches()
        enc.encode_cls(ids, types, mask)                         # the fusion is off for THIS encoder now (per-object option)
        assert lib.ac_gemm_ln_fusion_launches() == n_fused and enc.ln_gave_up == n_gave_up + 1 and enc.ln_fusion is False
    finally:
        nv.check(lib.ac_gemm_set_ln_fusion(1), "ac_gemm_set_ln_fusion")
    assert (enc.encode_cls(ids, types, mask).cpu() - want).abs().max().item() < 1e-4
    # ... and for nobody else: a second encoder in the same process still fuses (round 4 switched the whole process off)
    other = HipBertEncoder(model, device=cuda_dev)
    n_fused = lib.ac_gemm_ln_fusion_launches()
    assert (other.encode_cls(ids, types, mask).cpu() - want).abs().max().item() < 1e-4
    assert lib.ac_gemm_ln_fusion_launches() == n_fused + 4 and other.ln_fusion is None
    n_fused = lib.ac_gemm_ln_fusion_launches()
    enc.encode_cls(ids, types, mask)
    assert lib.ac_gemm_ln_fusion_launches() == n_fused


@pytest.mark.parametrize("model_type,hidden,layers,heads,inter,b,S,ragged", [
    ("roberta", 128, 2, 2, 512, 24, 16, True),          # packed (padding-free) path
    ("roberta", 128, 3, 2, 512, 40, 16, False),         # full rows, LayerNorm-fused GEMM epilogues
    ("xlm-roberta", 768, 2, 12, 3072, 8, 24, True),     # bert-base width
    ("roberta", 128, 2, 2, 512, 1, 12, False),          # a single short query: the one-launch kernel
])
def test_roberta_family_matches_transformers(model_type, hidden, layers, heads, inter, b, S, ragged, cuda_dev):
    """RoBERTa / XLM-RoBERTa (multilingual-e5-*): the BERT kernels with the position table entered at padding_idx + 1
    (modeling_roberta.py create_position_ids_from_input_ids, right-padded inputs).  Unit-norm CLS vs transformers fp32 <= 1e-4."""
    from adaptive_classifier.encoder import HipBertEncoder, make_encoder
    from oracle import bert_oracle
    model = bert_oracle.make_roberta(hidden, layers, heads, inter, vocab=2000, max_pos=S + 8, seed=4, model_type=model_type)
    ids, mask = bert_oracle.roberta_batch(b, S, vocab=2000, seed=17, ragged=ragged)
    want = bert_oracle.encode_cls_roberta(model, ids, mask)
    enc = make_encoder(model, device=cuda_dev)
    assert isinstance(enc, HipBertEncoder)
    got = enc.encode_cls(ids, None, mask).cpu()
    assert (got - want).abs().max().item() < 1e-4, (got - want).abs().max().item()
    assert enc.ccfg.max_pos == S + 8 - 2                       # the table is entered at padding_idx + 1 = 2
    # ... and the offset matters: with the rows of the position table moved by one the result is something else
    sd = model.state_dict()
    sd["embeddings.position_embeddings.weight"] = torch.roll(sd["embeddings.position_embeddings.weight"], 1, 0)
    other = bert_oracle.make_roberta(hidden, layers, heads, inter, vocab=2000, max_pos=S + 8, seed=4, model_type=model_type)
    other.load_state_dict(sd)
    assert (bert_oracle.encode_cls_roberta(other, ids, mask) - want).abs().max().item() > 1e-3


def test_electra_body_matches_transformers(cuda_dev):
    """ELECTRA (google/electra-base-*: embedding_size == hidden_size) is the BERT block under BERT's parameter names."""
    from transformers import ElectraConfig, ElectraModel
    from adaptive_classifier import _native as nv
    from adaptive_classifier.encoder import make_encoder
    from oracle import bert_oracle
    cfg = ElectraConfig(vocab_size=2000, embedding_size=128, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                        intermediate_size=512, max_position_embeddings=64)
    cfg._attn_implementation = "eager"
    torch.manual_seed(9)
    model = ElectraModel(cfg).eval()
    ids, types, mask = bert_oracle.synthetic_batch(20, 16, vocab=2000, seed=5)
    types[:, 8:] = 1
    want = bert_oracle.encode_cls(model, ids, types, mask)
    got = make_encoder(model, device=cuda_dev).encode_cls(ids, types, mask).cpu()
    assert (got - want).abs().max().item() < 1e-4
    small = ElectraModel(ElectraConfig(vocab_size=2000, embedding_size=64, hidden_size=128, num_hidden_layers=1,
                                       num_attention_heads=2, intermediate_size=512))
    with pytest.raises(nv.NativeError, match="embeddings_project"):
        make_encoder(small, device=cuda_dev)


def test_layernorm_verdict_is_sticky_over_the_chunks_of_a_call(cuda_dev, monkeypatch):
    """A batch beyond MAX_TOKENS runs as several native calls sharing one workspace.  The give-up of an EARLY chunk must not be
    erased by a later chunk (here the last chunk is too small for the fused epilogue and comes out finite): the verdict word
    is sticky over the call, and the default verify=True repairs every chunk."""
    from adaptive_classifier import _native as nv, encoder as encmod
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    model = bert_oracle.make_bert(128, 3, 2, 512, vocab=2000, seed=5)
    ids, types, mask = bert_oracle.synthetic_batch(24, 16, vocab=2000, seed=8, ragged=False)
    want = bert_oracle.encode_cls(model, ids, types, mask)
    enc = HipBertEncoder(model, device=cuda_dev)
    monkeypatch.setattr(encmod, "MAX_TOKENS", 14 * 16)           # chunks of 14 sequences (224 rows: fused) and 10 (160 rows: not)
    lib = nv.lib()
    try:
        nv.check(lib.ac_gemm_set_ln_fusion(2), "ac_gemm_set_ln_fusion")
        bad = enc.encode_cls(ids, types, mask, verify=False).cpu()
        assert torch.isnan(bad[:14]).all() and torch.isfinite(bad[14:]).all()
        assert enc.ln_fusion_aborted()                           # although the LAST native call of the batch was clean
        healed = enc.encode_cls(ids, types, mask).cpu()
        assert torch.isfinite(healed).all() and (healed - want).abs().max().item() < 1e-4
    finally:
        nv.check(lib.ac_gemm_set_ln_fusion(1), "ac_gemm_set_ln_fusion")


def test_per_call_options_do_not_leak_between_objects(cuda_dev):
    """VERDICT r04 item 6 / SURVEY 8b "no global mutable state": arithmetic and LayerNorm fusion are PER-CALL options inside
    ac_bert_config.  Three classifiers of three arithmetics on ONE shared encoder plus an encoder with its own option, alive in
    one process, calls interleaved: each returns, bit for bit, what a process running only that arithmetic returns (the
    process-wide hook ac_gemm_set_arith is used here only to produce those single-arithmetic references), and the process-wide
    value is never touched by any of them."""
    from adaptive_classifier import AdaptiveClassifier, _native as nv
    from adaptive_classifier.encoder import HipBertEncoder
    from helpers import HashTokenizer
    from oracle import bert_oracle
    lib = nv.lib()
    model = bert_oracle.make_bert(128, 3, 2, 512, vocab=2000, seed=11)
    ids, types, mask = bert_oracle.synthetic_batch(24, 16, vocab=2000, seed=12, ragged=False)    # 384 rows: ring kernels, planes, fp16x2 applies
    enc = HipBertEncoder(model, device=cuda_dev)
    enc.enable_f16x2()                                     # planes present; USED only by calls whose arithmetic says so
    before = lib.ac_gemm_get_arith()
    assert before == nv.AC_GEMM_BF16X3
    ref = {}
    try:                                                   # single-arithmetic references through the process-wide hook
        for name, mode in (("f32", 0), ("bf16x3", 1), ("f16x2", 2)):
            nv.check(lib.ac_gemm_set_arith(mode), "ac_gemm_set_arith")
            ref[name] = enc.encode_cls(ids, types, mask).clone()
    finally:
        nv.check(lib.ac_gemm_set_arith(before), "ac_gemm_set_arith")
    assert not torch.equal(ref["f32"], ref["bf16x3"]) and not torch.equal(ref["bf16x3"], ref["f16x2"])   # three arithmetics indeed
    # per-call option on the shared encoder, interleaved
    for _ in range(2):
        for name in ("f16x2", "f32", "bf16x3", "f32", "f16x2"):
            assert torch.equal(enc.encode_cls(ids, types, mask, arith=name), ref[name]), name
            assert lib.ac_gemm_get_arith() == before
    assert torch.equal(enc.encode_cls(ids, types, mask), ref["bf16x3"])           # no option: the process default
    # classifiers: config["gemm_arith"] is per object, even on a shared encoder
    clfs = {name: AdaptiveClassifier("synthetic", device="cuda:0", config={"gemm_arith": name}, encoder=enc, tokenizer=HashTokenizer())
            for name in ("f32", "bf16x3", "f16x2")}
    plain = AdaptiveClassifier("synthetic", device="cuda:0", encoder=enc, tokenizer=HashTokenizer())
    for _ in range(2):
        for name in ("f32", "f16x2", "bf16x3"):
            assert torch.equal(clfs[name]._encode_tokens(ids, types, mask), ref[name]), name
        assert torch.equal(plain._encode_tokens(ids, types, mask), ref["bf16x3"])
    assert lib.ac_gemm_get_arith() == before
    with pytest.raises(ValueError, match="gemm_arith"):
        AdaptiveClassifier("synthetic", device="cuda:0", config={"gemm_arith": "fp8"}, encoder=enc, tokenizer=HashTokenizer())
    # an encoder-level option, and the LayerNorm fusion per encoder
    enc2 = HipBertEncoder(model, device=cuda_dev).set_arith("f32")
    assert torch.equal(enc2.encode_cls(ids, types, mask), ref["f32"]) and torch.equal(enc.encode_cls(ids, types, mask), ref["bf16x3"])
    n0 = lib.ac_gemm_ln_fusion_launches()
    fused = enc.encode_cls(ids, types, mask)
    n1 = lib.ac_gemm_ln_fusion_launches()
    enc3 = HipBertEncoder(model, device=cuda_dev).disable_ln_fusion()
    unfused = enc3.encode_cls(ids, types, mask)
    assert n1 > n0 and lib.ac_gemm_ln_fusion_launches() == n1                      # enc fuses, enc3 does not, same process
    assert (fused - unfused).abs().max().item() < 5e-6
    enc.encode_cls(ids, types, mask)
    assert lib.ac_gemm_ln_fusion_launches() == n1 + (n1 - n0)                      # ... and enc still does
    # a C caller's zero-initialised option words mean "defaults"; out-of-range words are refused
    import ctypes
    bad = enc._call_cfg()
    bad.gemm_arith_opt = 9
    need = ctypes.c_size_t(0)
    assert lib.ac_bert_workspace(ctypes.byref(bad), 4, 16, ctypes.byref(need)) != 0 and b"options" in lib.ac_last_error()


@pytest.mark.parametrize("hidden,layers,heads,inter,b,S,ragged", [
    (768, 3, 12, 3072, 200, 32, True),     # the bench's launch shape: ~4000 packed rows, 16 row tiles, straddling sequences
    (768, 2, 12, 3072, 64, 32, False),     # every text at 32 tokens: unpacked, the mask is all ones (synthesised offsets), no straddler
    (128, 3, 2, 512, 40, 16, True),        # two heads, ragged last row tile (T < 2 tiles)
    (1024, 2, 16, 4096, 30, 24, True),     # bert-large width: 16 heads
    (768, 2, 12, 3072, 24, 64, True),      # sequences of up to 64 tokens: three staging passes, two query tiles
    (768, 2, 12, 3072, 30, 50, False),     # 50-token rows, unpacked: sequences straddle every tile boundary
    (128, 2, 2, 512, 700, 8, True),        # many short sequences (2 .. 8 tokens): ~20 sequences per staging pass and wave rounds
    (768, 2, 12, 3072, 640, 32, True),     # ~12 800 rows = 50 row tiles x 12 heads: more tiles than CUs -> several rounds, no in-launch exchange
])
@pytest.mark.parametrize("regime", REGIMES)
def test_attention_fused_into_the_qkv_gemm_epilogue(hidden, layers, heads, inter, b, S, ragged, regime, cuda_dev, monkeypatch):
    """Self-attention inside the QKV projection's epilogue (gemm_pipe.hip EPI_QKV_ATTN: a tile = 256 token rows x one head's
    q | k | v, sequences finished from LDS, boundary-straddling sequences by attention_mfma_kernel's boundary mode): the fused
    launches really run (counter), the CLS vectors are BIT-IDENTICAL to the two-launch route (same instructions on the same
    values) and meet the 1e-4 bar against transformers in both attention regimes."""
    from adaptive_classifier import _native as nv
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    vocab = 2000
    model = bert_oracle.make_bert(hidden, layers, heads, inter, vocab=vocab, seed=5, **regime_kw(regime, hidden))
    ids, types, mask = bert_oracle.synthetic_batch(b, S, vocab=vocab, seed=77, ragged=ragged)
    want = bert_oracle.encode_cls(model, ids, types, mask)
    enc = HipBertEncoder(model, device=cuda_dev)
    lib = nv.lib()
    monkeypatch.setenv("AC_QKV_ATTN_FUSION", "0")
    n0 = lib.ac_gemm_qkv_attn_launches()
    separate = enc.encode_cls(ids, types, mask).cpu()
    assert lib.ac_gemm_qkv_attn_launches() == n0
    monkeypatch.delenv("AC_QKV_ATTN_FUSION")
    fused = enc.encode_cls(ids, types, mask).cpu()
    assert lib.ac_gemm_qkv_attn_launches() == n0 + (layers - 1)                # every layer but the CLS-only last one
    assert torch.equal(fused, separate), (fused - separate).abs().max().item()
    assert (fused - want).abs().max().item() < 1e-4
    # the two ways a sequence that straddles a 256-row tile boundary is finished: inside the launch (the two tiles exchange its
    # rows; one-round launches, the default above where it applies) and by attention_mfma_kernel's boundary mode afterwards
    monkeypatch.setenv("AC_QKV_ATTN_EXCHANGE", "0")
    boundary = enc.encode_cls(ids, types, mask).cpu()
    monkeypatch.delenv("AC_QKV_ATTN_EXCHANGE")
    assert torch.equal(boundary, separate)
    assert not enc.ln_fusion_aborted()


def test_attention_fusion_leaves_long_or_masked_batches_alone(cuda_dev):
    """Sequences longer than 64 tokens, and an unpacked batch whose mask has holes (not a prefix: the padded route with its key
    mask), keep the stand-alone attention launch."""
    from adaptive_classifier import _native as nv
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    model = bert_oracle.make_bert(128, 2, 2, 512, vocab=2000, seed=5)
    enc = HipBertEncoder(model, device=cuda_dev)
    lib = nv.lib()
    ids, types, mask = bert_oracle.synthetic_batch(6, 80, vocab=2000, seed=3, ragged=True)
    n0 = lib.ac_gemm_qkv_attn_launches()
    got = enc.encode_cls(ids, types, mask).cpu()
    assert lib.ac_gemm_qkv_attn_launches() == n0
    assert (got - bert_oracle.encode_cls(model, ids, types, mask)).abs().max().item() < 1e-4
    ids, types, mask = bert_oracle.synthetic_batch(40, 16, vocab=2000, seed=4, ragged=False)
    mask[3, 5] = 0                                                             # a hole: not a right-padded mask
    got = enc.encode_cls(ids, types, mask).cpu()
    assert lib.ac_gemm_qkv_attn_launches() == n0
    assert (got - bert_oracle.encode_cls(model, ids, types, mask)).abs().max().item() < 1e-4


def test_modernbert_gemm_arith_is_a_per_call_option(cuda_dev):
    """`config["gemm_arith"]` / `encode_cls(arith=...)` reaches the ModernBERT encoder too -- a per-call option
    word inside ac_modernbert_config (thread-local call scope), not the process default.  "f32" runs the strict fp32-input MFMA
    kernels (different bits from the bf16x3 default, same 1e-4 bar), the process-wide arithmetic is never written, and a classifier
    configured with an arithmetic hands it to this encoder."""
    from adaptive_classifier import AdaptiveClassifier, _native as nv
    from adaptive_classifier.encoder import HipModernBertEncoder
    from oracle import bert_oracle
    model = bert_oracle.make_modernbert(128, 3, 2, 256, vocab=2000, max_pos=1024, local_attention=16, seed=3, init_scale=4.0)
    ids, _, mask = bert_oracle.synthetic_batch(40, 24, vocab=2000, seed=7, ragged=True)      # ~600 rows: the planes GEMMs run
    ids[:, 0] = 1
    want = bert_oracle.encode_cls_modernbert(model, ids, mask)
    enc = HipModernBertEncoder(model, device=cuda_dev)
    before = nv.lib().ac_gemm_get_arith()
    dflt = enc.encode_cls(ids, None, mask).cpu()
    f32 = enc.encode_cls(ids, None, mask, arith="f32").cpu()
    again = enc.encode_cls(ids, None, mask).cpu()
    split = enc.encode_cls(ids, None, mask, arith="bf16x3").cpu()
    assert nv.lib().ac_gemm_get_arith() == before
    assert torch.equal(dflt, again) and torch.equal(dflt, split)            # (the process default IS bf16x3 in this suite)
    assert not torch.equal(dflt, f32)                                       # another arithmetic really ran ...
    for got in (dflt, f32):
        assert (got - want).abs().max().item() < 1e-4                       # ... to the same bar
    with pytest.raises(ValueError):
        enc.encode_cls(ids, None, mask, arith="fp8")
    clf = AdaptiveClassifier("synthetic-modernbert", device=str(cuda_dev), encoder=enc, config={"gemm_arith": "f32"})
    assert torch.equal(clf._encode_tokens(ids, None, mask).cpu(), f32)


def test_unpad_one_call_equals_pack_then_encode(cuda_dev, monkeypatch):
    """ac_bert_encode_cls_unpad (one workgroup derives the packing, the row-tile table and the zeroed exchange words; the embedding
    kernel starts before the host knows the token count; no stream synchronisation) against the separate form it replaces
    (ac_bert_pack -> 16-byte read-back -> ac_bert_encode_cls_packed): the same CLS vectors BIT FOR BIT, the same token count,
    over ragged batches (packed path, with and without the fused attention / LayerNorm epilogues), an all-ones mask (padded path
    without a mask), a left-padded mask and an empty row (padded path with the mask), a ragged multi-chunk batch with a sticky
    verdict, and a width whose plan differs between the fewest and the most rows a batch can have (embedding launched late)."""
    import ctypes
    from adaptive_classifier import _native as nv, encoder as encmod
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    model = bert_oracle.make_bert(768, 3, 12, 3072, vocab=3000, seed=2)
    enc = HipBertEncoder(model, device=cuda_dev, unpad=True)

    def both(ids, types, mask):
        monkeypatch.setenv("AC_BERT_UNPAD_ONE_CALL", "0")
        a = enc.encode_cls(ids, types, mask).clone(); ta = enc.last_tokens
        monkeypatch.setenv("AC_BERT_UNPAD_ONE_CALL", "1")
        enc._ws[256:].fill_(0x7f)                                # nothing of the earlier call may be what makes this one right
        c = enc.encode_cls(ids, types, mask).clone(); tc = enc.last_tokens
        assert ta == tc, (ta, tc)
        assert torch.equal(torch.isnan(a), torch.isnan(c)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(c)), (a - c).abs().max().item()
        return c

    for (b, S, seed) in [(256, 32, 1234), (37, 48, 3), (5, 130, 9), (3, 16, 4), (700, 12, 6)]:
        ids, types, mask = bert_oracle.synthetic_batch(b, S, vocab=3000, seed=seed, ragged=True)
        types[:, S // 3:] = 1
        got = both(ids, types, mask)
        assert enc.last_tokens == int(mask.sum())
        assert (got.cpu() - bert_oracle.encode_cls(model, ids, types, mask)).abs().max().item() < 1e-4
    # every row full -> the [b, S] forward without a mask
    ids, types, mask = bert_oracle.synthetic_batch(40, 16, vocab=3000, seed=7, ragged=False)
    both(ids, types, mask)
    assert enc.last_tokens == 40 * 16
    # left padding, a hole, an empty row -> the [b, S] forward with the mask
    ids, types, mask = bert_oracle.synthetic_batch(6, 24, vocab=3000, seed=5, ragged=True)
    for variant in range(3):
        m = mask.clone()
        if variant == 0: m = torch.flip(m, dims=[1])
        elif variant == 1: m[3, 2] = 0
        else: m[4, :] = 0
        got = both(ids, types, m)
        assert enc.last_tokens == 6 * 24
    # the entry reports its path
    cfg = enc._call_cfg(None)
    total, path = ctypes.c_int(0), ctypes.c_int(0)
    out = torch.empty(6, 768, device=cuda_dev)
    for m, want_path in ((mask, nv.AC_BERT_PATH_PACKED), (torch.ones_like(mask), nv.AC_BERT_PATH_PADDED), (torch.flip(mask, dims=[1]), nv.AC_BERT_PATH_PADDED_MASK)):
        idd, md = ids.to(cuda_dev), m.to(cuda_dev).contiguous()
        nv.check(nv.lib().ac_bert_encode_cls_unpad(ctypes.byref(cfg), ctypes.byref(enc.weights), nv.ptr(idd), None, nv.ptr(md), 6, 24, nv.ptr(out), 768,
                                                   nv.ptr(enc._ws), enc._ws.numel(), 1, ctypes.byref(total), ctypes.byref(path), nv.stream_ptr(cuda_dev)), "unpad")
        assert path.value == want_path and total.value == (int(m.sum()) if want_path == nv.AC_BERT_PATH_PACKED else 6 * 24)
    # several chunks of one call share the workspace; the verdict words are cleared by the FIRST chunk only
    max_tokens = encmod.MAX_TOKENS
    monkeypatch.setattr(encmod, "MAX_TOKENS", 60 * 32)
    ids, types, mask = bert_oracle.synthetic_batch(150, 32, vocab=3000, seed=11, ragged=True)
    got = both(ids, types, mask)
    assert torch.isfinite(got).all() and enc.last_tokens == int(mask.sum()) and not enc.ln_fusion_aborted()
    monkeypatch.setattr(encmod, "MAX_TOKENS", max_tokens)
    # 100 sequences of <= 2 tokens: 100 rows do not take operand planes, 200 do -> the embedding waits for the count
    ids, types, mask = bert_oracle.synthetic_batch(100, 2, vocab=3000, seed=13, ragged=True)
    both(ids, types, mask)
    # bad arguments are refused before anything is launched
    rc = nv.lib().ac_bert_encode_cls_unpad(ctypes.byref(cfg), ctypes.byref(enc.weights), nv.ptr(idd), None, None, 6, 24, nv.ptr(out), 768,
                                           nv.ptr(enc._ws), enc._ws.numel(), 1, ctypes.byref(total), ctypes.byref(path), nv.stream_ptr(cuda_dev))
    assert rc == -1


@pytest.mark.parametrize("hidden,layers,heads,inter,b,S", [(768, 2, 12, 3072, 256, 32), (768, 2, 12, 3072, 70, 48), (1024, 2, 16, 4096, 96, 24),
                                                            (384, 2, 12, 1536, 130, 16)])
def test_last_layer_tail_in_one_launch(hidden, layers, heads, inter, b, S, cuda_dev, monkeypatch):
    """The CLS-only last layer ends with ONE launch (splitk_ln_normalize_kernel: the K slices of the FFN-down product + bias +
    residual, LayerNorm, F.normalize) instead of reduce -> LayerNorm -> normalize (AC_BERT_TAIL_FUSED=0), and its CLS attention
    runs on 32-key tiles when no sequence is longer: the same unit vectors to fp32 rounding (bar 2e-6: the norm's sum runs in a
    different lane order), 1e-4 from transformers, also into a wider output buffer (the padding columns are zeroed)."""
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    model = bert_oracle.make_bert(hidden, layers, heads, inter, vocab=3000, seed=4)
    ids, types, mask = bert_oracle.synthetic_batch(b, S, vocab=3000, seed=21, ragged=True)
    enc = HipBertEncoder(model, device=cuda_dev)
    monkeypatch.setenv("AC_BERT_TAIL_FUSED", "0")
    a = enc.encode_cls(ids, types, mask).clone()
    monkeypatch.setenv("AC_BERT_TAIL_FUSED", "1")
    c = enc.encode_cls(ids, types, mask).clone()
    assert torch.isfinite(c).all()
    assert (a - c).abs().max().item() <= 2e-6, (a - c).abs().max().item()
    assert (c.cpu() - bert_oracle.encode_cls(model, ids, types, mask)).abs().max().item() < 1e-4
    wide = torch.full((b, hidden + 24), 7.0, device=cuda_dev)
    enc.encode_cls(ids, types, mask, out=wide)
    assert torch.equal(wide[:, :hidden], c) and bool((wide[:, hidden:] == 0).all())
