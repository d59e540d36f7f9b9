"""AC_GEMM_F16X2 (opt-in): every operand of the BERT encoder's token-row GEMMs as TWO fp16 terms of x 2^s (22 significant
bits), three fp16 MFMA products per tile, fp32 accumulate (include/acamd.h).  Not fp32-exact on its inputs, and said so: these
tests MEASURE it next to the fp32-input MFMA and the bf16x3 split against fp64, and pin what the mode promises --
  * the planes are h = fp16(x 2^s), l = fp16(x 2^s - h), bit for bit;
  * per element |C - fp64| <= 3.5 * 2^-22 * sum|a||w| + the fp32 accumulation term (inside the a-priori bound K 2^-24 sum|a||w|
    of an fp32 dot product for the encoder's K);
  * the encoder stays within SURVEY 8c's 1e-4 of transformers fp32 (measured: ~1e-6), for flat and peaked attention;
  * an operand beyond the fp16 range gives NaN rows, never a wrong finite number, and the encoder then repeats the call in
    bf16x3 by itself.
The reference computes these products in torch fp32 (transformers BertModel called at classifier.py:1271)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F32, BF16X3, F16X2 = 0, 1, 2
ACT_LOG2, W_LOG2 = 6, 10


@pytest.fixture()
def arith():
    from adaptive_classifier import _native as nv
    lib = nv.lib()
    before = lib.ac_gemm_get_arith()
    yield lambda mode: nv.check(lib.ac_gemm_set_arith(mode), "ac_gemm_set_arith")
    lib.ac_gemm_set_arith(before)
    lib.ac_gemm_set_pipe_table_f16(None)
    lib.ac_gemm_set_krot(0)


def _planes_f16(nv, dev, Xd, log2):
    rows, K = Xd.shape
    P = torch.empty(2 * rows * K, dtype=torch.int16, device=dev)
    nv.check(nv.lib().ac_split_f16x2(nv.ptr(Xd), K, rows, K, log2, nv.ptr(P), nv.stream_ptr(dev)), "ac_split_f16x2")
    return P


def _unplane(P, rows, K):
    """planes[p][k // 8][row][k % 8] -> [p][row][k] float64"""
    P = P.cpu().numpy().view(np.float16).reshape(2, K // 8, rows, 8).transpose(0, 2, 1, 3).reshape(2, rows, K)
    return P.astype(np.float64)


def _linear_f16(nv, dev, A, W, b, R, act, planes_out=False):
    M, K = A.shape
    N = W.shape[0]
    Ad, Wd, bd = (torch.from_numpy(x).to(dev) for x in (A, W, b))
    Rd = torch.from_numpy(R).to(dev) if R is not None else None
    Ap, Wp = _planes_f16(nv, dev, Ad, ACT_LOG2), _planes_f16(nv, dev, Wd, W_LOG2)
    if planes_out:
        Cp = torch.empty(3 * M * N, dtype=torch.int16, device=dev)       # (the encoder's buffers hold three planes; two are used)
        nv.check(nv.lib().ac_linear_f16x2(nv.ptr(Ap), nv.ptr(Wp), nv.ptr(bd), None, 0, None, N, nv.ptr(Cp), M, N, K, act,
                                          nv.stream_ptr(dev)), "ac_linear_f16x2")
        hl = _unplane(Cp[:2 * M * N], M, N)
        return (hl[0] + hl[1]) / 2.0 ** ACT_LOG2
    C = torch.empty((M, N), device=dev)
    nv.check(nv.lib().ac_linear_f16x2(nv.ptr(Ap), nv.ptr(Wp), nv.ptr(bd), nv.ptr(Rd) if R is not None else None, N, nv.ptr(C), N,
                                      None, M, N, K, act, nv.stream_ptr(dev)), "ac_linear_f16x2")
    return C.cpu().numpy().astype(np.float64)


def _linear_f32api(nv, dev, A, W, b, R, act):
    M, K = A.shape
    N = W.shape[0]
    Ad, Wd, bd = (torch.from_numpy(x).to(dev) for x in (A, W, b))
    Rd = torch.from_numpy(R).to(dev) if R is not None else None
    C = torch.empty((M, N), device=dev)
    nv.check(nv.lib().ac_linear_f32(nv.ptr(Ad), K, nv.ptr(Wd), K, nv.ptr(bd), nv.ptr(Rd) if R is not None else None, N, nv.ptr(C),
                                    N, M, N, K, act, nv.stream_ptr(dev)), "ac_linear_f32")
    return C.cpu().numpy().astype(np.float64)


def _ref(A, W, b, R, act):
    z = A.astype(np.float64) @ W.astype(np.float64).T + b
    if act == 2:
        from scipy.special import erf
        z = 0.5 * z * (1 + erf(z / np.sqrt(2)))
    if R is not None:
        z = z + R
    return z


def test_f16x2_planes_are_the_two_fp16_terms(cuda_dev):
    from adaptive_classifier import _native as nv
    rng = np.random.default_rng(5)
    rows, K = 77, 40
    X = (rng.standard_normal((rows, K)) * 10.0 ** rng.uniform(-7, 1.5, (rows, K))).astype(np.float32)
    X[0, :6] = [0.0, -0.0, 1.0, -1.5, 1000.0, -1023.0]
    for log2 in (ACT_LOG2, W_LOG2):
        Xs = X if log2 == ACT_LOG2 else (X / 32.0).astype(np.float32)
        hl = _unplane(_planes_f16(nv, cuda_dev, torch.from_numpy(Xs).to(cuda_dev), log2), rows, K)
        xs = Xs.astype(np.float32) * np.float32(2.0 ** log2)                       # exact (power of two, no overflow here)
        h = xs.astype(np.float16)
        l = (xs - h.astype(np.float32)).astype(np.float16)                        # the fp32 subtraction is exact
        assert np.array_equal(hl[0], h.astype(np.float64)) and np.array_equal(hl[1], l.astype(np.float64)), log2
        err = np.abs(xs.astype(np.float64) - hl[0] - hl[1])
        assert np.all(err <= np.maximum(2.0 ** -22 * np.abs(xs), 2.0 ** -25)), log2


@pytest.mark.parametrize("M,N,K,act,res", [
    (5141, 2304, 768, 0, False),    # fused QKV at BASELINE configs[1]'s packed row count
    (5141, 768, 768, 0, True),      # attention output + residual
    (5141, 768, 3072, 0, True),     # FFN down + residual
    (8192, 3072, 768, 0, False),    # FFN up (fp32 rows out)
    (20564, 1024, 1024, 0, True),   # bert-large / e5-large rows of configs[4]
    (333, 264, 64, 0, True),        # ragged edges, two k-stages
])
def test_f16x2_gemm_error_measured_against_fp64(M, N, K, act, res, cuda_dev, arith):
    from adaptive_classifier import _native as nv
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    A[:, ::97] *= 20.0                                                  # LayerNorm-style outlier channels
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if res else None
    want = _ref(A, W, b, R, act)
    S = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T
    got = _linear_f16(nv, cuda_dev, A, W, b, R, act)
    e = np.abs(got - want)
    # what the mode promises: operand rounding + dropped l.l (3 * 2^-22) with slack, plus the fp32 accumulate / epilogue roundings
    assert np.all(e <= (3.5 * 2.0 ** -22 + (3 * K / 16 + 4) * 2.0 ** -24) * S + 1e-6), (e.max(), (e / S).max())
    assert np.all(e <= K * 2.0 ** -24 * S + 1e-6)                      # a-priori fp32 dot-product bound (K >= 64 here)
    # ... and the comparison this mode must be honest about: the other two arithmetics on the same operands
    arith(F32)
    e32 = np.abs(_linear_f32api(nv, cuda_dev, A, W, b, R, act) - want)
    arith(BF16X3)
    e3 = np.abs(_linear_f32api(nv, cuda_dev, A, W, b, R, act) - want)
    print(f"\n  {M}x{N}x{K}: max |err| / sum|a||w|  fp32-MFMA {(e32 / S).max():.2e}  bf16x3 {(e3 / S).max():.2e}  "
          f"fp16x2 {(e / S).max():.2e}   max |err| fp32 {e32.max():.2e} bf16x3 {e3.max():.2e} fp16x2 {e.max():.2e}")
    assert e.max() <= 32.0 * max(e32.max(), 1e-7), (e.max(), e32.max())   # "several times" the fp32 error, bounded here
    assert e.mean() <= 32.0 * e32.mean() + 1e-9


def test_f16x2_gemm_every_ring_configuration_agrees(cuda_dev, arith):
    """All fp16x2 tile configurations produce the same sums of the same products (order of the k-stages is the same): equal
    to the default's result to fp32 rounding of the accumulate, and each within the promised bound."""
    from adaptive_classifier import _native as nv
    rng = np.random.default_rng(11)
    M, N, K = 1100, 768, 256
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / 16).astype(np.float32)
    W[:, ::3] *= -3.0                                                   # asymmetric: a permuted fragment mapping cannot pass
    b = rng.standard_normal(N).astype(np.float32)
    want = _ref(A, W, b, None, 0)
    S = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T
    base = None
    nv.lib().ac_gemm_set_krot(0)                                        # (the default) k in order: tile-independent sums
    for cfg in (222232, 124261, 124262, 224242, 234232, 322432, 244232, 244242, 234242, 224262, 124282):
        nv.check(nv.lib().ac_gemm_set_pipe_table_f16(f"{N}x{K}={cfg}".encode()), "ac_gemm_set_pipe_table_f16")
        got = _linear_f16(nv, cuda_dev, A, W, b, None, 0)
        assert np.all(np.abs(got - want) <= (3.5 * 2.0 ** -22 + (3 * K / 16 + 4) * 2.0 ** -24) * S + 1e-6), cfg
        if base is None:
            base = got
        assert np.array_equal(got, base), cfg                           # same products, same order, whatever the tile
    # the rotated k order (ac_gemm_set_krot(1), experiment switch) with integer operands: every order is exact, so the exact product
    nv.lib().ac_gemm_set_krot(1)
    Ai = rng.integers(-3, 4, (M, K)).astype(np.float32)
    Wi = rng.integers(-3, 4, (N, K)).astype(np.float32)
    bi = rng.integers(-5, 6, N).astype(np.float32)
    wanti = _ref(Ai, Wi, bi, None, 0)
    for cfg in (222232, 124262, 224242, 234232, 322432, 244232, 244242, 234242, 224262, 124282):
        nv.check(nv.lib().ac_gemm_set_pipe_table_f16(f"{N}x{K}={cfg}".encode()), "ac_gemm_set_pipe_table_f16")
        assert np.array_equal(_linear_f16(nv, cuda_dev, Ai, Wi, bi, None, 0), wanti), cfg


def test_f16x2_planes_out_gelu_and_identity(cuda_dev, arith):
    """FFN1's form: bias + GELU, result emitted as the fp16x2 activation planes of the next GEMM; and A = I picks W^T out to
    the 22 bits the planes hold."""
    from adaptive_classifier import _native as nv
    rng = np.random.default_rng(17)
    M, N, K = 1024, 3072, 768
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    want = _ref(A, W, b, None, 2)
    got = _linear_f16(nv, cuda_dev, A, W, b, None, 2, planes_out=True)
    S = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T
    # the GEMM's own error passes through GELU (slope <= 1.13), then the result is rounded to two fp16 terms at scale 2^6
    assert np.all(np.abs(got - want) <= 1.2 * (3.5 * 2.0 ** -22 + (3 * K / 16 + 4) * 2.0 ** -24) * S
                  + np.maximum(2.0 ** -22 * np.abs(want), 2.0 ** -31) + 1e-6)
    eye = np.eye(256, 128, dtype=np.float32)
    Wn = (rng.standard_normal((264, 128)) * 10.0 ** rng.uniform(-4, 1, (264, 128))).astype(np.float32)
    got = _linear_f16(nv, cuda_dev, eye, Wn, np.zeros(264, np.float32), None, 0)
    assert np.all(np.abs(got[:128] - Wn.T) <= np.maximum(2.0 ** -22 * np.abs(Wn.T), 2.0 ** -35) * 1.01) and not got[128:].any()


def test_f16x2_out_of_range_operand_poisons_its_rows_with_nan(cuda_dev, arith):
    from adaptive_classifier import _native as nv
    rng = np.random.default_rng(19)
    M, N, K = 256, 256, 128
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / 8).astype(np.float32)
    A[7, 3] = 1024.0                                                    # 1024 * 2^6 = 65536 > fp16 max
    A[9, 5] = -5000.0
    got = _linear_f16(nv, cuda_dev, A, W, np.zeros(N, np.float32), None, 0)
    assert np.isnan(got[7]).all() and np.isnan(got[9]).all()
    ok = np.ones(M, bool)
    ok[[7, 9]] = False
    assert np.isfinite(got[ok]).all()
    A[7, 3], A[9, 5] = 1023.0, -1023.4                                  # the largest magnitudes that still fit
    got = _linear_f16(nv, cuda_dev, A, W, np.zeros(N, np.float32), None, 0)
    assert np.isfinite(got).all()
    assert np.allclose(got, A.astype(np.float64) @ W.astype(np.float64).T, atol=1e-3)


@pytest.mark.parametrize("peaked", [False, True])
@pytest.mark.parametrize("arch", ["base", "large"])
def test_encoder_under_f16x2_matches_transformers(arch, peaked, cuda_dev, arith):
    """1e-4 on the unit-norm CLS embedding.  Measured next to bf16x3 on the same batch; all forward forms
    (padded, padding-free, LayerNorm fused and not)."""
    from adaptive_classifier import _native as nv
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    kw = {"qk_scale": 5.0, "ln_outlier": 12.0} if peaked else {}       # trained-like: peaked softmax, LayerNorm outlier channels
    if arch == "base":
        model = bert_oracle.make_bert(768, 12, 12, 3072, vocab=2000, seed=0, **kw)
        b, S = 64, 32
    else:
        model = bert_oracle.make_bert(1024, 4, 16, 4096, vocab=2000, seed=1, **kw)
        b, S = 48, 32
    import copy
    ids, types, mask = bert_oracle.synthetic_batch(b, S, vocab=2000, seed=99, ragged=True)
    want = bert_oracle.encode_cls(model, ids, types, mask)                                    # the reference: transformers fp32
    exact = bert_oracle.encode_cls(copy.deepcopy(model).double(), ids, types, mask)            # ... and what it approximates
    ref_err = (want.double() - exact)
    enc = HipBertEncoder(model, device=cuda_dev).enable_f16x2()
    arith(BF16X3)
    assert not enc.f16x2_active()
    got3 = enc.encode_cls(ids, types, mask).cpu()
    arith(F16X2)
    assert enc.f16x2_active()
    errs, rms, rms64 = {}, {}, {}
    for name, unpad, fusion in (("packed+ln", True, 1), ("packed", True, 0), ("padded+ln", False, 1), ("padded", False, 0)):
        nv.check(nv.lib().ac_gemm_set_ln_fusion(fusion), "ac_gemm_set_ln_fusion")
        enc.unpad = unpad
        got = enc.encode_cls(ids, types, mask).cpu()
        assert torch.isfinite(got).all() and enc.f16x2_overflows == 0
        errs[name] = (got - want).abs().max().item()
        rms[name] = (got - want).pow(2).mean().sqrt().item()
        rms64[name] = (got.double() - exact).pow(2).mean().sqrt().item()
    nv.lib().ac_gemm_set_ln_fusion(1)
    e3, r3 = (got3 - want).abs().max().item(), (got3 - want).pow(2).mean().sqrt().item()
    r3_64, rref = (got3.double() - exact).pow(2).mean().sqrt().item(), ref_err.pow(2).mean().sqrt().item()
    print(f"\n  {arch} peaked={peaked}: vs transformers fp32: bf16x3 max {e3:.2e} rms {r3:.2e} | fp16x2 max "
          + " ".join(f"{k} {v:.2e}" for k, v in errs.items()) + " rms " + " ".join(f"{k} {v:.2e}" for k, v in rms.items())
          + f"\n      rms vs transformers fp64 (the exact forward): transformers fp32 {rref:.2e}  bf16x3 {r3_64:.2e}  fp16x2 "
          + " ".join(f"{k} {v:.2e}" for k, v in rms64.items()))
    assert max(errs.values()) < 1e-4, errs                              # the contract: within 1e-4 of the fp32 reference
    # the honest yardstick is the EXACT forward (fp64): fp16x2 must be as close to it as the fp32 reference itself is, and as
    # bf16x3 is (the peaked model amplifies any rounding ~100x, so all four numbers move together there)
    assert max(rms64.values()) <= 3.0 * max(rref, r3_64) + 1e-8, (rms64, rref, r3_64)


def test_encoder_f16x2_overflow_falls_back_to_bf16x3(cuda_dev, arith):
    """A model whose activations leave +-1023 (a LayerNorm gain of 400 on one channel): under fp16x2 the rows turn NaN, and
    encode_cls repeats the call in bf16x3 by itself -- the caller gets the bf16x3 embeddings, finite and equal."""
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    model = bert_oracle.make_bert(256, 3, 4, 1024, vocab=2000, seed=3)
    with torch.no_grad():
        model.encoder.layer[0].output.LayerNorm.weight[5] = 400.0
    ids, types, mask = bert_oracle.synthetic_batch(32, 32, vocab=2000, seed=5, ragged=True)
    arith(BF16X3)
    enc = HipBertEncoder(model, device=cuda_dev)
    want = enc.encode_cls(ids, types, mask).cpu()
    assert torch.isfinite(want).all()
    arith(F16X2)
    enc.enable_f16x2()
    assert enc.f16x2_active()
    raw = enc.encode_cls(ids, types, mask, verify=False).cpu()          # verify=False: what the kernels produced
    assert torch.isnan(raw).any()
    got = enc.encode_cls(ids, types, mask).cpu()
    assert enc.f16x2_overflows == 1 and not enc.f16x2_active()
    assert torch.equal(got, want)


def test_enable_f16x2_refuses_weights_beyond_the_fp16_range(cuda_dev):
    from adaptive_classifier import _native as nv
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    model = bert_oracle.make_bert(128, 2, 2, 512, vocab=300, seed=4)
    with torch.no_grad():
        model.encoder.layer[1].intermediate.dense.weight[3, 7] = 80.0
    enc = HipBertEncoder(model, device=cuda_dev)
    with pytest.raises(nv.NativeError, match="fp16 range"):
        enc.enable_f16x2()


def test_predict_under_f16x2_agrees_with_bf16x3_and_recovers_from_overflow(cuda_dev, arith):
    """End to end through AdaptiveClassifier: same labels, scores within 1e-5; and with the overflowing model the predict path's
    retry contract (classifier.py::_predict_with_retry) hands back finite scores."""
    from adaptive_classifier import AdaptiveClassifier
    from oracle import bert_oracle
    from adaptive_classifier.encoder import HipBertEncoder

    def build(model, gemm_arith):
        clf = AdaptiveClassifier("synthetic", device=str(cuda_dev), encoder=HipBertEncoder(model, device=cuda_dev),
                                 config={"gemm_arith": gemm_arith})
        rng = np.random.default_rng(0)
        emb = rng.standard_normal((400, model.config.hidden_size)).astype(np.float32)
        emb /= np.linalg.norm(emb, axis=1, keepdims=True)
        clf.add_embeddings([f"t{i}" for i in range(400)], torch.from_numpy(emb), [f"c{i % 4}" for i in range(400)])
        return clf

    model = bert_oracle.make_bert(256, 3, 4, 1024, vocab=2000, seed=3)
    ids, types, mask = bert_oracle.synthetic_batch(64, 32, vocab=2000, seed=5, ragged=True)
    a = build(model, "bf16x3").predict_tokens(ids, types, mask, k=3)
    clf = build(model, "f16x2")
    assert clf._f16x2_active() and not clf.model.f16x2_active()       # per OBJECT: the classifier's calls, not the encoder's default
    bres = clf.predict_tokens(ids, types, mask, k=3)
    assert [[l for l, _ in p] for p in a] == [[l for l, _ in p] for p in bres]
    assert max(abs(x[1] - y[1]) for p, q in zip(a, bres) for x, y in zip(p, q)) < 1e-5
    with torch.no_grad():
        model.encoder.layer[0].output.LayerNorm.weight[5] = 400.0
    clf = build(model, "f16x2")
    res = clf.predict_tokens(ids, types, mask, k=3)
    assert clf.model.f16x2_overflows == 1 and not clf._f16x2_active()
    assert all(s == s for p in res for _, s in p)
