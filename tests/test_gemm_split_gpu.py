"""The bf16x3 split-operand GEMM (AC_GEMM_BF16X3: six bf16 MFMA products per tile, fp32 accumulate) must be
fp32-GRADE: its error against an fp64 product is bounded by a small multiple of the error the fp32-input
MFMA kernel (an exact fma chain) makes on the same operands, and by the a-priori fp32 dot-product bound
K * 2^-24 * sum|a||b|.  Tolerances are written out below."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F32, BF16X3 = 0, 1


@pytest.fixture()
def arith():
    from adaptive_classifier import _native as nv
    lib = nv.lib()
    before = lib.ac_gemm_get_arith()
    yield lambda mode: nv.check(lib.ac_gemm_set_arith(mode), "ac_gemm_set_arith")
    lib.ac_gemm_set_arith(before)
    lib.ac_gemm_set_krot(0)


def _planes(nv, dev, Xd):
    rows, K = Xd.shape
    P = torch.empty(3 * rows * K, dtype=torch.int16, device=dev)
    nv.check(nv.lib().ac_split_bf16x3(nv.ptr(Xd), K, rows, K, nv.ptr(P), nv.stream_ptr(dev)), "ac_split_bf16x3")
    return P


def _linear(nv, dev, A, W, b, R, act, planes=""):
    """planes: "" -> ac_linear_f32; "w" -> W pre-split; "aw" -> both operands pre-split."""
    M, K = A.shape
    N = W.shape[0]
    Ad, Wd, bd = (torch.from_numpy(x).to(dev) for x in (A, W, b))
    Rd = torch.from_numpy(R).to(dev) if R is not None else None
    C = torch.empty((M, N), device=dev)
    if not planes:
        nv.check(nv.lib().ac_linear_f32(nv.ptr(Ad), K, nv.ptr(Wd), K, nv.ptr(bd), nv.ptr(Rd) if R is not None else None,
                                        N, nv.ptr(C), N, M, N, K, act, nv.stream_ptr(dev)), "ac_linear_f32")
    else:
        Wp = _planes(nv, dev, Wd)
        Ap = _planes(nv, dev, Ad) if "a" in planes else None
        nv.check(nv.lib().ac_linear_bf16x3(nv.ptr(Ad), K, nv.ptr(Ap) if Ap is not None else None, nv.ptr(Wd), K,
                                           nv.ptr(Wp), nv.ptr(bd), nv.ptr(Rd) if R is not None else None, N,
                                           nv.ptr(C), N, None, M, N, K, act, nv.stream_ptr(dev)), "ac_linear_bf16x3")
    return C.cpu().numpy().astype(np.float64)


def _ref(A, W, b, R, act):
    z = A.astype(np.float64) @ W.astype(np.float64).T + b
    if act == 1:
        z = np.maximum(z, 0)
    elif act == 2:
        from scipy.special import erf
        z = 0.5 * z * (1 + erf(z / np.sqrt(2)))
    if R is not None:
        z = z + R
    return z


@pytest.mark.parametrize("M,N,K,act,res", [
    (8192, 768, 768, 0, True),      # attention output projection at BASELINE configs[1] (256 x 32 tokens)
    (8192, 2304, 768, 0, False),    # fused QKV
    (1024, 3072, 768, 2, False),    # FFN up + GELU
    (1024, 768, 3072, 0, True),     # FFN down + residual
    (200, 130, 96, 1, False),       # ragged edges: M, N not tile multiples, 64-row tile
    (333, 257, 32, 0, True),        # single k-tile
    (8192, 3072, 128, 2, False),    # >= 1.5 rounds of 256 x 128 tiles: the 8-wave tile is auto-selected ("aw" mode)
    (8200, 3072, 64, 0, True),      # ... with a ragged last row tile and a residual
])
def test_split_gemm_is_fp32_grade(M, N, K, act, res, cuda_dev, arith):
    from adaptive_classifier import _native as nv
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if res else None
    want = _ref(A, W, b, R, act)
    arith(F32)
    e32 = np.abs(_linear(nv, cuda_dev, A, W, b, R, act) - want)
    arith(BF16X3)
    bound = K * 2.0 ** -24 * (np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T) + 1e-6
    for planes in ("", "w", "aw"):        # split in-kernel / weights pre-split / both operands pre-split
        got = _linear(nv, cuda_dev, A, W, b, R, act, planes)
        es = np.abs(got - want)
        assert np.all(es <= bound), (planes, es.max(), bound.min())          # a-priori fp32 dot-product bound
        assert es.max() <= 3.0 * e32.max() + 1e-7, (planes, es.max(), e32.max())
        assert es.mean() <= 3.0 * e32.mean() + 1e-9, (planes, es.mean(), e32.mean())


def test_split_planes_reconstruct_exactly(cuda_dev):
    """h + m + l == x bit for bit (fp32 has 24 significand bits, the three bf16 terms carry 8 each plus signs),
    and the k-slot-major layout is planes[p][k // 8][row][k % 8]."""
    from adaptive_classifier import _native as nv
    rng = np.random.default_rng(3)
    rows, K = 77, 40
    X = (rng.standard_normal((rows, K)) * 10.0 ** rng.uniform(-8, 8, (rows, K))).astype(np.float32)
    X[0, :4] = [0.0, -0.0, 1.0, -1.5]
    P = _planes(nv, cuda_dev, torch.from_numpy(X).to(cuda_dev)).cpu().numpy().view(np.uint16)
    P = P.reshape(3, K // 8, rows, 8).transpose(0, 2, 1, 3).reshape(3, rows, K)
    f = (P.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    assert np.array_equal((f[0] + f[1] + f[2]).astype(np.float32), X)
    assert np.all(np.abs(f[1]) <= np.abs(f[0]) * 2.0 ** -8 + 1e-45) and np.all(np.abs(f[2]) <= np.abs(f[0]) * 2.0 ** -16 + 1e-45)


def test_split_gemm_wide_dynamic_range_and_asymmetric_layout(cuda_dev, arith):
    """Operands spanning 12 orders of magnitude (every plane of the split matters) and an asymmetric W so
    a transposed / permuted fragment mapping cannot pass."""
    from adaptive_classifier import _native as nv
    rng = np.random.default_rng(7)
    M, N, K = 256, 256, 128
    A = (rng.standard_normal((M, K)) * 10.0 ** rng.uniform(-6, 6, (M, K))).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 10.0 ** rng.uniform(-6, 6, (N, K))).astype(np.float32)
    W[:, ::3] *= -3.0
    b = np.zeros(N, np.float32)
    want = _ref(A, W, b, None, 0)
    arith(BF16X3)
    scale = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T
    eye = np.eye(256, 128, dtype=np.float32)
    for planes in ("", "w", "aw"):
        got = _linear(nv, cuda_dev, A, W, b, None, 0, planes)
        assert np.max(np.abs(got - want) / scale) < 2.0 ** -20, planes      # fp32 fma chain bound: K * 2^-24 = 2^-17
        # identity check: A = I picks out W^T exactly (h + m + l == x, products with 1.0 exact)
        got = _linear(nv, cuda_dev, eye, W, b, None, 0, planes)
        assert np.array_equal(got[:128], W.T.astype(np.float64)) and not got[128:].any(), planes


def test_encoder_parity_holds_under_split_arithmetic(cuda_dev, arith):
    """SURVEY 8c tolerance (1e-4 max-abs on the unit-norm CLS embedding) with the split GEMMs, and the
    distance to the fp32-MFMA embedding is at fp32 rounding level."""
    from adaptive_classifier.encoder import HipBertEncoder
    from oracle import bert_oracle
    model = bert_oracle.make_bert(768, 12, 12, 3072, vocab=2000, seed=0)
    ids, types, mask = bert_oracle.synthetic_batch(16, 32, vocab=2000, seed=99, ragged=True)
    want = bert_oracle.encode_cls(model, ids, types, mask)
    enc = HipBertEncoder(model, device=cuda_dev)
    arith(F32)
    g32 = enc.encode_cls(ids, types, mask).cpu()
    arith(BF16X3)
    gs = enc.encode_cls(ids, types, mask).cpu()
    e32, es = (g32 - want).abs().max().item(), (gs - want).abs().max().item()
    assert es < 1e-4, es
    assert es <= 3 * e32 + 1e-6, (es, e32)
    assert (gs - g32).abs().max().item() < 5e-6
    # the same arithmetic without pre-split planes (operands split inside the GEMM tiles): the planes path
    # (LayerNorm / attention / GELU producers emitting operand planes) is a layout change, not a numeric one
    saved = [getattr(enc.weights, k) for k in ("qkv_w3", "ao_w3", "ff1_w3", "ff2_w3")]
    for k in ("qkv_w3", "ao_w3", "ff1_w3", "ff2_w3"):
        setattr(enc.weights, k, None)
    gi = enc.encode_cls(ids, types, mask).cpu()
    for k, v in zip(("qkv_w3", "ao_w3", "ff1_w3", "ff2_w3"), saved):
        setattr(enc.weights, k, v)
    assert (gi - want).abs().max().item() < 1e-4
    assert (gi - gs).abs().max().item() < 5e-6


@pytest.mark.parametrize("M,N,K,act", [(8192, 3072, 768, 2), (200, 136, 96, 0), (1024, 768, 64, 2)])
def test_planes_output_equals_split_of_fp32_output(M, N, K, act, cuda_dev, arith):
    """d_C_planes: the GEMM emits its result directly as the next GEMM's operand planes.  They must be
    bit-identical to ac_split_bf16x3 of the fp32 result of the same kernel."""
    from adaptive_classifier import _native as nv
    rng = np.random.default_rng(N)
    arith(BF16X3)
    nv.lib().ac_gemm_set_krot(0)      # (the fp32-out GELU form runs on the two-buffer kernel: compare in-order sums with in-order sums)
    Ad = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(cuda_dev)
    Wd = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(cuda_dev)
    bd = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(cuda_dev)
    Ap, Wp = _planes(nv, cuda_dev, Ad), _planes(nv, cuda_dev, Wd)
    C = torch.empty((M, N), device=cuda_dev)
    Cp = torch.zeros(3 * M * N, dtype=torch.int16, device=cuda_dev)
    lib, st = nv.lib(), nv.stream_ptr(cuda_dev)
    nv.check(lib.ac_linear_bf16x3(nv.ptr(Ad), K, nv.ptr(Ap), nv.ptr(Wd), K, nv.ptr(Wp), nv.ptr(bd), None, 0,
                                  nv.ptr(C), N, None, M, N, K, act, st), "fp32 out")
    nv.check(lib.ac_linear_bf16x3(nv.ptr(Ad), K, nv.ptr(Ap), nv.ptr(Wd), K, nv.ptr(Wp), nv.ptr(bd), None, 0,
                                  None, N, nv.ptr(Cp), M, N, K, act, st), "planes out")
    assert torch.equal(Cp, _planes(nv, cuda_dev, C))
    # a shape that does not take the pre-split kernel must refuse planes output loudly
    rc = lib.ac_linear_bf16x3(nv.ptr(Ad), K, nv.ptr(Ap), nv.ptr(Wd), K, nv.ptr(Wp), nv.ptr(bd), None, 0,
                              None, N, nv.ptr(Cp), 16, N, K, act, st)
    assert rc != 0


def test_fused_geglu_epilogue(cuda_dev, arith):
    """act = 3: GeGLU over 32-column blocks fused into the planes-output epilogue equals gelu(in) * gate computed
    from the fp32 GEMM output of the same kernel, split into planes."""
    from adaptive_classifier import _native as nv
    arith(BF16X3)
    rng = np.random.default_rng(11)
    M, I, K = 520, 192, 128
    N = 2 * I
    Ad = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(cuda_dev)
    Wd = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(cuda_dev)
    bd = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(cuda_dev)
    Ap, Wp = _planes(nv, cuda_dev, Ad), _planes(nv, cuda_dev, Wd)
    U = torch.empty((M, N), device=cuda_dev)
    Gp = torch.zeros(3 * M * I, dtype=torch.int16, device=cuda_dev)
    lib, st = nv.lib(), nv.stream_ptr(cuda_dev)
    nv.check(lib.ac_linear_bf16x3(nv.ptr(Ad), K, nv.ptr(Ap), nv.ptr(Wd), K, nv.ptr(Wp), nv.ptr(bd), None, 0,
                                  nv.ptr(U), N, None, M, N, K, 0, st), "fp32 out")
    nv.check(lib.ac_linear_bf16x3(nv.ptr(Ad), K, nv.ptr(Ap), nv.ptr(Wd), K, nv.ptr(Wp), nv.ptr(bd), None, 0,
                                  None, I, nv.ptr(Gp), M, N, K, 3, st), "geglu planes out")
    u = U.view(M, I // 32, 2, 32)
    x, gate = u[:, :, 0, :], u[:, :, 1, :]
    want = (0.5 * x * (1 + torch.erf(x * 0.70710678118654752440)) * gate).reshape(M, I).contiguous()
    got = Gp.cpu().numpy().view(np.uint16).reshape(3, I // 8, M, 8).transpose(0, 2, 1, 3).reshape(3, M, I)
    got = (got.astype(np.uint32) << 16).view(np.float32).astype(np.float64).sum(0)
    assert np.abs(got - want.cpu().numpy().astype(np.float64)).max() < 2e-6       # erff vs torch.erf, 1-2 ulp
    # without planes output the fused activation is refused
    assert lib.ac_linear_bf16x3(nv.ptr(Ad), K, nv.ptr(Ap), nv.ptr(Wd), K, nv.ptr(Wp), nv.ptr(bd), None, 0,
                                nv.ptr(U), N, None, M, N, K, 3, st) != 0


# every ring configuration gemm_pipe.hip builds (AC_PIPE_CONFIGS): tm tn wmw wnw ring-depth pipelining, one digit each
PIPE_CFGS = [222232, 124261, 124262, 224242, 234232, 322432, 244232]


@pytest.mark.parametrize("M,N,K,act,res", [
    (5141, 768, 768, 0, True),      # the attention-output GEMM of the timed batch (ragged last row tile)
    (1000, 2304, 64, 0, False),     # 4 k-stages: shorter than the deepest ring (over-issued tail stages)
    (261, 200, 96, 2, False),       # ragged rows AND columns, 6 stages
    (4096, 3072, 1024, 0, True),    # bert-large width
])
def test_ring_staged_kernels_equal_the_two_buffer_kernels_bit_for_bit(M, N, K, act, res, cuda_dev, arith):
    """gemm_pipe.hip (operand stages in an LDS ring, counted vmcnt, raw barriers, software-pipelined fragment reads): every
    built configuration computes the SAME six products in the same order per output element as gemm_planes_nt, so the results
    are bit-identical to the two-buffer kernels (variant 1) -- a stage read before its DMA landed, or overwritten before its
    last read, would show up here as differing elements -- and fp32-grade against fp64."""
    from adaptive_classifier import _native as nv
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if res else None
    want = _ref(A, W, b, R, act)
    arith(BF16X3)
    lib = nv.lib()
    lib.ac_gemm_set_krot(0)          # (the default) every workgroup walks k in order: the same sums in the same order whatever the tile
    try:
        nv.check(lib.ac_gemm_set_variant(1), "ac_gemm_set_variant")
        base = _linear(nv, cuda_dev, A, W, b, R, act, "aw")
        bound = K * 2.0 ** -24 * (np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T) + 1e-6
        assert np.all(np.abs(base - want) <= bound)
        for cfg in PIPE_CFGS:
            nv.check(lib.ac_gemm_set_variant(cfg), "ac_gemm_set_variant")
            for rep in range(2):                                # (twice: a race would not repeat itself)
                got = _linear(nv, cuda_dev, A, W, b, R, act, "aw")
                assert np.array_equal(got, base), (cfg, rep, int((got != base).sum()), float(np.abs(got - base).max()))
        nv.check(lib.ac_gemm_set_variant(0), "ac_gemm_set_variant")      # the default dispatch (per-shape choice)
        assert np.array_equal(_linear(nv, cuda_dev, A, W, b, R, act, "aw"), base)
    finally:
        lib.ac_gemm_set_variant(0)
        lib.ac_gemm_set_krot(0)


@pytest.mark.parametrize("M,N,K,res", [
    (5141, 768, 768, True),         # one round of 128 x 128 tiles: every XCD starts at a different stage
    (1000, 2304, 64, False),        # 4 k-stages: fewer than XCDs (start stages 0, 0, 0, 0, 2, 2, 2, 2)
    (261, 200, 96, False),          # ragged, 6 stages
    (5141, 768, 3072, True),        # the long k-loop
])
def test_rotated_k_order_sums_every_stage_exactly_once(M, N, K, res, cuda_dev, arith):
    """ac_gemm_set_krot(1) (experiment switch, default off): workgroups on XCD x walk the k stages x nk / 8 ... nk - 1, 0 ... x nk / 8 - 1.  With small
    INTEGER operands every partial sum is exact in fp32 whatever the order, so each configuration must return the exact integer
    product bit for bit -- a stage skipped, repeated or read from the wrong ring slot after the wrap cannot pass.  On random
    operands the rotated result stays within fp32 rounding of the in-order one (and inside the a-priori bound)."""
    from adaptive_classifier import _native as nv
    rng = np.random.default_rng(M + N + K)
    A = rng.integers(-3, 4, (M, K)).astype(np.float32)
    W = rng.integers(-3, 4, (N, K)).astype(np.float32)
    b = rng.integers(-5, 6, N).astype(np.float32)
    R = rng.integers(-9, 10, (M, N)).astype(np.float32) if res else None
    want = _ref(A, W, b, R, 0)
    Ar = rng.standard_normal((M, K)).astype(np.float32)
    Wr = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    wantr = _ref(Ar, Wr, b, R, 0)
    bound = K * 2.0 ** -24 * (np.abs(Ar).astype(np.float64) @ np.abs(Wr).astype(np.float64).T) + 1e-6
    arith(BF16X3)
    lib = nv.lib()
    try:
        lib.ac_gemm_set_krot(0)
        nv.check(lib.ac_gemm_set_variant(0), "ac_gemm_set_variant")
        inorder = _linear(nv, cuda_dev, Ar, Wr, b, R, 0, "aw")
        lib.ac_gemm_set_krot(1)
        for cfg in [0] + PIPE_CFGS:
            nv.check(lib.ac_gemm_set_variant(cfg), "ac_gemm_set_variant")
            for rep in range(2):
                got = _linear(nv, cuda_dev, A, W, b, R, 0, "aw")
                assert np.array_equal(got, want), (cfg, rep, int((got != want).sum()))
            got = _linear(nv, cuda_dev, Ar, Wr, b, R, 0, "aw")
            assert np.all(np.abs(got - wantr) <= bound), cfg
            assert np.abs(got - inorder).max() <= 64 * 2.0 ** -24 * np.abs(wantr).max() + 1e-6, cfg
    finally:
        lib.ac_gemm_set_variant(0)
        lib.ac_gemm_set_krot(0)


@pytest.mark.parametrize("M,N,K,act,res", [
    (256, 768, 768, 1, False),      # the head's first layer over a predict batch (192 tiles of 32 x 32)
    (256, 384, 768, 1, False),      # its second layer
    (256, 4, 384, 0, False),        # its output layer: one column tile, 28 of 32 columns beyond N
    (256, 3072, 768, 2, False),     # the last encoder layer's FFN over the CLS rows
    (256, 768, 3072, 0, True),
    (65, 33, 72, 0, True),          # ragged rows and columns, K = 4.5 slots of 16 (the tail slot is half zeros)
    (500, 40, 136, 1, True),        # an odd number of 128-column rounds
])
def test_few_tile_kernel_is_an_exact_fp32_product(M, N, K, act, res, cuda_dev, arith):
    """gemm_fewtiles_nt (gemm.hip): fp32 MFMA products, fp32 accumulation over 8 interleaved K-slices summed in a fixed order.
    Bars: the a-priori fp32 dot-product bound K * 2^-24 * sum|a||b| (+ one rounding of the epilogue), whatever arithmetic the
    process is set to (the kernel takes these shapes under both), and the launch must be THIS kernel: same result as with the
    kernel switched off only to fp32 roundoff, not bit for bit (different summation order) -- the bound is the test."""
    from adaptive_classifier import _native as nv
    rng = np.random.default_rng(7 * M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if res else None
    want = _ref(A, W, b, R, act)
    bound = K * 2.0 ** -24 * (np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T) + 2.0 ** -23 * np.abs(want) + 1e-6
    for mode in (F32, BF16X3):
        arith(mode)
        got = _linear(nv, cuda_dev, A, W, b, R, act)
        assert np.isfinite(got).all()
        es = np.abs(got - want)
        assert np.all(es <= bound), (mode, es.max(), bound.min())
