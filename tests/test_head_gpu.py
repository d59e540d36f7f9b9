"""Parity of the HIP head / EWC / AdamW path against the torch-CPU oracle (oracle/head_oracle.py).
Tolerance from BASELINE.json north_star: logits and EWC loss within 1e-4 (fp32)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _data(B, D, C, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1)
    y = torch.randint(0, C, (B,), generator=g)
    return X, y


@pytest.mark.parametrize("M,N,K,act", [(32, 768, 768, 1), (1, 4, 384, 0), (256, 768, 768, 2), (1000, 3072, 768, 2),
                                        (333, 100, 64, 0), (8192, 768, 3072, 0), (7, 5, 12, 1)])
def test_linear_matches_fp64(M, N, K, act, cuda_dev):
    import ctypes
    from adaptive_classifier import _native as nv
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    ref = A.double() @ W.double().T + b.double()
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref)
    ref = ref + R.double()
    Ad, Wd, bd, Rd = (t.to(cuda_dev) for t in (A, W, b, R))
    C = torch.empty(M, N, device=cuda_dev)
    nv.check(nv.lib().ac_linear_f32(nv.ptr(Ad), K, nv.ptr(Wd), K, nv.ptr(bd), nv.ptr(Rd), N, nv.ptr(C), N,
                                    M, N, K, act, nv.stream_ptr(cuda_dev)), "ac_linear_f32")
    err = (C.cpu().double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("tA,tB,M,N,K", [(1, 0, 768, 768, 32), (0, 0, 32, 384, 4), (1, 0, 4, 384, 32), (0, 1, 50, 70, 33),
                                          (1, 1, 40, 30, 20)])
def test_gemm_layouts(tA, tB, M, N, K, cuda_dev):
    from adaptive_classifier import _native as nv
    g = torch.Generator().manual_seed(K)
    A = torch.randn((K, M) if tA else (M, K), generator=g)
    B = torch.randn((N, K) if tB else (K, N), generator=g)
    C0 = torch.randn(M, N, generator=g)
    ref = 0.5 * ((A.T if tA else A).double() @ (B.T if tB else B).double()) + 2.0 * C0.double()
    Ad, Bd, Cd = A.to(cuda_dev), B.to(cuda_dev), C0.to(cuda_dev).clone()
    nv.check(nv.lib().ac_gemm_f32(tA, tB, M, N, K, 0.5, nv.ptr(Ad), A.shape[1], nv.ptr(Bd), B.shape[1], 2.0,
                                  nv.ptr(Cd), N, nv.stream_ptr(cuda_dev)), "ac_gemm_f32")
    assert (Cd.cpu().double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("B,C", [(1, 4), (8, 4), (256, 4), (33, 77), (1024, 64)])
def test_head_forward_logits(B, C, cuda_dev):
    from adaptive_classifier.models import AdaptiveHead
    from oracle import head_oracle
    D = 768
    X, _ = _data(B, D, C)
    ref = head_oracle.make_head(D, C).eval()
    head = AdaptiveHead(D, C, [D, D // 2]).to(cuda_dev).eval()
    with torch.no_grad():
        want = ref(X)
        got = head(X.to(cuda_dev))
    assert got.shape == (B, C)
    assert (got.cpu() - want).abs().max().item() < TOL
    # [D] input keeps the batch dimension (models.py:74-80)
    with torch.no_grad():
        assert head(X[0].to(cuda_dev)).shape == (1, C)


def _make_pair(D, C, cuda_dev):
    from adaptive_classifier.models import AdaptiveHead
    from adaptive_classifier.training import HeadTrainer
    from oracle import head_oracle
    ref = head_oracle.make_head(D, C)
    ref.train()
    opt = torch.optim.AdamW(ref.parameters(), lr=0.001, weight_decay=0.01, betas=(0.9, 0.999))
    head = AdaptiveHead(D, C, [D, D // 2]).to(cuda_dev)
    tr = HeadTrainer(head)
    return ref, opt, head, tr


def test_train_step_with_dropout_masks(cuda_dev):
    """One step of classifier.py:1489-1505: loss, grad-norm, updated params, m, v."""
    from oracle import head_oracle
    D, C, B = 768, 4, 32
    ref, opt, head, tr = _make_pair(D, C, cuda_dev)
    X, y = _data(B, D, C, seed=1)
    g = torch.Generator().manual_seed(7)
    m1 = (torch.rand(B, D, generator=g) >= 0.1)
    m2 = (torch.rand(B, D // 2, generator=g) >= 0.1)
    for step in range(3):
        ce, _, gn = head_oracle.train_step(ref, opt, X, y, masks=[m1, m2])
        loss, out = tr.step(X.to(cuda_dev), y.to(cuda_dev), m1.to(cuda_dev, torch.uint8), m2.to(cuda_dev, torch.uint8))
        assert abs(loss.item() - ce) < TOL
        assert abs(out[1].item() - gn) < TOL * max(1.0, gn)
        assert (tr.flat.cpu() - head_oracle.flat(ref)).abs().max().item() < 5e-5
    st = opt.state_dict()["state"]
    m_ref = torch.cat([st[i]["exp_avg"].reshape(-1) for i in range(6)])
    v_ref = torch.cat([st[i]["exp_avg_sq"].reshape(-1) for i in range(6)])
    assert (tr.m.cpu() - m_ref).abs().max().item() < 1e-6
    assert (tr.v.cpu() - v_ref).abs().max().item() < 1e-8
    # the nn.Parameters are views of the trained flat block
    assert torch.equal(head.model[0].weight.detach().reshape(-1), tr.flat[: D * D])


def test_training_trajectory_no_dropout(cuda_dev):
    from oracle import head_oracle
    D, C, B = 768, 5, 32
    ref, opt, head, tr = _make_pair(D, C, cuda_dev)
    for step in range(25):
        X, y = _data(B, D, C, seed=100 + step)
        ce, _, _ = head_oracle.train_step(ref, opt, X, y, masks=None)
        loss, _ = tr.step(X.to(cuda_dev), y.to(cuda_dev), None, None, 0.0)
        assert abs(loss.item() - ce) < TOL
    head.eval()
    Xt, _ = _data(64, D, C, seed=999)
    with torch.no_grad():
        assert (head(Xt.to(cuda_dev)).cpu() - ref.eval()(Xt)).abs().max().item() < TOL


def test_ewc_fisher_penalty_and_fused_step(cuda_dev):
    """EWC class contract (ewc.py; tests/test_ewc.py:128-153 scenario: params += 0.1)."""
    import ctypes
    from adaptive_classifier import _native as nv
    from adaptive_classifier.ewc import EWC
    from oracle import head_oracle
    D, C, B = 768, 4, 32
    ref, opt, head, tr = _make_pair(D, C, cuda_dev)
    # Fisher from given sampled labels on 3 batches (sizes 32, 32, 6)
    Xs, _ = _data(70, D, C, seed=3)
    batches = [Xs[:32], Xs[32:64], Xs[64:]]
    g = torch.Generator().manual_seed(11)
    sampled = [torch.randint(0, C, (b.shape[0],), generator=g) for b in batches]
    f_ref = head_oracle.fisher_from_labels(ref, batches, sampled)
    ds = torch.utils.data.TensorDataset(Xs, torch.zeros(70, dtype=torch.long))
    ewc = EWC.__new__(EWC)
    ewc.model, ewc.device, ewc.ewc_lambda, ewc._native = head, str(cuda_dev), 5.0, True
    ewc.old_flat = head.flat_params().detach().clone()
    ewc.old_params = {n: p.data.clone() for n, p in head.named_parameters()}
    head.eval()
    ewc.fisher_info = ewc._compute_fisher_native([(b, None) for b in batches],
                                                 sampled_labels=[s.to(cuda_dev) for s in sampled])
    rel = (ewc.fisher_flat.cpu() - f_ref).abs().max().item() / f_ref.abs().max().item()
    assert rel < 1e-4, rel
    assert set(ewc.fisher_info) == {n for n, _ in head.named_parameters()}
    # penalty after p += 0.1
    old_ref = head_oracle.flat(ref).clone()
    with torch.no_grad():
        for p in ref.parameters():
            p += 0.1
        for p in head.parameters():
            p += 0.1
    want = float(head_oracle.ewc_penalty(ref, f_ref, old_ref, 5.0))
    want32 = float(head_oracle.ewc_penalty(ref, f_ref, old_ref, 5.0 / 32))
    with torch.no_grad():
        got, got32 = ewc.ewc_loss().item(), ewc.ewc_loss(batch_size=32).item()
    assert got > 0 and got32 > 0 and got != got32                      # tests/test_ewc.py:147-153
    assert abs(got - want) < TOL * max(1.0, abs(want))
    assert abs(got32 - want32) < TOL * max(1.0, abs(want32))
    # fused EWC + clip + AdamW step == autograd(CE + penalty) + clip_grad_norm_ + AdamW
    head.train(); ref.train()
    X, y = _data(B, D, C, seed=5)
    for step in range(2):
        ce, pen, gn = head_oracle.train_step(ref, opt, X, y, masks=None, fisher_flat=f_ref, old_flat=old_ref,
                                             lam_over_B=5.0 / B)
        loss, out = tr.step(X.to(cuda_dev), y.to(cuda_dev), None, None, 0.0, fisher=ewc.fisher_flat,
                            old_params=ewc.old_flat, lambda_over_B=5.0 / B)
        assert abs(loss.item() - ce) < TOL
        assert abs(out[0].item() - pen) < TOL * max(1.0, abs(pen))
        assert abs(out[1].item() - gn) < TOL * max(1.0, gn)
        assert (tr.flat.cpu() - head_oracle.flat(ref)).abs().max().item() < 5e-5


def _dropout_keep_np(seed, n_rows, n_cols, p):
    """numpy port of ac::dropout_keep (csrc/common.h): keep iff u(seed, row*N+col) >= p."""
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n_rows * n_cols, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u >= np.float32(p)).reshape(n_rows, n_cols)


def test_softmax_and_normalize_rows(cuda_dev):
    from adaptive_classifier.ops import l2_normalize_rows, softmax_rows
    g = torch.Generator().manual_seed(0)
    for B, C in [(1, 4), (37, 77), (256, 1000)]:
        z = torch.randn(B, C, generator=g) * 3
        assert (softmax_rows(z.to(cuda_dev)).cpu() - torch.softmax(z, 1)).abs().max().item() < 1e-6
    x = torch.randn(33, 768, generator=g) * 5
    x[3] = 0                                               # zero row: eps clamp, stays zero
    got = l2_normalize_rows(x.to(cuda_dev)).cpu()
    assert (got - torch.nn.functional.normalize(x, p=2, dim=1)).abs().max().item() < 1e-6


def test_fused_train_step_with_in_kernel_dropout_and_gather(cuda_dev):
    """ac_head_train_step: batch gather + counter-based dropout + CE + backward + clip + AdamW in one call
    equals the oracle step fed with the same (reconstructed) masks and batch."""
    from oracle import head_oracle
    D, C, B, n = 768, 4, 32, 100
    ref, opt, head, tr = _make_pair(D, C, cuda_dev)
    Xall, yall = _data(n, D, C, seed=8)
    Xd, yd = Xall.to(cuda_dev), yall.to(cuda_dev)
    g = torch.Generator().manual_seed(3)
    tr.loss_accum.zero_()
    tot = 0.0
    for step in range(3):
        idx = torch.randperm(n, generator=g)[:B]
        seed = 12345 + step
        m1 = torch.from_numpy(_dropout_keep_np(seed, B, D, 0.1))
        m2 = torch.from_numpy(_dropout_keep_np(seed ^ 0xA5A5A5A5A5A5A5A5, B, D // 2, 0.1))
        ce, _, gn = head_oracle.train_step(ref, opt, Xall[idx], yall[idx], masks=[m1, m2])
        out = tr.fused_step(Xd, yd, idx.to(cuda_dev), 0.1, seed)
        assert abs(out[0].item() - ce) < TOL and abs(out[2].item() - gn) < TOL * max(1.0, gn)
        assert (tr.flat.cpu() - head_oracle.flat(ref)).abs().max().item() < 5e-5
        tot += ce
    assert abs(tr.loss_accum.item() - tot) < 1e-3            # device-side epoch loss accumulation
    # keep-rate of the counter-based masks is ~0.9
    assert abs(_dropout_keep_np(1, 512, 768, 0.1).mean() - 0.9) < 0.005


@pytest.mark.parametrize("with_ewc", [False, True])
def test_fused_epoch_equals_the_step_loop(with_ewc, cuda_dev):
    """ac_head_train_epoch (one call per epoch) is bit-identical to the per-step calls it replaces: same batches
    (consecutive slices of the epoch order, short last batch), seeds seed0 + i, AdamW steps t0 + i, and the EWC
    weight lambda_B / rows_i."""
    D, C, B, n = 768, 4, 32, 77          # 77 = 2 full batches + one of 13
    _, _, head_a, tr_a = _make_pair(D, C, cuda_dev)
    _, _, head_b, tr_b = _make_pair(D, C, cuda_dev)
    assert torch.equal(tr_a.flat, tr_b.flat)
    Xall, yall = _data(n, D, C, seed=21)
    Xd, yd = Xall.to(cuda_dev), yall.to(cuda_dev)
    fisher = old = None
    if with_ewc:
        g = torch.Generator().manual_seed(5)
        fisher = torch.rand(tr_a.flat.numel(), generator=g).to(cuda_dev)
        old = (tr_a.flat + 0.01 * torch.randn(tr_a.flat.numel(), generator=g).to(cuda_dev)).contiguous()
    order = torch.randperm(n, generator=torch.Generator().manual_seed(9)).to(cuda_dev)
    for tr in (tr_a, tr_b):
        tr.loss_accum.zero_()
    for epoch in range(2):
        seed0 = 777 + 10 * epoch
        off = i = 0
        while off < n:
            nb = min(B, n - off)
            tr_a.fused_step(Xd, yd, order[off:off + nb], 0.1, seed0 + i, fisher=fisher, old_params=old,
                            lambda_over_B=(5.0 / nb) if with_ewc else 0.0)
            off += nb
            i += 1
        if epoch == 0:
            done = tr_b.fused_epoch(Xd, yd, order, B, 0.1, seed0, fisher=fisher, old_params=old,
                                    lambda_B=5.0 if with_ewc else 0.0)
        else:      # rows pre-arranged in epoch order, order=None (what the classifier's loop does): same bits
            done = tr_b.fused_epoch(Xd.index_select(0, order), yd.index_select(0, order), None, B, 0.1, seed0,
                                    fisher=fisher, old_params=old, lambda_B=5.0 if with_ewc else 0.0)
        assert done == i == 3 and tr_a.t == tr_b.t
        assert torch.equal(tr_a.flat, tr_b.flat) and torch.equal(tr_a.m, tr_b.m) and torch.equal(tr_a.v, tr_b.v)
        assert tr_a.loss_accum.item() == tr_b.loss_accum.item()
        assert torch.equal(tr_a.out3, tr_b.out3)


@pytest.mark.parametrize("D,C,n,with_ewc", [(768, 4, 109, False), (768, 4, 77, True), (768, 13, 96, False), (128, 3, 70, True)])
def test_persistent_epoch_kernel_agrees_with_the_launch_by_launch_path(D, C, n, with_ewc, cuda_dev):
    """head_epoch.hip (weights + AdamW moments stationary in LDS, one launch per epoch) against the step-by-step kernels of
    head.hip on the same batches, dropout seeds and optimizer state: the two differ only in summation order (fma chains /
    wave-strided dots vs MFMA tiles), so parameters, moments and the epoch loss agree to fp32 round-off -- not bit for bit."""
    from adaptive_classifier import _native as nv
    B = 32
    _, _, head_a, tr_a = _make_pair(D, C, cuda_dev)
    _, _, head_b, tr_b = _make_pair(D, C, cuda_dev)
    Xall, yall = _data(n, D, C, seed=33)
    Xd, yd = Xall.to(cuda_dev), yall.to(cuda_dev)
    fisher = old = None
    if with_ewc:
        g = torch.Generator().manual_seed(6)
        fisher = torch.rand(tr_a.flat.numel(), generator=g).to(cuda_dev)
        old = (tr_a.flat + 0.01 * torch.randn(tr_a.flat.numel(), generator=g).to(cuda_dev)).contiguous()
    order = torch.randperm(n, generator=torch.Generator().manual_seed(10)).to(cuda_dev)
    prev = nv.lib().ac_set_persistent_kernels(-1)
    try:
        for tr, mask in ((tr_a, prev | 1), (tr_b, prev & ~1)):
            nv.lib().ac_set_persistent_kernels(mask)
            tr.loss_accum.zero_()
            for epoch in range(1):      # (a few steps: AdamW turns round-off on near-zero gradients into O(lr) differences over time)
                tr.fused_epoch(Xd, yd, order, B, 0.1, 500 + epoch, fisher=fisher, old_params=old, lambda_B=5.0 if with_ewc else 0.0)
            torch.cuda.synchronize()
    finally:
        nv.lib().ac_set_persistent_kernels(prev)
    assert tr_a.t == tr_b.t
    assert (tr_a.flat - tr_b.flat).abs().max().item() < 5e-5
    assert (tr_a.m - tr_b.m).abs().max().item() < 1e-5 and (tr_a.v - tr_b.v).abs().max().item() < 1e-6
    assert abs(tr_a.loss_accum.item() - tr_b.loss_accum.item()) < 1e-3 * max(1.0, abs(tr_b.loss_accum.item()))
    assert (tr_a.out3 - tr_b.out3).abs().max().item() < 1e-4
    assert (tr_a.grads - tr_b.grads).abs().max().item() < 1e-5          # raw gradients of the last step


def test_linear_randomised_shapes(cuda_dev):
    """Random (M, N, K, act, residual) through ac_linear_f32: exercises the small-M weight-streaming kernel,
    the direct kernel and both LDS tile heights, with ragged edges."""
    from adaptive_classifier import _native as nv
    rng = np.random.default_rng(7)
    for case in range(36):
        M = int(rng.choice([1, 3, 16, 17, 33, 64, 65, 150, 192, 200, 257, 640, 1000]))
        N = int(rng.choice([4, 16, 20, 100, 128, 130, 384, 768, 1000]))
        K = int(rng.choice([4, 8, 12, 32, 36, 64, 96, 100, 384, 768]))
        act = int(rng.integers(0, 3))
        use_res = bool(rng.integers(0, 2))
        g = torch.Generator().manual_seed(case)
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        R = torch.randn(M, N, generator=g)
        ref = A.double() @ W.double().T + b.double()
        ref = torch.relu(ref) if act == 1 else torch.nn.functional.gelu(ref) if act == 2 else ref
        if use_res:
            ref = ref + R.double()
        Ad, Wd, bd, Rd = (t.to(cuda_dev) for t in (A, W, b, R))
        C = torch.full((M, N), float("nan"), device=cuda_dev)
        nv.check(nv.lib().ac_linear_f32(nv.ptr(Ad), K, nv.ptr(Wd), K, nv.ptr(bd), nv.ptr(Rd) if use_res else None, N,
                                        nv.ptr(C), N, M, N, K, act, nv.stream_ptr(cuda_dev)), "ac_linear_f32")
        err = (C.cpu().double() - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (M, N, K, act, use_res, err)
