"""Parity of the HIP kNN path (ac_knn_l2_topk through the C ABI) against the oracle.

Bar (BASELINE.json north_star): identical top-k ids; distances are the exact squared distance
rounded to fp32 (tolerance: 1 ulp, because the fp64 summation order differs from numpy's).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ulp_close(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    return np.all(np.abs(a - b) <= np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)))


def _check(P, Q, k, cuda_dev, row_offset=0):
    from adaptive_classifier import index as ix
    from oracle import c_oracle
    Pd = torch.from_numpy(P).to(cuda_dev)
    ld = (P.shape[1] + 3) // 4 * 4
    store = torch.zeros((max(P.shape[0], 1), ld), dtype=torch.float32, device=cuda_dev)
    if P.shape[0]:
        store[: P.shape[0], : P.shape[1]] = Pd
    Qd = torch.from_numpy(Q).to(cuda_dev)
    stats = torch.zeros(4, dtype=torch.int32, device=cuda_dev)
    D, I = ix.knn_l2_topk(store, P.shape[0], P.shape[1], Qd, k, row_offset=row_offset, stats=stats)
    torch.cuda.synchronize()
    oD, oI = c_oracle.knn_l2_topk(P, Q, k, row_offset) if P.shape[0] else (
        np.full((Q.shape[0], k), np.finfo(np.float32).max, np.float32), np.full((Q.shape[0], k), -1, np.int64))
    I = I.cpu().numpy()
    D = D.cpu().numpy()
    assert np.array_equal(I, oI), f"id mismatch: {(I != oI).sum()} of {I.size}"
    assert _ulp_close(D, oD)
    return int(stats[0].item())


@pytest.mark.parametrize("N,D,nq,k", [
    (4, 768, 1, 4),          # the reference's own shape: one prototype per class, k = #classes
    (100, 768, 8, 5),
    (77, 768, 3, 77),        # k = N (banking77-like)
    (1000, 128, 16, 16),     # bert-tiny dim
    (5000, 768, 33, 16),     # 2 query tiles, ragged
    (4097, 1024, 17, 32),    # e5-large dim -> TQ=16 variant by LDS budget
    (3000, 384, 64, 1),
    (2500, 100, 5, 10),      # D % 8 != 0 (tail group), D % 4 == 0
    (999, 770, 7, 9),        # D % 4 != 0 (zero padded ld)
    (20000, 768, 256, 16),
    (300, 64, 40, 200),      # large k -> TQ=16, cap 512
    (300, 64, 40, 300),      # k > 248 = #classes-style search: small-store exact path
    (1000, 768, 5, 1000),    # k = N, full ordering of 1000 prototypes
    (500, 4096, 3, 10),      # D too wide for the LDS query tile: small-store exact path
    (10, 768, 2, 300),       # small-store path with k > N (padding)
])
def test_knn_matches_oracle(N, D, nq, k, cuda_dev):
    from oracle import synth
    P = synth.synth_unit_rows(N, D, seed=1)
    Q = synth.synth_unit_rows(nq, D, seed=2)
    _check(P, Q, k, cuda_dev)


def test_knn_row_offset_and_unnormalised(cuda_dev):
    rng = np.random.default_rng(0)
    P = (rng.standard_normal((3000, 768)) * 3).astype(np.float32)
    Q = (rng.standard_normal((9, 768)) * 0.5).astype(np.float32)
    _check(P, Q, 8, cuda_dev, row_offset=10_000_000_000)


def test_knn_duplicates_and_ties(cuda_dev):
    """Exact duplicates straddling the k boundary: ids must come back lowest-first (faiss tie rule).
    The certificate cannot separate exact ties, so these queries take the exact fallback."""
    from oracle import synth
    base = synth.synth_unit_rows(50, 768, seed=5)
    P = np.concatenate([base] * 40, axis=0)           # every row appears 40 times
    Q = base[:6].copy()
    nfb = _check(P, Q, 16, cuda_dev)
    assert nfb == 6


def test_knn_k_greater_than_N_pads(cuda_dev):
    from oracle import synth
    P = synth.synth_unit_rows(5, 768, seed=1)
    Q = synth.synth_unit_rows(3, 768, seed=2)
    _check(P, Q, 12, cuda_dev)


def test_knn_empty_index(cuda_dev):
    Q = np.ones((2, 768), np.float32)
    _check(np.zeros((0, 768), np.float32), Q, 4, cuda_dev)


def test_knn_adversarial_order(cuda_dev):
    """Rows sorted so that every later tile beats all earlier ones (worst case for the running
    threshold: every row is pushed, lists overflow and are pruned every tile)."""
    from oracle import synth
    P = synth.synth_unit_rows(6000, 768, seed=7)
    q = synth.synth_unit_rows(1, 768, seed=8)
    d = ((P - q) ** 2).sum(1)
    P = P[np.argsort(-d)]                              # farthest first
    Q = np.repeat(q, 20, axis=0) + synth.synth_unit_rows(20, 768, seed=9) * 1e-3
    _check(P, Q.astype(np.float32), 32, cuda_dev)


def test_synth_rows_bit_identical(cuda_dev):
    from adaptive_classifier import index as ix
    from oracle import synth
    for (n, D, seed, off) in [(100, 768, 1, 0), (37, 1024, 3, 12345), (5, 770, 9, 1 << 33)]:
        dev = ix.synth_unit_rows(n, D, seed, off, device=cuda_dev)
        ref = synth.synth_unit_rows(n, D, seed, off)
        assert np.array_equal(dev[:, :D].cpu().numpy().view(np.uint32), ref.view(np.uint32))
        assert float(dev[:, D:].abs().sum()) == 0.0


def test_large_sweep_properties(cuda_dev):
    """1M x 768 (3 GB): oracle on an 8-query subset, plus size-independent properties:
    ascending distances, unique ids, and shard-consistency (top-k of the union of two halves ==
    merge of the halves' top-k)."""
    from adaptive_classifier import index as ix
    from oracle import c_oracle
    N, D, nq, k = 1_000_000, 768, 32, 32
    P = ix.synth_unit_rows(N, D, 1, device=cuda_dev)
    Q = ix.synth_unit_rows(nq, D, 2, device=cuda_dev)
    Dd, Id = ix.knn_l2_topk(P, N, D, Q, k)
    h = N // 2
    D0, I0 = ix.knn_l2_topk_exact(P[:h], h, D, Q, k)                # shards contribute exact fp64 distances
    D1, I1 = ix.knn_l2_topk_exact(P[h:], N - h, D, Q, k, row_offset=h)
    Dm, Im = ix.topk_merge(torch.stack([D0, D1]), torch.stack([I0, I1]))
    torch.cuda.synchronize()
    assert torch.equal(Im, Id) and torch.equal(Dm, Dd)
    d = Dd.cpu().numpy()
    assert np.all(np.diff(d, axis=1) >= 0)
    i = Id.cpu().numpy()
    assert all(len(set(r)) == k for r in i)
    sub = 8
    oD, oI = c_oracle.knn_l2_topk(P.cpu().numpy(), Q[:sub].cpu().numpy(), k)
    assert np.array_equal(i[:sub], oI)
    assert _ulp_close(d[:sub], oD)


def test_topk_merge_and_scores(cuda_dev):
    from adaptive_classifier import index as ix
    from oracle import knn_oracle
    rng = np.random.default_rng(3)
    S, nq, k = 8, 13, 32
    Din = np.sort(rng.random((S, nq, k)).astype(np.float32), axis=2)
    Iin = rng.permutation(S * nq * k).reshape(S, nq, k).astype(np.int64)
    Din[2, :, 20:] = np.finfo(np.float32).max
    Iin[2, :, 20:] = -1
    Din[3, 0, 0] = Din[4, 0, 0]                       # a tie across shards
    D, I = ix.topk_merge(torch.from_numpy(Din).to(cuda_dev), torch.from_numpy(Iin).to(cuda_dev))
    oD, oI = knn_oracle.topk_merge(Din, Iin, k)
    assert np.array_equal(I.cpu().numpy(), oI) and np.array_equal(D.cpu().numpy(), oD)
    sc = ix.proto_scores(D, I).cpu().numpy()
    ref = knn_oracle.proto_scores(oD, oI)
    assert np.allclose(sc, ref, atol=1e-6)            # fp32 exp/softmax: 1e-6 absolute
    assert np.allclose(sc.sum(1), 1.0, atol=1e-5)     # tests/test_memory.py:84-85


def test_knn_many_fallbacks_slots_and_direct(cuda_dev):
    """More flagged queries (100) than fallback slots (64): slab-parallel path and single-block path agree
    with the oracle; ties resolved lowest-id first."""
    from oracle import synth
    base = synth.synth_unit_rows(100, 256, seed=11)
    P = np.concatenate([base] * 30, axis=0)          # 30 exact copies: the tie group straddles k' = k + 8
    Q = base[:100].copy()
    nfb = _check(P, Q, 16, cuda_dev)
    assert nfb == 100


def test_knn_randomised_shapes(cuda_dev):
    """40 seeded random (N, D, nq, k) combinations incl. odd sizes, k near the limits, scaled / shifted data
    and sprinkled duplicate rows: ids bit-exact, distances to 1 ulp."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        N = int(rng.choice([1, 2, 7, 63, 64, 65, 127, 129, 500, 1023, 2049, 6000, 12345]))
        D = int(rng.choice([4, 12, 60, 64, 100, 128, 384, 768, 772, 1024, 1536]))
        nq = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 70]))
        k = int(rng.choice([1, 2, 5, 16, 32, 57, 100, 240, 248]))
        scale = float(rng.choice([1.0, 1e-3, 50.0]))
        P = (rng.standard_normal((N, D)) * scale + float(rng.choice([0.0, 3.0]))).astype(np.float32)
        if N > 10 and rng.random() < 0.5:
            P[rng.integers(0, N, size=N // 5)] = P[rng.integers(0, N)]      # duplicate rows
        Q = (rng.standard_normal((nq, D)) * scale).astype(np.float32)
        if rng.random() < 0.3:
            Q[0] = P[0]
        _check(P, Q, k, cuda_dev, row_offset=int(rng.choice([0, 7, 1 << 40])))


def test_index_remove_ids_compacts_like_faiss(cuda_dev):
    """HipFlatL2Index.remove_ids (the faiss protocol method memory.py:156-159 calls before re-adding a prototype):
    rows compact (later rows shift down), ntotal / return value follow faiss, and searches after remove + add see
    exactly the store the oracle shim holds after the same operations."""
    from adaptive_classifier.index import HipFlatL2Index
    from oracle import faiss_shim, synth
    D = 96
    X = synth.synth_unit_rows(300, D, 21)
    Q = synth.synth_unit_rows(7, D, 22)
    idx, ref = HipFlatL2Index(D, device=cuda_dev), faiss_shim.IndexFlatL2(D)
    for blk in (X[:100], X[100:250]):
        idx.add(blk)
        ref.add(blk)
    assert idx.remove_ids(torch.tensor([5])) == ref.remove_ids(np.array([5])) == 1        # the call shape memory.py:158 uses
    assert idx.ntotal == ref.ntotal == 249
    ids = np.array([0, 17, 17, 248, 400, -3, 100])                                         # duplicates / out of range ignored
    assert idx.remove_ids(ids) == ref.remove_ids(ids) == 4
    assert idx.ntotal == ref.ntotal == 245
    idx.add(X[250:])                                                                       # re-add after compaction
    ref.add(X[250:])
    d, i = idx.search(Q, 12)
    rd, ri = ref.search(Q, 12)
    assert np.array_equal(i, ri) and _ulp_close(d, rd)
    assert np.array_equal(idx._store[: idx.ntotal, :D].cpu().numpy(), ref._x)              # same rows, same order
    # removing everything leaves an empty, still usable index
    assert idx.remove_ids(np.arange(idx.ntotal)) == 295 and idx.ntotal == 0
    idx.add(X[:3])
    d, i = idx.search(Q[:1], 3)
    assert sorted(i[0].tolist()) == [0, 1, 2]
    # pending (not yet uploaded) rows are removable too
    idx2 = HipFlatL2Index(D, device=cuda_dev)
    idx2.add(X[:10])
    assert idx2.remove_ids([2, 3]) == 2 and idx2.ntotal == 8
    d2, i2 = idx2.search(Q, 4)
    ref2 = faiss_shim.IndexFlatL2(D); ref2.add(X[:10]); ref2.remove_ids([2, 3])
    assert np.array_equal(i2, ref2.search(Q, 4)[1])


@pytest.mark.parametrize("N,D,nq,k,ring", [
    (200_000, 768, 16, 10, True),     # the roofline configuration's shape class: 16 resident queries
    (70_001, 768, 1, 16, True),       # single query, ragged last tile (rows clamped)
    (50_000, 384, 7, 32, True),       # 12 chunks per row: the register-resident fragments only
    (30_000, 512, 16, 100, True),     # exactly the register-resident 16 chunks, long lists
    (30_000, 544, 3, 5, True),        # one chunk of fragments in LDS
    (20_000, 64, 12, 8, True),        # two chunks per row
    (20_000, 768, 17, 8, False),      # 17 queries: two sub-tiles -> knn_sweep<2>
    (20_000, 770, 4, 8, False),       # D % 32 != 0 -> knn_sweep<1>
    (20_000, 1024, 4, 8, True),       # e5-large width: 32 chunks per row, half of the fragments in LDS
    (20_000, 1056, 4, 8, False),      # D > 1024 -> knn_sweep<1>
])
def test_lds_ring_sweep_matches_oracle(N, D, nq, k, ring, cuda_dev):
    """knn_sweep_ring (rows through wave-private LDS rings by non-temporal DMA, query fragments in registers / LDS) against
    the oracle: identical ids, distances to 1 ulp -- and d_stats[1] says which form of the sweep ran."""
    from adaptive_classifier import index as ix
    from oracle import c_oracle, synth
    P = synth.synth_unit_rows(N, D, seed=5)
    Q = synth.synth_unit_rows(nq, D, seed=6)
    ld = (D + 3) // 4 * 4
    store = torch.zeros((N, ld), dtype=torch.float32, device=cuda_dev)
    store[:, :D] = torch.from_numpy(P).to(cuda_dev)
    stats = torch.zeros(4, dtype=torch.int32, device=cuda_dev)
    Dd, Id = ix.knn_l2_topk(store, N, D, torch.from_numpy(Q).to(cuda_dev), k, stats=stats)
    torch.cuda.synchronize()
    oD, oI = c_oracle.knn_l2_topk(P, Q, k, 0)
    assert np.array_equal(Id.cpu().numpy(), oI)
    assert _ulp_close(Dd.cpu().numpy(), oD)
    assert int(stats[1].item()) == (1 if ring else 0)
    assert int(stats[0].item()) == 0
