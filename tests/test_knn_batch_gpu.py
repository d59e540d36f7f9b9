"""The prepared-store kNN paths -- ac_knn_prepare_store + ac_knn_l2_topk_batch: the GEMM-form fp16 proposal sweep (>= 64
queries) and the bandwidth-bound fp16-plane sweep (knn_plane_sweep, 1 .. 63 queries) -- against the exact oracle and against
the fp32 sweep path: same contract, ids bit-exact, distances = exact fp64 rounded once (1 ulp across summation orders).
The proposal arithmetic differs (one fp16 product), the answer may not."""
import numpy as np
import pytest
import torch

from helpers import near_tie_store

pytestmark = pytest.mark.gpu


def _ulp_close(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    return np.all(np.abs(a - b) <= np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)))


def _store(Ph, dev):
    N, D = Ph.shape
    ld = (D + 3) // 4 * 4
    P = torch.zeros((N, ld), dtype=torch.float32, device=dev)
    P[:, :D] = torch.from_numpy(Ph).to(dev)
    return P


def _run(Ph, Qh, k, dev, row_offset=0):
    from adaptive_classifier import index as ix
    N, D = Ph.shape
    P, Q = _store(Ph, dev), torch.from_numpy(Qh).to(dev)
    prep = ix.prepare_store(P, N, D)
    assert ix.batch_applies(N, Q.shape[0], k)
    st = torch.zeros(4, dtype=torch.int32, device=dev)
    ex = torch.empty((Q.shape[0], k), dtype=torch.float64, device=dev)
    Db, Ib = ix.knn_l2_topk(P, N, D, Q, k, row_offset=row_offset, stats=st, exact_out=ex, prepared=prep)
    Ds, Is = ix.knn_l2_topk(P, N, D, Q, k, row_offset=row_offset)              # the fp32 sweep path
    torch.cuda.synchronize()
    assert torch.equal(Ib, Is) and torch.equal(Db, Ds)
    assert torch.equal(ex.float(), Db)
    _run.form = int(st[1].item())                  # 2 = the fp16-plane sweep (knn_plane_sweep) ran
    return Db.cpu().numpy(), Ib.cpu().numpy(), int(st[0].item())


@pytest.mark.parametrize("N,D,nq,k", [
    (200_000, 768, 256, 16),       # BASELINE configs[1] shape at twice the rows
    (70_001, 100, 65, 8),          # D % 16 != 0 (zero-padded k-slots), ragged row / query tiles
    (131_072, 1024, 128, 32),      # e5-large width, exact tile multiples
    (65_536, 770, 64, 100),        # D % 4 != 0, the largest k of the batched path, the smallest store
])
def test_batched_path_matches_oracle(N, D, nq, k, cuda_dev):
    from oracle import c_oracle, synth
    Ph = synth.synth_unit_rows(N, D, 1)
    Qh = synth.synth_unit_rows(nq, D, 2)
    d, i, nfb = _run(Ph, Qh, k, cuda_dev, row_offset=7)
    oD, oI = c_oracle.knn_l2_topk_batch(Ph, Qh, k, row_offset=7)
    assert np.array_equal(i, oI), f"{(i != oI).sum()} id mismatches"
    assert _ulp_close(d, oD)
    assert nfb <= 2                                                   # uniform synthetic rows: the certificate holds


def test_batched_path_unnormalised_and_scaled_rows(cuda_dev):
    """Rows and queries far from unit norm (the error bound scales with (|p| + |q|)^2)."""
    from oracle import c_oracle
    rng = np.random.default_rng(5)
    Ph = (rng.standard_normal((80_000, 256)) * 3 + 0.5).astype(np.float32)
    Qh = (rng.standard_normal((70, 256)) * 0.3).astype(np.float32)
    d, i, _ = _run(Ph, Qh, 10, cuda_dev)
    oD, oI = c_oracle.knn_l2_topk_batch(Ph, Qh, 10)
    assert np.array_equal(i, oI) and _ulp_close(d, oD)


def test_batched_path_near_ties_and_duplicates(cuda_dev):
    """Stores full of fp32-unresolvable near-ties and exact duplicates: whatever the bf16 proposal orders, the result is
    the exact-definition top-k (fp64 re-rank, certificate, exact fallback; ties to the lower id)."""
    from oracle import c_oracle, synth
    D, k = 768, 16
    Ph, centres = near_tie_store(90_000, D, 7)
    Ph[50_000:50_300] = Ph[100:400]                                    # exact duplicates of earlier rows
    Qh = np.concatenate([(centres[:40] + synth.synth_unit_rows(40, D, 8) * 1e-3), Ph[100:124]]).astype(np.float32)
    d, i, nfb = _run(Ph, Qh, k, cuda_dev)
    oD, oI = c_oracle.knn_l2_topk_batch(Ph, Qh, k)
    assert np.array_equal(i, oI) and _ulp_close(d, oD)
    print("near-tie store, batched path: exact-fallback queries =", nfb)


def test_batched_path_candidate_overflow_goes_to_exact_fallback(cuda_dev):
    """One tight cluster: every row passes every query's sample threshold, the candidate buffers overflow, and all
    queries must come back exact through the fp64 fallback."""
    from oracle import c_oracle, synth
    D, k, N = 128, 8, 70_000
    c = synth.synth_unit_rows(1, D, 3)
    rng = np.random.default_rng(1)
    Ph = (c + rng.standard_normal((N, D)).astype(np.float32) * 1e-4).astype(np.float32)
    Qh = (c + rng.standard_normal((64, D)).astype(np.float32) * 1e-4).astype(np.float32)
    d, i, nfb = _run(Ph, Qh, k, cuda_dev)
    oD, oI = c_oracle.knn_l2_topk_batch(Ph, Qh, k)
    assert np.array_equal(i, oI) and _ulp_close(d, oD)
    assert nfb == 64


@pytest.mark.parametrize("two_phase", ["1", "0"])
def test_batched_path_thresholds_from_the_sweep_itself(two_phase, cuda_dev, monkeypatch):
    """65 .. 256 queries on a store of >= 256 row tiles take their thresholds from the main sweep's own first tile round
    (knn_batch.hip two_phase: per-workgroup minima, two grid barriers, the k'-th smallest of 256 minima per query) instead of a
    sample stage (AC_KNN_TWO_PHASE=0).  Both forms must return the exact-definition answer on the stores that stress the
    thresholds: near-ties and exact duplicates (many rows at the threshold), one tight cluster (every row passes: overflow ->
    exact fallback for every query), far-from-unit norms, k at the path's maximum (k' = 116 of 256 minima), a ragged last tile."""
    from oracle import c_oracle, synth
    monkeypatch.setenv("AC_KNN_TWO_PHASE", two_phase)
    D, k = 768, 16
    Ph, centres = near_tie_store(90_000, D, 7)
    Ph[50_000:50_300] = Ph[100:400]
    nc = min(50, len(centres))
    Qh = np.concatenate([(centres[:nc] + synth.synth_unit_rows(nc, D, 8) * 1e-3), Ph[100:100 + 100 - nc]]).astype(np.float32)     # 100 queries
    d, i, nfb = _run(Ph, Qh, k, cuda_dev)
    oD, oI = c_oracle.knn_l2_topk_batch(Ph, Qh, k)
    assert np.array_equal(i, oI) and _ulp_close(d, oD)
    # one tight cluster: everything passes every threshold
    D2, N2 = 128, 70_000
    c = synth.synth_unit_rows(1, D2, 3)
    rng = np.random.default_rng(1)
    P2 = (c + rng.standard_normal((N2, D2)).astype(np.float32) * 1e-4).astype(np.float32)
    Q2 = (c + rng.standard_normal((80, D2)).astype(np.float32) * 1e-4).astype(np.float32)
    d, i, nfb = _run(P2, Q2, 8, cuda_dev)
    oD, oI = c_oracle.knn_l2_topk_batch(P2, Q2, 8)
    assert np.array_equal(i, oI) and _ulp_close(d, oD) and nfb == 80
    # unnormalised rows, the largest k, a ragged last row tile, 256 queries
    P3 = (rng.standard_normal((66_001, 96)) * 3 + 0.5).astype(np.float32)
    Q3 = (rng.standard_normal((256, 96)) * 0.3).astype(np.float32)
    d, i, _ = _run(P3, Q3, 100, cuda_dev)
    oD, oI = c_oracle.knn_l2_topk_batch(P3, Q3, 100)
    assert np.array_equal(i, oI) and _ulp_close(d, oD)


def test_index_uses_the_batched_path_and_invalidates_on_change(cuda_dev, request):
    """HipFlatL2Index prepares the store lazily for many-query searches; the plane follows in-place row updates and is dropped
    by a compaction (remove_ids)."""
    from adaptive_classifier.index import HipFlatL2Index
    from oracle import c_oracle, synth
    D = 64
    X = synth.synth_unit_rows(70_000, D, 11)
    Q = synth.synth_unit_rows(80, D, 12)
    idx = HipFlatL2Index(D, device=cuda_dev)
    idx.add(X)
    from adaptive_classifier import index as ixm
    old = ixm.BATCH_MIN_PAIRS
    ixm.BATCH_MIN_PAIRS = 1.0                                          # (the auto heuristic would keep this small case on the sweep)
    request.addfinalizer(lambda: setattr(ixm, "BATCH_MIN_PAIRS", old))
    d, i = idx.search(Q, 5)
    assert idx._prepared is not None
    assert np.array_equal(i, c_oracle.knn_l2_topk_batch(X, Q, 5)[1])
    idx.update_rows([3], Q[:1])                                        # row 3 := query 0 -> must be found at distance 0
    assert idx._prepared is not None                                   # (round 4: the plane follows the row, ac_knn_update_store)
    d, i = idx.search(Q, 5)
    assert i[0, 0] == 3 and d[0, 0] == 0.0
    X2 = X.copy(); X2[3] = Q[0]
    assert np.array_equal(i, c_oracle.knn_l2_topk_batch(X2, Q, 5)[1])
    d1, i1 = idx.search(Q[:8], 5)                                      # few queries: the fp32 sweep path, same answer
    assert np.array_equal(i1, i[:8]) and np.array_equal(d1, d[:8])
    idx.remove_ids(np.array([10]))
    assert idx._prepared is None


def test_batched_path_two_threshold_stages_of_similar_size(cuda_dev):
    """Stores of ~0.92 - 1.5 M rows (a 10M-row store sharded 8 ways) are where the second threshold stage starts: its sample
    must be a few times the first stage's, or a share of the queries leaves it with fewer than k' rows, loses its threshold,
    overflows its candidate buffer and takes the exact fallback (round 3: 211 of 4096 queries at 1M rows, 1.3 s per batch)."""
    from adaptive_classifier import index as ix
    N, D, nq, k = 1_000_000, 64, 768, 32
    P = ix.synth_unit_rows(N, D, 1, device=cuda_dev)
    Q = ix.synth_unit_rows(nq, D, 2, device=cuda_dev)
    prep = ix.prepare_store(P, N, D)
    st = torch.zeros(4, dtype=torch.int32, device=cuda_dev)
    Db, Ib = ix.knn_l2_topk(P, N, D, Q, k, stats=st, prepared=prep)
    Ds, Is = ix.knn_l2_topk(P, N, D, Q, k)                                 # the fp32 sweep path (oracle-checked elsewhere)
    assert torch.equal(Ib, Is) and torch.equal(Db, Ds)
    assert int(st[0].item()) <= 1, "exact-fallback queries: %d" % int(st[0].item())


# ---------------------------------------------------------------------------------------------- knn_plane_sweep (1 .. 63 queries)
@pytest.mark.parametrize("N,D,nq,k", [
    (70_001, 768, 1, 16),          # one query, ragged last row tile
    (70_001, 768, 16, 32),
    (131_072, 768, 32, 32),        # a full 32-column tile, exact tile multiples
    (100_003, 768, 33, 32),        # 64-column tile, one column past the first sub-tile
    (90_000, 768, 48, 32),
    (65_536, 768, 63, 8),          # the smallest store, the most queries
    (70_001, 1024, 40, 32),        # e5-large width: the query tile fits LDS only 32 columns at a time -> two passes
    (70_001, 100, 17, 8),          # D % 16 != 0 (zero-padded k-slots)
    (80_000, 64, 63, 100),         # the largest k (k' = 124 candidates per list)
    (66_000, 770, 5, 1),           # D % 4 != 0, k = 1
])
def test_plane_sweep_matches_oracle(N, D, nq, k, cuda_dev):
    from oracle import c_oracle, synth
    Ph = synth.synth_unit_rows(N, D, 1)
    Qh = synth.synth_unit_rows(nq, D, 2)
    d, i, nfb = _run(Ph, Qh, k, cuda_dev, row_offset=11)
    assert _run.form == 2, "the fp16-plane sweep did not run"
    oD, oI = c_oracle.knn_l2_topk_batch(Ph, Qh, k, row_offset=11)
    assert np.array_equal(i, oI), f"{(i != oI).sum()} id mismatches"
    assert _ulp_close(d, oD)
    assert nfb <= 1                                                   # uniform synthetic rows: the certificate holds


def test_plane_sweep_unnormalised_near_ties_duplicates_and_clusters(cuda_dev):
    """The hard stores of the batched path's tests, with few queries: rows far from unit norm; a store full of fp32-
    unresolvable near-ties and exact duplicates; one tight cluster (every row is a candidate: the lists are pruned over and
    over, the certificate fails, the exact fallback answers)."""
    from oracle import c_oracle, synth
    rng = np.random.default_rng(5)
    Ph = (rng.standard_normal((80_000, 256)) * 3 + 0.5).astype(np.float32)
    Qh = (rng.standard_normal((20, 256)) * 0.3).astype(np.float32)
    d, i, _ = _run(Ph, Qh, 10, cuda_dev)
    assert _run.form == 2
    oD, oI = c_oracle.knn_l2_topk_batch(Ph, Qh, 10)
    assert np.array_equal(i, oI) and _ulp_close(d, oD)

    D, k = 768, 16
    Ph, centres = near_tie_store(90_000, D, 7)
    Ph[50_000:50_300] = Ph[100:400]                                    # exact duplicates of earlier rows
    Qh = np.concatenate([(centres[:30] + synth.synth_unit_rows(30, D, 8) * 1e-3), Ph[100:124]]).astype(np.float32)
    d, i, nfb = _run(Ph, Qh, k, cuda_dev)
    assert _run.form == 2
    oD, oI = c_oracle.knn_l2_topk_batch(Ph, Qh, k)
    assert np.array_equal(i, oI) and _ulp_close(d, oD)
    print("near-tie store, fp16-plane sweep: exact-fallback queries =", nfb)

    D, k, N = 128, 8, 70_000
    c = synth.synth_unit_rows(1, D, 3)
    rng = np.random.default_rng(1)
    Ph = (c + rng.standard_normal((N, D)).astype(np.float32) * 1e-4).astype(np.float32)
    Qh = (c + rng.standard_normal((40, D)).astype(np.float32) * 1e-4).astype(np.float32)
    d, i, nfb = _run(Ph, Qh, k, cuda_dev)
    assert _run.form == 2
    oD, oI = c_oracle.knn_l2_topk_batch(Ph, Qh, k)
    assert np.array_equal(i, oI) and _ulp_close(d, oD)
    assert nfb == 40


def test_index_prepares_for_small_batches_from_the_second_search_on(cuda_dev, request):
    """HipFlatL2Index: a big store's first few-query search runs the fp32 sweep (no preparation cost), the following ones the
    fp16-plane sweep; an in-place row update is folded into the plane, a compaction drops it.  Same answers throughout."""
    from adaptive_classifier import index as ixm
    from adaptive_classifier.index import HipFlatL2Index
    from oracle import c_oracle, synth
    D = 64
    X = synth.synth_unit_rows(70_000, D, 11)
    Q = synth.synth_unit_rows(9, D, 12)
    old = ixm.PLANE_MIN_ROWS
    ixm.PLANE_MIN_ROWS = 65_536                                        # (the auto heuristic keeps stores this small on the fp32 sweep)
    request.addfinalizer(lambda: setattr(ixm, "PLANE_MIN_ROWS", old))
    idx = HipFlatL2Index(D, device=cuda_dev)
    idx.add(X)
    want = c_oracle.knn_l2_topk_batch(X, Q, 5)[1]
    d0, i0 = idx.search(Q, 5)
    assert idx._prepared is None and np.array_equal(i0, want)
    d1, i1 = idx.search(Q, 5)
    assert idx._prepared is not None and int(idx._stats[1].item()) == 2
    assert np.array_equal(i1, want) and np.array_equal(d1, d0)
    idx.update_rows([3], Q[:1])                                        # the plane follows the row
    assert idx._prepared is not None
    d2, i2 = idx.search(Q, 5)
    assert int(idx._stats[1].item()) == 2 and i2[0, 0] == 3 and d2[0, 0] == 0.0
    idx.remove_ids(np.array([7]))                                      # a compaction drops it; prepared anew from the second search on
    assert idx._prepared is None
    d3, i3 = idx.search(Q, 5)
    assert idx._prepared is None
    d4, i4 = idx.search(Q, 5)
    assert idx._prepared is not None and np.array_equal(i4, i3) and np.array_equal(d4, d3)


def test_load_rows_prepare_makes_the_first_small_search_use_the_plane(cuda_dev, request):
    from adaptive_classifier import PrototypeMemory, index as ixm
    from oracle import c_oracle, synth
    old = ixm.PLANE_MIN_ROWS
    ixm.PLANE_MIN_ROWS = 65_536
    request.addfinalizer(lambda: setattr(ixm, "PLANE_MIN_ROWS", old))
    D, N = 64, 70_000
    X = synth.synth_unit_rows(N, D, 21)
    Q = synth.synth_unit_rows(5, D, 22)
    mem = PrototypeMemory(D, device=cuda_dev)
    mem.load_rows(ixm.synth_unit_rows(N, D, 21, device=cuda_dev), torch.arange(N, dtype=torch.int32) % 3, ["a", "b", "c"], prepare=True)
    assert mem.index._prepared is not None
    S, I, Dd = mem.search_batch(torch.from_numpy(Q).to(cuda_dev), 7)
    assert int(mem.index._stats[1].item()) == 2                        # the very first search ran the fp16-plane sweep
    assert np.array_equal(I.cpu().numpy(), c_oracle.knn_l2_topk_batch(X, Q, 7)[1])
    assert abs(float(S.sum(dim=1).mean()) - 1.0) < 1e-5


def test_sweeps_under_maximal_push_pressure(cuda_dev):
    """A store ordered by DECREASING distance to the queries' common direction: almost every row a workgroup meets beats its
    current threshold, so the candidate lists are pushed and pruned (radix select) about once per 28 rows instead of ~13 times
    per sweep -- the worst case for the list maintenance of all three sweep forms.  Results must still be the oracle's."""
    from oracle import c_oracle, synth
    N, D, k = 70_000, 64, 16
    rng = np.random.default_rng(3)
    c = synth.synth_unit_rows(1, D, 5)[0]
    X = (c + rng.standard_normal((N, D)).astype(np.float32) * 0.5).astype(np.float32)
    d = ((X - c) ** 2).sum(1)
    X = np.ascontiguousarray(X[np.argsort(-d, kind="stable")])              # farthest first, nearest last
    for nq in (3, 24, 40):                                                  # ring sweep / 32-column plane tile / 64-column plane tile
        Q = (c + rng.standard_normal((nq, D)).astype(np.float32) * 0.01).astype(np.float32)
        dd, ii, nfb = _run(X, Q, k, cuda_dev)                               # (_run also compares with the fp32 sweep path)
        assert _run.form == 2
        oD, oI = c_oracle.knn_l2_topk_batch(X, Q, k)
        assert np.array_equal(ii, oI) and _ulp_close(dd, oD)


def test_prepared_store_follows_appends_and_row_updates_incrementally(cuda_dev, request):
    """ac_knn_update_store through HipFlatL2Index: an index that is searched AND added to keeps its fp16 plane -- appended rows
    (across a tile boundary, across a reallocation of the store) and overwritten rows are folded in; a row that moves the
    store's power-of-two scale drops the plane (prepared anew later).  The answers are the oracle's throughout."""
    from adaptive_classifier import index as ixm
    from adaptive_classifier.index import HipFlatL2Index
    from oracle import c_oracle, synth
    old = ixm.PLANE_MIN_ROWS
    ixm.PLANE_MIN_ROWS = 65_536
    request.addfinalizer(lambda: setattr(ixm, "PLANE_MIN_ROWS", old))
    D, k = 64, 7
    X = synth.synth_unit_rows(70_000, D, 31)
    Q = synth.synth_unit_rows(6, D, 32)
    idx = HipFlatL2Index(D, device=cuda_dev)
    idx.add(X)

    def check(Xh, plane):
        d, i = idx.search(Q, k)
        assert (int(idx._stats[1].item()) == 2) == plane, (int(idx._stats[1].item()), plane)
        oD, oI = c_oracle.knn_l2_topk_batch(Xh, Q, k)
        assert np.array_equal(i, oI) and _ulp_close(d, oD)
    check(X, False)                                    # first search of a fresh store: fp32 sweep
    check(X, True)                                     # second: prepared
    planes0 = idx._prepared[0]
    A = synth.synth_unit_rows(300, D, 33)              # 70 000 -> 70 300 rows: the last tile fills up and a new one starts
    A[5] = Q[2] * 0.999                                # ... and one of the new rows is query 2's nearest neighbour
    idx.add(torch.from_numpy(A).to(cuda_dev))
    X1 = np.concatenate([X, A])
    assert idx._prepared is not None
    check(X1, True)
    idx.update_rows([3, 4, 5, 69_999, 70_200], np.stack([Q[0], Q[1] * 1.001, Q[3], Q[4] * 0.998, Q[5]]))      # in place, two tiles
    X2 = X1.copy(); X2[[3, 4, 5, 69_999, 70_200]] = np.stack([Q[0], Q[1] * 1.001, Q[3], Q[4] * 0.998, Q[5]])
    assert idx._prepared is not None
    check(X2, True)
    B = synth.synth_unit_rows(90_000, D, 34)           # beyond the store's capacity: matrix and plane buffers are reallocated
    idx.add(B)                                         # (host rows: uploaded by the next search)
    X3 = np.concatenate([X2, B])
    check(X3, True)
    assert idx._prepared is not None and idx._prepared[0].numel() > planes0.numel()
    big = (Q[0] * 40.0)[None, :]                       # |p| = 40: the store's scale 2^e_p must grow -> every plane entry is stale
    idx.add(torch.from_numpy(big).to(cuda_dev))
    X4 = np.concatenate([X3, big])
    assert idx._prepared is None
    check(X4, True)                                    # prepared anew by this search (the store has been searched before)
    idx.remove_ids(np.array([0, 70_005]))              # compaction: the plane cannot follow
    assert idx._prepared is None
    X5 = np.delete(X4, [0, 70_005], axis=0)
    check(X5, False)                                   # like a fresh store: fp32 sweep first,
    check(X5, True)                                    # prepared from the second search on
