"""CPU suite: the SECOND oracles (fp32 restatements of faiss's two flat-L2 code paths, oracle/knn_faiss_forms.c)
against the exact-definition oracle the HIP path is pinned to.

faiss itself is absent (kNN parity vs a real faiss build stays "parity unpinned"); the next-best evidence is that
every fp32 evaluation order faiss could use -- sequential / AVX2 / AVX-512 sums of (x-y)^2 for nq < 20, and
|x|^2 + |y|^2 - 2 x.y with an sgemm-style dot for nq >= 20 -- returns the exact-definition ids except inside
provable fp32 near-ties.  The criterion (c_oracle.classify_disagreements): at every position where the ids differ,
|d_exact(form id) - d_exact(exact id)| <= 2E with E the a-priori rounding bound of the form.
"""
import numpy as np
import pytest

from oracle import c_oracle, knn_oracle, synth


from helpers import near_tie_store as _near_tie_store


def test_batched_exact_oracle_equals_per_query_oracle():
    P = synth.synth_unit_rows(5000, 768, 1)
    Q = synth.synth_unit_rows(9, 768, 2)
    D1, I1 = c_oracle.knn_l2_topk(P, Q, 16, row_offset=123)
    D2, I2 = c_oracle.knn_l2_topk_batch(P, Q, 16, row_offset=123)
    assert np.array_equal(I1, I2) and np.array_equal(D1, D2)
    # chunked form (what the 10M-row GPU test uses) == unchunked
    chunks = [(o, P[o:o + 1300]) for o in range(0, 5000, 1300)]
    D3, I3 = c_oracle.knn_l2_topk_chunked(chunks, Q, 16)
    assert np.array_equal(I3, I1 - 123) and np.array_equal(D3, D1)
    D4, I4 = c_oracle.knn_l2_topk_batch(P[:7], Q, 16)              # k > N pads like faiss
    assert (I4[:, 7:] == -1).all() and (D4[:, 7:] == np.finfo(np.float32).max).all()
    # duplicates: ties to the lower id
    Pd = np.concatenate([P[:20]] * 3)
    _, I5 = c_oracle.knn_l2_topk_batch(Pd, P[:2], 6)
    assert I5[0, :3].tolist() == [0, 20, 40]


def test_forms_agree_with_numpy_fp32_restatement():
    """The C forms really are the arithmetic they claim: scalar-sequential forms equal a step-by-step numpy fp32
    evaluation bit for bit."""
    rng = np.random.default_rng(0)
    P = rng.standard_normal((40, 37)).astype(np.float32)
    Q = rng.standard_normal((3, 37)).astype(np.float32)
    Dd, Id = c_oracle.knn_form_topk(P, Q, 40, "seq_scalar")
    Db, Ib = c_oracle.knn_form_topk(P, Q, 40, "blas_scalar")
    for q in range(3):
        d_seq = np.zeros(40, np.float32)
        ip = np.zeros(40, np.float32)
        pn = np.zeros(40, np.float32)
        qn = np.float32(0)
        for c in range(37):
            t = (Q[q, c] - P[:, c]).astype(np.float32)
            d_seq = (d_seq + (t * t).astype(np.float32)).astype(np.float32)
            ip = (ip + (Q[q, c] * P[:, c]).astype(np.float32)).astype(np.float32)
            pn = (pn + (P[:, c] * P[:, c]).astype(np.float32)).astype(np.float32)
            qn = np.float32(qn + np.float32(Q[q, c] * Q[q, c]))
        d_blas = np.maximum(((qn + pn).astype(np.float32) - (np.float32(2) * ip).astype(np.float32)).astype(np.float32), 0)
        assert np.array_equal(Dd[q], np.sort(d_seq, kind="stable"))
        assert np.array_equal(Id[q], np.argsort(d_seq, kind="stable"))
        assert np.array_equal(Db[q], np.sort(d_blas, kind="stable"))
        assert np.array_equal(Ib[q], np.argsort(d_blas, kind="stable"))


@pytest.mark.parametrize("store", ["uniform", "near_ties"])
def test_every_form_disagreement_is_a_provable_near_tie(store):
    D, k = 768, 16
    if store == "uniform":
        P = synth.synth_unit_rows(20000, D, 1)
        Q = synth.synth_unit_rows(24, D, 2)
    else:
        P, centres = _near_tie_store(12000, D, 7)
        Q = (centres[:24] + synth.synth_unit_rows(24, D, 8) * 1e-3).astype(np.float32)
    _, I_exact = c_oracle.knn_l2_topk_batch(P, Q, k)
    assert np.array_equal(I_exact[:3], knn_oracle.knn_l2_topk(P, Q[:3], k)[1])       # numpy oracle agrees
    total = {}
    for name in c_oracle.FORMS:
        Df, If = c_oracle.knn_form_topk(P, Q, k, name)
        bound = c_oracle.form_error_bound(P, Q, name)
        n_mis, n_unexplained = c_oracle.classify_disagreements(P, Q, I_exact, If, bound)
        total[name] = n_mis
        assert n_unexplained == 0, (name, n_mis, n_unexplained)
        # the fp32 values themselves sit within the bound of the exact distance of the id they come with
        de = c_oracle.exact_dist_of_ids(P, Q, If)
        assert np.all(np.abs(Df.astype(np.float64) - de) <= bound[0] * de + bound[1][:, None]), name
        assert np.all(np.diff(Df, axis=1) >= 0)
    if store == "near_ties":
        # the store is built so that fp32 rounding DOES reorder neighbours: the criterion is exercised, not vacuous
        assert total["seq_avx2_fma"] > 0 and total["blas_avx2_fma"] > 0, total
    else:
        # well-separated synthetic rows: the norm/dot forms may still swap a pair; the difference forms agree
        assert total["seq_avx2_fma"] <= 2 and total["seq_scalar"] <= 2, total


def test_near_tie_criterion_rejects_a_wrong_answer():
    """The criterion has teeth: swapping in a row that is NOT a near-tie is flagged."""
    P = synth.synth_unit_rows(3000, 256, 3)
    Q = synth.synth_unit_rows(4, 256, 4)
    _, I_exact = c_oracle.knn_l2_topk_batch(P, Q, 8)
    bad = I_exact.copy()
    bad[1, 3] = int(np.setdiff1d(np.arange(3000), I_exact[1])[0])      # an arbitrary far row
    n_mis, n_unexplained = c_oracle.classify_disagreements(P, Q, I_exact, bad, c_oracle.form_error_bound(P, Q, "blas_scalar"))
    assert n_mis == 1 and n_unexplained == 1
