"""TEST INFRASTRUCTURE -- not part of the product path.

CPU restatement of the kNN site T1 of the reference: `faiss.IndexFlatL2.search`
as called by PrototypeMemory.get_nearest_prototypes (memory.py:113-114), i.e. exact
squared-L2 brute force, k smallest per query in ascending order.

faiss (third party, `faiss-cpu>=1.7.4`, requirements.txt:4) is neither vendored under
/root/reference nor installable here, so its SIMD / sgemm rounding cannot be reproduced;
**kNN parity versus a real faiss build is therefore "parity unpinned"** (SURVEY 8c).
What is pinned is faiss's *specification*: the k rows with the smallest exact squared
distance.  This oracle evaluates that specification without fp32 roundoff:

    d(q, n) = sum_c (float64(P[n,c]) - float64(Q[q,c]))**2      (fp64; the difference of two
                                                                fp32 values and its square are exact
                                                                in fp64, only the 768-term sum rounds,
                                                                rel. error <= 1e-13)
    order by (d, n) ascending  -> ties go to the lower row id (faiss's heap also prefers lower ids
                                  on equal distances for IndexFlat)
    returned distance = float32(d)

Any correct fp32 implementation (faiss's two code paths included) returns these ids except
where two candidates are closer than its own roundoff.
"""
import numpy as np

FLT_MAX = np.float32(np.finfo(np.float32).max)


def exact_sqdist(P, Q, chunk_rows=32768):
    """fp64 [nq, N] exact squared distances (chunked over rows of P)."""
    P = np.asarray(P, dtype=np.float32)
    Q = np.asarray(Q, dtype=np.float32)
    nq, N = Q.shape[0], P.shape[0]
    out = np.empty((nq, N), dtype=np.float64)
    Q64 = Q.astype(np.float64)
    for s in range(0, N, chunk_rows):
        e = min(N, s + chunk_rows)
        P64 = P[s:e].astype(np.float64)
        for q in range(nq):
            diff = P64 - Q64[q]
            out[q, s:e] = np.einsum("nc,nc->n", diff, diff)
    return out


def knn_l2_topk(P, Q, k, row_offset=0, return_exact=False):
    """(float32 [nq,k], int64 [nq,k]) -- the oracle for ac_knn_l2_topk.

    k > N pads with (FLT_MAX, -1), faiss's convention for an under-full heap.
    return_exact=True: (float64 exact distances [nq,k] padded with +inf, ids) -- the oracle for ac_knn_l2_topk_x's
    d_outD64, i.e. what one row shard contributes to a sharded search.
    """
    P = np.asarray(P, dtype=np.float32)
    Q = np.asarray(Q, dtype=np.float32)
    nq, N = Q.shape[0], P.shape[0]
    outD = np.full((nq, k), FLT_MAX, dtype=np.float32)
    outE = np.full((nq, k), np.inf, dtype=np.float64)
    outI = np.full((nq, k), -1, dtype=np.int64)
    if N == 0 or nq == 0:
        return (outE, outI) if return_exact else (outD, outI)
    d = exact_sqdist(P, Q)
    kk = min(k, N)
    ids = np.arange(N, dtype=np.int64)
    for q in range(nq):
        if kk < N:
            # partial selection then exact (d, id) ordering of a safe superset
            kth = np.partition(d[q], kk - 1)[kk - 1]
            cand = ids[d[q] <= kth]
        else:
            cand = ids
        order = np.lexsort((cand, d[q][cand]))[:kk]
        sel = cand[order]
        outD[q, :kk] = d[q][sel].astype(np.float32)
        outE[q, :kk] = d[q][sel]
        outI[q, :kk] = sel + row_offset
    return (outE, outI) if return_exact else (outD, outI)


def topk_merge(D_in, I_in, k):
    """Oracle for ac_topk_merge / ac_topk_merge_f64: [shards, nq, k] ascending lists -> global top-k by (d, id) in the
    precision of D_in (float32, or the shards' exact float64 distances); the distances come out rounded to float32."""
    D_in = np.asarray(D_in)
    D_in = D_in.astype(np.float64 if D_in.dtype == np.float64 else np.float32)
    I_in = np.asarray(I_in, dtype=np.int64)
    S, nq, kk = D_in.shape
    outD = np.full((nq, k), FLT_MAX, dtype=np.float32)
    outI = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        d = D_in[:, q, :].reshape(-1)
        i = I_in[:, q, :].reshape(-1)
        keep = i >= 0
        d, i = d[keep], i[keep]
        order = np.lexsort((i, d))[:k]
        outD[q, :len(order)] = d[order]
        outI[q, :len(order)] = i[order]
    return outD, outI


def proto_scores(D, I):
    """memory.py:117,129-130: softmax(exp(-d)) over the valid hits of each query, fp32."""
    D = np.asarray(D, dtype=np.float32)
    I = np.asarray(I)
    out = np.zeros_like(D, dtype=np.float32)
    for q in range(D.shape[0]):
        v = I[q] >= 0
        if not v.any():
            continue
        s = np.exp(-D[q][v]).astype(np.float32)            # memory.py:117
        e = np.exp((s - s.max()).astype(np.float32))       # torch softmax, fp32
        out[q][v] = (e / e.sum(dtype=np.float32)).astype(np.float32)
    return out
