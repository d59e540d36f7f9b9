"""TEST INFRASTRUCTURE -- ctypes binding of oracle/liboracle.so (knn_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        srcs = [os.path.join(_HERE, f) for f in ("knn_oracle.c", "knn_faiss_forms.c")]
        if not os.path.exists(path) or any(os.path.getmtime(f) > os.path.getmtime(path) for f in srcs):
            build()
        L = ctypes.CDLL(path)
        L.oracle_num_threads.restype = ctypes.c_int
        for fn in (L.oracle_knn_l2_topk, L.oracle_knn_l2_topk_f32, L.oracle_knn_form_topk,
                   L.oracle_exact_dist_of_ids, L.oracle_knn_l2_topk_batch):
            fn.restype = ctypes.c_int
        _LIB = L
    return _LIB


def num_threads():
    return lib().oracle_num_threads()


def set_threads(n):
    lib().oracle_set_threads(ctypes.c_int(int(n)))


def usable_cores():
    """CPUs this process may actually use: min(affinity mask, cgroup v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def knn_l2_topk(P, Q, k, row_offset=0):
    P = np.ascontiguousarray(P, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    N, D = P.shape
    nq = Q.shape[0]
    outD = np.empty((nq, k), dtype=np.float32)
    outI = np.empty((nq, k), dtype=np.int64)
    rc = lib().oracle_knn_l2_topk(
        P.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(N), ctypes.c_int64(P.shape[1]), ctypes.c_int(D),
        Q.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(nq), ctypes.c_int64(Q.shape[1]), ctypes.c_int(k),
        ctypes.c_int64(row_offset),
        outD.ctypes.data_as(ctypes.c_void_p), outI.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError(f"oracle_knn_l2_topk failed rc={rc}")
    return outD, outI


def knn_l2_topk_f32(P, Q, k):
    P = np.ascontiguousarray(P, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    N, D = P.shape
    nq = Q.shape[0]
    outD = np.empty((nq, k), dtype=np.float32)
    outI = np.empty((nq, k), dtype=np.int64)
    rc = lib().oracle_knn_l2_topk_f32(
        P.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(N), ctypes.c_int64(P.shape[1]), ctypes.c_int(D),
        Q.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(nq), ctypes.c_int64(Q.shape[1]), ctypes.c_int(k),
        outD.ctypes.data_as(ctypes.c_void_p), outI.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError(f"oracle_knn_l2_topk_f32 failed rc={rc}")
    return outD, outI


def knn_l2_topk_batch(P, Q, k, row_offset=0):
    """Exact oracle, row-outer / query-inner (same definition as knn_l2_topk; for big stores x many queries)."""
    P = np.ascontiguousarray(P, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    N, D = P.shape
    nq = Q.shape[0]
    outD = np.empty((nq, k), dtype=np.float32)
    outI = np.empty((nq, k), dtype=np.int64)
    rc = lib().oracle_knn_l2_topk_batch(
        P.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(N), ctypes.c_int64(P.shape[1]), ctypes.c_int(D),
        Q.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(nq), ctypes.c_int64(Q.shape[1]), ctypes.c_int(k),
        ctypes.c_int64(row_offset), outD.ctypes.data_as(ctypes.c_void_p), outI.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError(f"oracle_knn_l2_topk_batch failed rc={rc}")
    return outD, outI


def knn_l2_topk_chunked(row_chunks, Q, k):
    """Exact oracle over a store delivered in row chunks (an iterable of (row_offset, float32 [n, D]) blocks, e.g.
    device -> host copies of a 10M-row store): per-chunk batched oracle + knn_oracle.topk_merge."""
    from . import knn_oracle
    Ds, Is = [], []
    for off, rows in row_chunks:
        d, i = knn_l2_topk_batch(rows, Q, k, row_offset=off)
        Ds.append(d)
        Is.append(i)
    return knn_oracle.topk_merge(np.stack(Ds), np.stack(Is), k)


FORMS = {
    # name: (form, lanes, fma) -- see knn_faiss_forms.c
    "seq_scalar": (0, 1, 0), "seq_avx2": (0, 8, 0), "seq_avx2_fma": (0, 8, 1), "seq_avx512_fma": (0, 16, 1),
    "seq_avx512x2_fma": (0, 32, 1),
    "blas_scalar": (1, 1, 0), "blas_avx2_fma": (1, 8, 1), "blas_avx512_fma": (1, 16, 1), "blas_avx512x2_fma": (1, 32, 1),
}


def knn_form_topk(P, Q, k, name):
    """One member of the fp32 faiss-form family (second oracles; knn_faiss_forms.c)."""
    form, lanes, fma = FORMS[name]
    P = np.ascontiguousarray(P, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    N, D = P.shape
    nq = Q.shape[0]
    outD = np.empty((nq, k), dtype=np.float32)
    outI = np.empty((nq, k), dtype=np.int64)
    rc = lib().oracle_knn_form_topk(
        P.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(N), ctypes.c_int64(P.shape[1]), ctypes.c_int(D),
        Q.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(nq), ctypes.c_int64(Q.shape[1]), ctypes.c_int(k),
        ctypes.c_int(form), ctypes.c_int(lanes), ctypes.c_int(fma),
        outD.ctypes.data_as(ctypes.c_void_p), outI.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError(f"oracle_knn_form_topk failed rc={rc}")
    return outD, outI


def exact_dist_of_ids(P, Q, ids):
    """fp64 exact squared distance of every (query, id) pair; ids int64 [nq, k] -> float64 [nq, k]."""
    P = np.ascontiguousarray(P, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    out = np.empty(ids.shape, dtype=np.float64)
    lib().oracle_exact_dist_of_ids(
        P.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(P.shape[0]), ctypes.c_int64(P.shape[1]), ctypes.c_int(P.shape[1]),
        Q.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(Q.shape[0]), ctypes.c_int64(Q.shape[1]),
        ids.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(ids.shape[1]), out.ctypes.data_as(ctypes.c_void_p))
    return out


def form_error_bound(P, Q, name):
    """A-priori rounding bound of a form, as (eps_rel, E_abs[nq]): |fp32 value - exact d| <= eps_rel * d + E_abs.
    Difference forms: every term is non-negative and passes through <= D + 3 roundings -> purely RELATIVE,
                      eps_rel = 1.01 (D + 3) u, E_abs = 0.
    Norm/dot forms:   |x|^2 + |y|^2 - 2 x.y cancels -> ABSOLUTE in the operand norms, E_abs = 1.01 (D + 3) u (|p|max + |q|)^2.
    u = 2^-24 (fp32 unit roundoff)."""
    form = FORMS[name][0]
    P64 = np.asarray(P, dtype=np.float64)
    Q64 = np.asarray(Q, dtype=np.float64)
    D = P64.shape[1]
    g = 1.01 * (D + 3) * 2.0 ** -24
    if form == 0:
        return g, np.zeros(Q64.shape[0])
    pn = np.sqrt((P64 * P64).sum(1).max()) if P64.shape[0] else 0.0
    qn = np.sqrt((Q64 * Q64).sum(1))
    return 0.0, g * (pn + qn) ** 2


def classify_disagreements(P, Q, I_exact, I_form, bound):
    """Compare a form's ids with the exact oracle's.  Returns (n_mismatched_positions, n_unexplained).
    If every fp32 value f(row) obeys |f - d| <= eps d + E (bound = (eps, E[nq])), then sorting by f instead of d moves
    the j-th order statistic by at most that much, so the form's j-th id MUST satisfy
        |d_exact(I_form[q, j]) - d_exact(I_exact[q, j])| <= 2 (eps * max(d, d') + E[q]) / (1 - eps)
    A mismatching position that satisfies it is a provable fp32 near-tie (EXPLAINED); one that does not is a bug."""
    eps, E = bound
    de_form = exact_dist_of_ids(P, Q, I_form)
    de_exact = exact_dist_of_ids(P, Q, I_exact)
    mism = I_form != I_exact
    gap = np.abs(de_form - de_exact)
    tol = 2.0 * (eps * np.maximum(de_form, de_exact) + np.asarray(E)[:, None]) / (1.0 - eps)
    unexplained = mism & ~(gap <= tol)
    return int(mism.sum()), int(unexplained.sum())
