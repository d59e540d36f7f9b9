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
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.oracle_num_threads.restype = ctypes.c_int
        for fn in (L.oracle_knn_l2_topk, L.oracle_knn_l2_topk_f32):
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
