"""TEST INFRASTRUCTURE -- not part of the product path.

A stand-in for the `faiss` module exposing exactly the protocol memory.py uses
(memory.py:34,106,113,114,158,159,164,172,182,190,242): IndexFlatL2(d) with add / search /
remove_ids / ntotal.  Injecting it as sys.modules["faiss"] lets the UNMODIFIED reference
package (/root/reference/src) import and run in the build container, which is how the
fixtures under tests/golden/ are generated (tests/golden/gen_golden.py).

search() is the numpy oracle (oracle/knn_oracle.py): exact squared L2, ascending, ties to the
lower id.  Real faiss is not installable here ("parity unpinned" vs real faiss).
"""
import sys
import types

import numpy as np

from . import knn_oracle


class IndexFlatL2:
    def __init__(self, d):
        self.d = int(d)
        self._x = np.zeros((0, self.d), dtype=np.float32)

    @property
    def ntotal(self):
        return self._x.shape[0]

    def add(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.d)
        self._x = np.concatenate([self._x, x], axis=0)

    def remove_ids(self, ids):
        ids = np.asarray(ids).reshape(-1).astype(np.int64)
        keep = np.ones(self.ntotal, dtype=bool)
        keep[ids[(ids >= 0) & (ids < self.ntotal)]] = False
        removed = int((~keep).sum())
        self._x = self._x[keep]          # IndexFlat compacts: later rows shift down
        return removed

    def search(self, x, k):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.d)
        return knn_oracle.knn_l2_topk(self._x, x, int(k))

    def reset(self):
        self._x = np.zeros((0, self.d), dtype=np.float32)


def install():
    """Register the shim as `faiss` (no-op if a real faiss is importable)."""
    try:
        import faiss  # noqa: F401
        return False
    except Exception:
        mod = types.ModuleType("faiss")
        mod.IndexFlatL2 = IndexFlatL2
        mod.__shim__ = True
        sys.modules["faiss"] = mod
        return True
