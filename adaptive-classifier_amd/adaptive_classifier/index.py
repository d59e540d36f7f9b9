"""HipFlatL2Index: the `faiss.IndexFlatL2` protocol used by memory.py, on MI355X HBM.

Protocol kept (reference call sites in /root/reference/src/adaptive_classifier/memory.py):
  IndexFlatL2(d) :34,164,182,242 | .add(x[n,d]) :159,172,190 | .search(x[nq,d], k) :114 |
  .remove_ids(ids) :158 (compacting) | .ntotal :106,113

Rows live in one resident, row-major fp32 device matrix with a 16-byte aligned, zero padded
leading dimension (growth by doubling, so `add` is amortised O(1) like faiss's vector).
search() is one call into the C ABI (`ac_knn_l2_topk`): exact squared L2, ascending, ties to
the lower id.
"""
import ctypes

import numpy as np
import torch

from . import _native as nv


def knn_workspace_bytes(N, D, nq, k):
    b = ctypes.c_size_t(0)
    nv.check(nv.lib().ac_knn_l2_topk_workspace(N, D, nq, k, ctypes.byref(b)), "ac_knn_l2_topk_workspace")
    return b.value


# Where the prepared-store paths apply (library limits: N >= 65536, k <= 100) and pay.
#   >= 64 queries: the GEMM-form proposal sweep (knn_batch_sweep); its fixed cost (sample stages, thresholds, query plane, a deeper
#      merge: ~0.3 ms) beats the fp32 sweep from ~20 M query-row pairs on (measured, round 3, fp32 sweep vs batched: 256 x 100k 0.56
#      vs 0.39 ms inside the predict step; 1024 x 2M x 1024 56 vs 6.5 ms; 4096 x 10M 552 vs 78 ms)
#   1 .. 63 queries (round 4): ONE bandwidth-bound pass over the fp16 plane with the query tile resident (knn_plane_sweep): half
#      the bytes of the fp32 sweep; pays once the sweep, not the launch chain, is what a search costs (PLANE_MIN_ROWS)
BATCH_MIN_QUERIES, BATCH_MIN_ROWS, BATCH_MAX_K, BATCH_MIN_PAIRS = 64, 65536, 100, 2.0e7
PLANE_MIN_ROWS = 1 << 18


def store_buffers(rows, D, device):
    """Empty (plane int16, norms fp32) buffers of a prepared store of up to `rows` rows (`ac_knn_store_bytes`)."""
    pb, nb = ctypes.c_size_t(0), ctypes.c_size_t(0)
    nv.check(nv.lib().ac_knn_store_bytes(rows, D, ctypes.byref(pb), ctypes.byref(nb)), "ac_knn_store_bytes")
    return (torch.empty(pb.value // 2, dtype=torch.int16, device=device), torch.empty(nb.value // 4, dtype=torch.float32, device=device))


def store_buffers_sizes(rows, D):
    """(plane elements, norm elements) a prepared store of `rows` rows occupies"""
    pb, nb = ctypes.c_size_t(0), ctypes.c_size_t(0)
    nv.check(nv.lib().ac_knn_store_bytes(rows, D, ctypes.byref(pb), ctypes.byref(nb)), "ac_knn_store_bytes")
    return pb.value // 2, nb.value // 4


def update_store(P, n_old, n_new, D, prepared, row0, nrows):
    """`ac_knn_update_store`: rows [row0, row0 + nrows) of P changed / were appended.  Returns True when the prepared store is
    ready again, False when the new rows moved the store's power-of-two scale (every plane entry is stale: prepare anew).
    Reads a 4-byte verdict back (one stream sync per update)."""
    planes, norms = prepared
    flag = torch.zeros(1, dtype=torch.int32, device=P.device)
    with torch.cuda.device(P.device):
        nv.check(nv.lib().ac_knn_update_store(nv.ptr(P), n_old, n_new, P.stride(0), D, nv.ptr(planes), nv.ptr(norms), row0, nrows,
                                              nv.ptr(flag), nv.stream_ptr(P.device)), "ac_knn_update_store")
    return int(flag.item()) == 0


def prepare_store(P, N, D, capacity=None):
    """(plane, norms) of the first N rows of the store P for the batched search (`ac_knn_prepare_store`): one fp16
    operand plane + |p|^2 per row.  Costs two passes over the rows and 2 B per element.  After rows change: `update_store`
    (appended / overwritten rows) or prepare anew.  capacity: size the buffers for that many rows (stores that grow)."""
    nv.require_gpu()
    pb, nb = ctypes.c_size_t(0), ctypes.c_size_t(0)
    nv.check(nv.lib().ac_knn_store_bytes(max(N, capacity or 0), D, ctypes.byref(pb), ctypes.byref(nb)), "ac_knn_store_bytes")
    planes = torch.empty(pb.value // 2, dtype=torch.int16, device=P.device)
    norms = torch.empty(nb.value // 4, dtype=torch.float32, device=P.device)
    with torch.cuda.device(P.device):
        nv.check(nv.lib().ac_knn_prepare_store(nv.ptr(P), N, P.stride(0), D, nv.ptr(planes), nv.ptr(norms),
                                               nv.stream_ptr(P.device)), "ac_knn_prepare_store")
    return planes, norms


def batch_applies(N, nq, k, auto=False):
    """Library limits of ac_knn_l2_topk_batch (any number of queries: below 64 it runs the fp16-plane sweep); auto=True adds the
    size heuristic the index uses to pick a path."""
    ok = nq >= 1 and N >= BATCH_MIN_ROWS and k <= BATCH_MAX_K
    if not ok or not auto:
        return ok
    return float(N) * nq >= BATCH_MIN_PAIRS if nq >= BATCH_MIN_QUERIES else N >= PLANE_MIN_ROWS


def knn_l2_topk(P, N, D, Q, k, row_offset=0, out=None, workspace=None, stats=None, exact_out=None, prepared=None):
    """Low-level device call.  P: [>=N, ldP] fp32 cuda tensor, Q: [nq, >=D] fp32 cuda tensor.

    Returns (dist fp32 [nq,k], ids int64 [nq,k]) on the same device.  Asynchronous.
    exact_out: optional float64 [nq,k] cuda tensor that receives the exact fp64 distances (shard merges).
    prepared: optional (planes, norms) from `prepare_store`: the search then proposes from the store's fp16 plane
              (`ac_knn_l2_topk_batch`: one bandwidth-bound pass for < 64 queries, the GEMM-form sweep on the matrix pipe from 64
              on) -- same exact result, half the bytes / several times the throughput.
    """
    if prepared is not None and batch_applies(N, Q.shape[0], k):
        return _knn_l2_topk_batch(P, N, D, Q, k, prepared, row_offset, out, workspace, stats, exact_out)
    nv.require_gpu()
    assert P.dtype == torch.float32 and Q.dtype == torch.float32 and P.is_cuda and Q.is_cuda
    assert P.stride(1) == 1 and Q.stride(1) == 1
    nq = Q.shape[0]
    dev = Q.device
    if out is None:
        outD = torch.empty((nq, k), dtype=torch.float32, device=dev)
        outI = torch.empty((nq, k), dtype=torch.int64, device=dev)
    else:
        outD, outI = out
    need = knn_workspace_bytes(N, D, nq, k)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = nv.lib().ac_knn_l2_topk_x(
            nv.ptr(P), N, P.stride(0), D, nv.ptr(Q), nq, Q.stride(0), k, row_offset,
            nv.ptr(outD), nv.ptr(exact_out), nv.ptr(outI), nv.ptr(workspace), workspace.numel(),
            nv.ptr(stats), nv.stream_ptr(dev))
    nv.check(rc, "ac_knn_l2_topk_x")
    return outD, outI


def _knn_l2_topk_batch(P, N, D, Q, k, prepared, row_offset, out, workspace, stats, exact_out):
    nv.require_gpu()
    planes, norms = prepared
    nq, dev = Q.shape[0], Q.device
    if out is None:
        outD = torch.empty((nq, k), dtype=torch.float32, device=dev)
        outI = torch.empty((nq, k), dtype=torch.int64, device=dev)
    else:
        outD, outI = out
    b = ctypes.c_size_t(0)
    nv.check(nv.lib().ac_knn_l2_topk_batch_workspace(N, D, nq, k, ctypes.byref(b)), "ac_knn_l2_topk_batch_workspace")
    if workspace is None or workspace.numel() < b.value:
        workspace = torch.empty(b.value, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = nv.lib().ac_knn_l2_topk_batch(
            nv.ptr(P), N, P.stride(0), D, nv.ptr(planes), nv.ptr(norms), nv.ptr(Q), nq, Q.stride(0), k, row_offset,
            nv.ptr(outD), nv.ptr(exact_out), nv.ptr(outI), nv.ptr(workspace), workspace.numel(), nv.ptr(stats),
            nv.stream_ptr(dev))
    nv.check(rc, "ac_knn_l2_topk_batch")
    return outD, outI


def knn_batch_workspace_bytes(N, D, nq, k):
    b = ctypes.c_size_t(0)
    nv.check(nv.lib().ac_knn_l2_topk_batch_workspace(N, D, nq, k, ctypes.byref(b)), "ac_knn_l2_topk_batch_workspace")
    return b.value


def knn_l2_topk_exact(P, N, D, Q, k, row_offset=0, workspace=None, stats=None, prepared=None):
    """(exact fp64 dist [nq,k], ids [nq,k]): what a row shard contributes to a sharded search."""
    ex = torch.empty((Q.shape[0], k), dtype=torch.float64, device=Q.device)
    _, I = knn_l2_topk(P, N, D, Q, k, row_offset=row_offset, workspace=workspace, stats=stats, exact_out=ex, prepared=prepared)
    return ex, I


class HipFlatL2Index:
    """Drop-in for faiss.IndexFlatL2 as used by PrototypeMemory.

    Host-side bookkeeping (add / ntotal / remove_ids) works without a GPU: added rows are queued on
    the host and uploaded in one copy when the device matrix is first needed.  search() has no CPU
    implementation -- without a GPU it raises."""

    def __init__(self, d, device=None):
        self.d = int(d)
        self._device_arg = device
        self.ld = (self.d + 3) // 4 * 4
        self._n = 0                  # rows resident in the device matrix
        self._store = None
        self._pending = []           # host row blocks not yet uploaded
        self._npending = 0
        self._ws = None
        self._stats = None
        self._prepared = None        # (planes, norms) of the resident rows for the batched search; dropped when rows change
        self._searches_since_change = 0

    def _rows_changed(self):
        """the resident rows changed in a way the prepared plane cannot follow (compaction, adoption of another matrix, reset)"""
        self._prepared = None
        self._searches_since_change = 0

    def _rows_written(self, n_old, row0, nrows):
        """rows [row0, row0 + nrows) were overwritten / appended (n_old -> self._n rows): the prepared plane follows
        incrementally (`ac_knn_update_store`: those rows only) unless the new rows change the store's scale -- an index that is
        both searched and added to keeps its fp16 plane instead of paying two passes over all rows per change (round 3 dropped
        the plane on every change)."""
        if self._prepared is None:
            return
        planes, norms = self._prepared
        need_p, need_n = store_buffers_sizes(self._n, self.d)
        if planes.numel() < need_p or norms.numel() < need_n:            # grow: the first n_old rows' entries are a prefix
            cap = max(self._store.shape[0], self._n)
            planes2, norms2 = store_buffers(cap, self.d, self.device)
            old_p, old_n = store_buffers_sizes(n_old, self.d)
            planes2[:old_p] = planes[:old_p]
            norms2[:old_n] = norms[:old_n]
            self._prepared = (planes2, norms2)
        if not update_store(self._store, n_old, self._n, self.d, self._prepared, row0, nrows):
            self._prepared = None                                         # scale changed: prepared anew at the next search

    @property
    def device(self):
        if self._device_arg is not None:
            return torch.device(self._device_arg)
        nv.require_gpu()
        return torch.device(f"cuda:{torch.cuda.current_device()}")

    # -- faiss protocol ------------------------------------------------------------------
    @property
    def ntotal(self):
        return self._n + self._npending

    def _as_rows(self, x):
        if isinstance(x, torch.Tensor):
            t = x.detach().to(dtype=torch.float32)
        else:
            t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        return t.reshape(-1, self.d)

    def _reserve(self, n):
        dev = self.device
        if self._store is not None and n <= self._store.shape[0]:
            return
        have = 0 if self._store is None else self._store.shape[0]
        cap = max(n, 2 * have, 64)
        new = torch.zeros((cap, self.ld), dtype=torch.float32, device=dev)
        if self._n:
            new[: self._n] = self._store[: self._n]
        self._store = new

    def _materialize(self):
        """Upload queued rows (one H2D copy)."""
        nv.require_gpu()
        if self._pending:
            rows = torch.cat(self._pending, 0) if len(self._pending) > 1 else self._pending[0]
            self._pending, self._npending = [], 0
            m = rows.shape[0]
            self._reserve(self._n + m)
            self._store[self._n: self._n + m, : self.d] = rows.to(self.device, non_blocking=True)
            self._n += m
            self._rows_written(self._n - m, self._n - m, m) if self._n > m else self._rows_changed()
        if self._store is None:
            self._reserve(1)
        if self._stats is None:
            self._stats = torch.zeros(4, dtype=torch.int32, device=self.device)

    def add(self, x):
        rows = self._as_rows(x)
        if rows.shape[0] == 0:
            return
        if rows.is_cuda:
            self._materialize()
            m = rows.shape[0]
            self._reserve(self._n + m)
            self._store[self._n: self._n + m, : self.d] = rows.to(self.device)
            self._n += m
            self._rows_written(self._n - m, self._n - m, m) if self._n > m else self._rows_changed()
        else:
            self._pending.append(rows.clone())
            self._npending += rows.shape[0]

    def add_device_rows(self, rows):
        """Adopt an existing [n, ld] device matrix without copying (large synthetic stores)."""
        assert rows.is_cuda and rows.dtype == torch.float32 and rows.stride(1) == 1
        assert rows.stride(0) % 4 == 0 and rows.stride(0) >= self.ld and rows.data_ptr() % 16 == 0
        self._device_arg = rows.device
        self._pending, self._npending = [], 0
        self._store = rows
        self._n = rows.shape[0]
        self._rows_changed()

    def remove_ids(self, ids):
        if isinstance(ids, torch.Tensor):
            ids = ids.detach().cpu().numpy()
        ids = np.unique(np.asarray(ids).reshape(-1).astype(np.int64))
        ids = ids[(ids >= 0) & (ids < self.ntotal)]
        if ids.size == 0:
            return 0
        self._materialize()
        keep = torch.ones(self._n, dtype=torch.bool, device=self.device)
        keep[torch.from_numpy(ids).to(self.device)] = False
        kept = self._store[: self._n][keep]             # IndexFlat compacts: later rows shift down
        self._store[: kept.shape[0]] = kept
        self._n = kept.shape[0]
        self._rows_changed()
        return int(ids.size)

    def reset(self):
        self._n = 0
        self._pending, self._npending = [], 0
        self._rows_changed()

    def update_rows(self, rows, values):
        """Overwrite existing rows in place (ids keep their meaning; no compaction)."""
        rows = torch.as_tensor(rows, dtype=torch.int64)
        if rows.numel() == 0:
            return
        if int(rows.max()) >= self.ntotal or int(rows.min()) < 0:
            raise IndexError("update_rows: row id out of range")
        self._materialize()
        vals = self._as_rows(values).to(self.device)
        # (the runs are computed BEFORE the store is touched: nothing below can fail between the write and the plane's update)
        runs = None
        if self._prepared is not None:
            ids = np.unique(rows.detach().cpu().numpy())
            runs = np.split(ids, np.nonzero(np.diff(ids) != 1)[0] + 1)       # contiguous runs of row ids
        self._store[rows.to(self.device), : self.d] = vals
        if runs is not None:
            if len(runs) > 16:
                self._rows_changed()                                      # many scattered rows: a fresh preparation is cheaper
            else:
                try:
                    for run in runs:
                        if self._prepared is None:
                            break
                        self._rows_written(self._n, int(run[0]), int(run.size))
                except Exception:
                    self._rows_changed()                                  # never leave a plane that no longer matches the rows
                    raise

    def search_device(self, q, k):
        """q: [nq, d] fp32 tensor (any device) -> (dist, ids) CUDA tensors; no host sync."""
        self._materialize()
        q = q.detach().to(device=self.device, dtype=torch.float32)
        if q.dim() == 1:
            q = q.unsqueeze(0)
        if q.stride(-1) != 1:
            q = q.contiguous()
        batch = batch_applies(self._n, q.shape[0], k, auto=True)
        if batch and self._prepared is None:
            # preparing costs two passes over the rows (~0.14 s at 10M x 768): at once for a many-query search (it pays within
            # the call); for a small batch only when the store has already served a search since it was last rebuilt / compacted
            # (appends and in-place updates do not count: once prepared, the plane follows them incrementally)
            if q.shape[0] >= BATCH_MIN_QUERIES or self._searches_since_change >= 1:
                # (sized for the store's CAPACITY: the first append after the preparation then updates the plane in place
                #  instead of allocating capacity-sized buffers and copying the whole old plane beside them)
                self._prepared = prepare_store(self._store, self._n, self.d, capacity=self._store.shape[0])
            else:
                batch = False
        self._searches_since_change += 1
        need = knn_batch_workspace_bytes(self._n, self.d, q.shape[0], k) if batch else knn_workspace_bytes(self._n, self.d, q.shape[0], k)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 256), dtype=torch.uint8, device=self.device)
        return knn_l2_topk(self._store, self._n, self.d, q, k, workspace=self._ws, stats=self._stats,
                           prepared=self._prepared if batch else None)

    def search(self, x, k):
        """faiss signature: numpy in, (float32 [nq,k], int64 [nq,k]) numpy out."""
        q = self._as_rows(x)
        D, I = self.search_device(q, int(k))
        return D.cpu().numpy(), I.cpu().numpy()

    @property
    def exact_fallbacks(self):
        """Queries of the last search() that needed the exact fp64 fallback sweep."""
        return 0 if self._stats is None else int(self._stats[0].item())


def topk_merge(D_in, I_in):
    """[shards, nq, k] per-shard ascending lists -> global (dist fp32, ids) [nq, k].  float64 input (the shards' exact
    distances, `knn_l2_topk_exact`) is merged by (exact distance, id) -- ac_topk_merge_f64, what a sharded search needs to
    equal the unsharded one bit for bit; float32 input by (fp32 distance, id) -- ac_topk_merge."""
    nv.require_gpu()
    S, nq, k = D_in.shape
    D_in = D_in.contiguous()
    I_in = I_in.contiguous()
    outD = torch.empty((nq, k), dtype=torch.float32, device=D_in.device)
    outI = torch.empty((nq, k), dtype=torch.int64, device=D_in.device)
    fn, name = ((nv.lib().ac_topk_merge_f64, "ac_topk_merge_f64") if D_in.dtype == torch.float64
                else (nv.lib().ac_topk_merge, "ac_topk_merge"))
    assert D_in.dtype in (torch.float32, torch.float64)
    with torch.cuda.device(D_in.device):
        nv.check(fn(nv.ptr(D_in), nv.ptr(I_in), S, nq, k, nv.ptr(outD), nv.ptr(outI), nv.stream_ptr(D_in.device)), name)
    return outD, outI


def proto_scores(D, I):
    """memory.py:117,129-130 on device: softmax(exp(-d)) over each query's valid hits."""
    nv.require_gpu()
    D = D.contiguous()
    I = I.contiguous()
    out = torch.empty_like(D)
    with torch.cuda.device(D.device):
        nv.check(nv.lib().ac_proto_scores(nv.ptr(D), nv.ptr(I), D.shape[0], D.shape[1], nv.ptr(out),
                                          nv.stream_ptr(D.device)), "ac_proto_scores")
    return out


def synth_unit_rows(n, D, seed, row_offset=0, device=None, ld=None):
    """Deterministic unit-norm rows generated on the device (bit-identical to oracle/synth.py)."""
    nv.require_gpu()
    device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    ld = ld or (D + 3) // 4 * 4
    out = torch.empty((n, ld), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        nv.check(nv.lib().ac_synth_unit_rows(nv.ptr(out), n, ld, D, seed, row_offset, nv.stream_ptr(device)),
                 "ac_synth_unit_rows")
    return out
