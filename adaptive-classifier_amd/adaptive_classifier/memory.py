"""PrototypeMemory -- drop-in for /root/reference/src/adaptive_classifier/memory.py:11-295 with the
prototype store resident in MI355X HBM and the kNN on the HIP sweep kernel.

Surface kept (asserted by the reference's tests/test_memory.py):
  PrototypeMemory(embedding_dim, config) | add_example | get_nearest_prototypes | _update_prototype |
  _rebuild_index | _restore_from_save | _prune_examples | get_stats | clear
  attributes: examples (defaultdict label -> [Example]), prototypes (label -> CPU tensor [D]),
  index, label_to_index, index_to_label, updates_since_rebuild, embedding_dim, config.
The host dict/list mirrors stay authoritative (the classifier reads and mutates them directly);
the device matrix behind `.index` is a cache refreshed lazily.

Deliberate differences from the reference (DESIGN.md "quirks"):
  * `_update_prototype` rewrites the prototype's row in place instead of faiss remove_ids+add
    (memory.py:156-159), which in the reference leaves label<->row maps stale until the next rebuild.
  * class means come from an fp64 running sum (O(D) per add instead of re-stacking every stored
    example, memory.py:149-150); the mean differs from torch.mean's fp32 result by <= 1 ulp.
Beyond the reference: `load_rows()` puts an arbitrary [N, D] device matrix with an
int32 row->class map behind the same search API (N >> #classes, BASELINE configs 1-2, 4).
"""
import ctypes
import logging
import operator
import threading
from collections import defaultdict
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _native as nv
from .index import HipFlatL2Index, proto_scores
from .models import Example, ModelConfig

logger = logging.getLogger(__name__)


_EMBEDDING_OF = operator.attrgetter("embedding")


class PrototypeMemory:
    """Per-class example store + class-mean prototypes + exact L2 kNN over the prototypes."""

    def __init__(self, embedding_dim: int, config: Optional[ModelConfig] = None, device=None):
        self.embedding_dim = embedding_dim
        self.config = config or ModelConfig()
        self.examples = defaultdict(list)
        self.prototypes = {}
        self.strategic_prototypes = {}
        self._device = device
        self.index = self._new_index()
        self.label_to_index = {}
        self.index_to_label = {}
        self.updates_since_rebuild = 0
        self._sums = {}                      # label -> (fp64 running sum of the stored embeddings, count)
        self._mats = {}                      # label -> [host matrix of the stored embeddings, count]
        self._fps = {}                       # label -> (count, identity fingerprint) the two caches above describe
        self._fp_refs = {}                   # label -> the fingerprinted embedding tensors (strong references: ids stay unique)
        self._dm_refs = {}
        self._dmats = {}                     # label -> [DEVICE matrix of the stored embeddings, count, fingerprint]: input of the
                                             # device prune and of the training-set assembly (no per-call re-upload)
        self._dirty = set()                  # labels whose index row is out of date
        self._lock = threading.RLock()       # add_example is called from threads (test_memory.py:226-256)
        self._row_labels = None              # int32 device tensor when load_rows() is in use
        self._row_label_names = None

    def _new_index(self):
        return HipFlatL2Index(self.embedding_dim, device=self._device)

    # ------------------------------------------------------------------ mirror validity
    # `examples` is a public attribute that callers (the classifier, the reference's tests) edit directly:
    # lists are assigned, entries replaced, embeddings swapped.  The fp64 class sums (`_sums`) and host class
    # matrices (`_mats`) are caches of those lists; they are trusted only while the list still holds the very same
    # embedding tensors in the same order (identity fingerprint), not merely the same NUMBER of entries.
    @staticmethod
    def _fingerprint(exs, n):
        # (C-level loops: this runs a few times per class and call over up to max_examples_per_class entries)
        return hash(tuple(map(id, map(_EMBEDDING_OF, exs if n == len(exs) else exs[:n]))))

    def _stamp(self, label):
        exs = self.examples[label]
        self._fps[label] = (len(exs), self._fingerprint(exs, len(exs)))
        # the fingerprint is made of object ids: hold the fingerprinted tensors so that CPython cannot hand one of their ids
        # to a replacement embedding while the stamp is still trusted (A -> B -> C with C at A's address)
        self._fp_refs[label] = tuple(map(_EMBEDDING_OF, exs))

    def _mirror_valid(self, label, n):
        """True iff the caches of `label` describe exactly examples[label][:n]."""
        fp = self._fps.get(label)
        return fp is not None and fp[0] == n and fp[1] == self._fingerprint(self.examples[label], n)

    def device_class_matrix(self, label, dev, room=0, fp=None):
        """[rows >= n + room, D] fp32 matrix on `dev` whose first n = len(examples[label]) rows are the class's stored
        embeddings in list order.  Kept in step by add_examples_batch (appends / device prune write it in place); rebuilt
        from the list whenever the list was edited from outside (identity fingerprint)."""
        exs = self.examples[label]
        n = len(exs)
        dev = torch.device(dev)
        if dev.type == "cuda" and dev.index is None:      # "cuda" names the current device; tensors carry "cuda:<index>"
            dev = torch.device("cuda", torch.cuda.current_device())
        ent = self._dmats.get(label)
        if fp is None:                      # (callers that just computed the list's fingerprint pass it in)
            fp = self._fingerprint(exs, n)
        if ent is None or ent[0].device != dev or ent[1] != n or ent[2] != fp:
            d = torch.empty((max(n + room, 64), self.embedding_dim), dtype=torch.float32, device=dev)
            if n:
                d[:n] = torch.stack([e.embedding.detach().to(torch.float32) for e in exs]).to(dev)
            ent = [d, n, fp]
            self._dm_refs[label] = tuple(map(_EMBEDDING_OF, exs))     # (same reason as _fp_refs)
        elif ent[0].shape[0] < n + room:
            d = torch.empty((max(n + room, 2 * ent[0].shape[0]), self.embedding_dim), dtype=torch.float32, device=dev)
            d[:n] = ent[0][:n]
            ent[0] = d
        self._dmats[label] = ent
        return ent

    # ------------------------------------------------------------------ add / prune / prototype
    def _class_matrix(self, label, n_needed):
        """[rows >= n_needed, D] host matrix mirroring self.examples[label][:count]; rebuilt when the
        list was edited behind our back (the classifier assigns / deletes lists directly)."""
        ent = self._mats.get(label)
        exs = self.examples[label]
        count = len(exs) - 1                       # rows that must already be mirrored (all but the new one)
        valid = ent is not None and ent[1] == count and self._mirror_valid(label, count)
        if not valid or ent[0].shape[0] < n_needed:
            rows = max(n_needed, 2 * (ent[0].shape[0] if ent is not None else 0), 64)
            mat = torch.empty((rows, self.embedding_dim), dtype=torch.float32)
            if count:
                if valid:
                    mat[:count] = ent[0][:count]
                else:
                    mat[:count] = torch.stack([ex.embedding for ex in exs[:count]])
            ent = [mat, count]
            self._mats[label] = ent
        return ent

    def add_example(self, example: Example, label: str):
        if example.embedding is None:
            raise ValueError("Example must have an embedding")
        if example.embedding.size(-1) != self.embedding_dim:
            raise ValueError(
                f"Example embedding dimension {example.embedding.size(-1)} "
                f"does not match memory dimension {self.embedding_dim}")
        with self._lock:
            if example.embedding.is_cuda:
                example.embedding = example.embedding.detach().cpu()   # prototypes stay on the host
            self.examples[label].append(example)
            n = len(self.examples[label])
            trusted = self._mirror_valid(label, n - 1)
            ent = self._class_matrix(label, n)
            ent[0][n - 1] = example.embedding
            ent[1] = n
            cached = self._sums.get(label)
            if cached is None or cached[1] != n - 1 or not trusted:   # first add, or the list was edited behind us
                s = ent[0][: n - 1].double().sum(0)
            else:
                s = cached[0]
            self._sums[label] = (s + example.embedding.detach().double(), n)
            self._stamp(label)
            if n > self.config.max_examples_per_class:
                self._prune_examples(label)
            self._update_prototype(label)
            # counter / lazy rebuild logic of memory.py:70-81
            if not getattr(self, "just_rebuilt", False):
                self.updates_since_rebuild += 1
            if self.updates_since_rebuild >= self.config.prototype_update_frequency:
                self._rebuild_index()
                self.just_rebuilt = True
            else:
                self.just_rebuilt = False

    def add_examples_batch(self, examples: List[Example], labels: List[str]):
        """`add_example` for every (example, label) in order, with the same final state, but with the
        per-class add -> mean -> drop-farthest loop of memory.py:41-83 / :196-217 run ON THE DEVICE for all the
        examples a class receives (`ac_memory_add_prune`, one launch per class that overflows its cap) and the
        Python lists / class matrices / prototypes updated once per class instead of once per example.
        Falls back to `add_example` per example when there is no GPU or the class is beyond the kernel's
        on-chip working set."""
        if len(examples) != len(labels):
            raise ValueError("Mismatched example and label lists")
        for ex in examples:                               # same validation, before anything is mutated
            if ex.embedding is None:
                raise ValueError("Example must have an embedding")
            if ex.embedding.size(-1) != self.embedding_dim:
                raise ValueError(f"Example embedding dimension {ex.embedding.size(-1)} "
                                 f"does not match memory dimension {self.embedding_dim}")
        with self._lock:
            cap = int(self.config.max_examples_per_class)
            D = self.embedding_dim
            try:
                dev = self.index.device
            except (nv.NativeError, AttributeError):       # no GPU (CPU-side tests): per-example host logic below
                dev = None
            groups: Dict[str, List[Example]] = {}
            jobs = []                                      # classes that overflow their cap in this call
            for ex, label in zip(examples, labels):
                if ex.embedding.is_cuda:
                    ex.embedding = ex.embedding.detach().cpu()
                groups.setdefault(label, []).append(ex)
            for label, new in groups.items():
                lst = self.examples[label]
                n0, k = len(lst), len(new)
                on_chip = n0 + k <= 8192 and D <= 4096 and (n0 + k) * 9 + D * 12 + 16 <= 150 * 1024
                if n0 + k > cap and (dev is None or dev.type != "cuda" or n0 > cap or not on_chip):
                    for ex in new:                        # per-example host logic (counters handled below)
                        self._add_one_no_counters(ex, label)
                    continue
                fresh = torch.stack([e.embedding.detach().to(torch.float32) for e in new])
                if dev is not None and dev.type == "cuda":
                    # device-resident class matrix: only the k new rows cross PCIe
                    fp0 = self._fingerprint(lst, n0)        # one identity pass over the stored list per class and call
                    trusted = self._fps.get(label) == (n0, fp0)
                    dm = self.device_class_matrix(label, dev, room=k, fp=fp0)
                    cached = self._sums.get(label)
                    total = (cached[0] if cached is not None and cached[1] == n0 and trusted
                             else dm[0][:n0].double().sum(0).cpu())
                    dm[0][n0:n0 + k] = fresh.to(dev, non_blocking=True)
                    self._mats.pop(label, None)            # the host mirror is not maintained on this path
                    if n0 + k <= cap:
                        lst.extend(new)
                        self._sums[label] = (total + fresh.double().sum(0), n0 + k)
                        self._stamp(label)
                        dm[1], dm[2] = n0 + k, self._fps[label][1]
                        self._dm_refs[label] = self._fp_refs[label]
                        self._update_prototype(label, trusted=True)
                    else:
                        jobs.append((label, lst, new, dm, None, total, n0, k))
                    continue
                ent = self._mats.get(label)
                trusted = self._mirror_valid(label, n0)
                if ent is None or ent[1] != n0 or not trusted:   # (re)build the mirror of the stored rows
                    rows0 = max(n0 + k + 1, 64)
                    mat = torch.empty((rows0, D), dtype=torch.float32)
                    if n0:
                        mat[:n0] = torch.stack([e.embedding for e in lst])
                    ent = [mat, n0]
                cached = self._sums.get(label)
                total = (cached[0] if cached is not None and cached[1] == n0 and trusted
                         else ent[0][:n0].double().sum(0))
                if n0 + k <= cap:
                    # nothing to prune: O(k) appends (list, class matrix with geometric growth, fp64 sum)
                    if ent[0].shape[0] < n0 + k:
                        mat = torch.empty((max(n0 + k, 2 * ent[0].shape[0]), D), dtype=torch.float32)
                        mat[:n0] = ent[0][:n0]
                        ent = [mat, n0]
                    ent[0][n0:n0 + k] = fresh
                    ent[1] = n0 + k
                    lst.extend(new)
                    self._mats[label] = ent
                    self._sums[label] = (total + fresh.double().sum(0), n0 + k)
                    self._stamp(label)
                    self._update_prototype(label)
                    continue
                raise AssertionError("unreachable: an overflowing class without a GPU takes the per-example path above")
            if jobs:
                self._run_prune_jobs(jobs, cap, D, dev)
            # the counter / lazy-rebuild state machine of memory.py:70-81, once per example
            rebuild = False
            for _ in examples:
                if not getattr(self, "just_rebuilt", False):
                    self.updates_since_rebuild += 1
                if self.updates_since_rebuild >= self.config.prototype_update_frequency:
                    self.updates_since_rebuild = 0
                    rebuild = True
                    self.just_rebuilt = True
                else:
                    self.just_rebuilt = False
            if rebuild:
                pending = self.updates_since_rebuild       # _rebuild_index zeroes the counter
                self._rebuild_index()
                self.updates_since_rebuild = pending

    def _run_prune_jobs(self, jobs, cap, D, dev):
        """One `ac_memory_add_prune` launch for all overflowing classes of a call, reading the classes' DEVICE matrices
        in place (old rows + the just-written new rows), then the update of each class: list in ascending-distance order,
        device matrix compacted in that order, fp64 sum, prototype."""
        nj = len(jobs)
        ntot = sum(j[6] + j[7] for j in jobs)
        d_sum = torch.stack([j[5] for j in jobs]).to(dev).contiguous()         # [nj, D] fp64
        d_alive = torch.empty(ntot, dtype=torch.uint8, device=dev)
        d_dist = torch.empty(ntot, dtype=torch.float64, device=dev)
        d_drop = torch.empty(sum(j[7] for j in jobs), dtype=torch.int32, device=dev)
        arr = (nv.ac_prune_job * nj)()
        off = doff = 0
        for i, (_, _, _, dm, _, _, n0, k) in enumerate(jobs):
            arr[i].rows = dm[0].data_ptr()
            arr[i].ld = dm[0].stride(0)
            arr[i].n_old, arr[i].n_new, arr[i].cap, arr[i].reserved = n0, k, cap, 0
            arr[i].sum = d_sum.data_ptr() + i * D * 8
            arr[i].alive = d_alive.data_ptr() + off
            arr[i].dist = d_dist.data_ptr() + off * 8
            arr[i].dropped = d_drop.data_ptr() + doff * 4
            off += n0 + k
            doff += k
        raw = bytes(arr)
        d_jobs = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        with torch.cuda.device(dev):
            nv.check(nv.lib().ac_memory_add_prune(ctypes.cast(arr, ctypes.c_void_p), nv.ptr(d_jobs), nj, D,
                                                  nv.stream_ptr(dev)), "ac_memory_add_prune")
        alive_all = d_alive.cpu().numpy().astype(bool)
        dist_all = d_dist.cpu().numpy()
        sums = d_sum.cpu()
        off = 0
        for i, (label, lst, new, dm, _, _, n0, k) in enumerate(jobs):
            alive, dist = alive_all[off: off + n0 + k], dist_all[off: off + n0 + k]
            off += n0 + k
            kept = np.nonzero(alive)[0]
            # the last prune leaves the list in ascending distance to the then-current mean (:212-214)
            order = kept[np.argsort(dist[kept], kind="stable")]
            everything = lst + new
            self.examples[label] = [everything[i_] for i_ in order.tolist()]
            n = len(order)
            sel = torch.from_numpy(np.ascontiguousarray(order)).to(dev)
            dm[0][:n] = dm[0][sel]                                   # (advanced indexing materialises the gather first)
            self._sums[label] = (sums[i].clone(), n)
            self._stamp(label)
            dm[1], dm[2] = n, self._fps[label][1]
            self._dm_refs[label] = self._fp_refs[label]
            self._update_prototype(label, trusted=True)

    def _add_one_no_counters(self, example: Example, label: str):
        """add_example minus the rebuild counters (used by add_examples_batch's host fallback)."""
        self.examples[label].append(example)
        n = len(self.examples[label])
        trusted = self._mirror_valid(label, n - 1)
        ent = self._class_matrix(label, n)
        ent[0][n - 1] = example.embedding
        ent[1] = n
        cached = self._sums.get(label)
        s = ent[0][: n - 1].double().sum(0) if cached is None or cached[1] != n - 1 or not trusted else cached[0]
        self._sums[label] = (s + example.embedding.detach().double(), n)
        self._stamp(label)
        if n > self.config.max_examples_per_class:
            self._prune_examples(label)
        self._update_prototype(label)

    def _update_prototype(self, label: str, trusted: bool = False):
        """trusted: the caller stamped the list a moment ago (same critical section): skip the identity pass."""
        examples = self.examples[label]
        if not examples:
            return
        cached = self._sums.get(label)
        if cached is None or cached[1] != len(examples) or not (trusted or self._mirror_valid(label, len(examples))):
            # the list was assigned / edited by the caller: recompute from the list, as the reference always does
            mat = torch.stack([ex.embedding.detach().to(torch.float32) for ex in examples])
            cached = (mat.double().sum(0), len(examples))
            self._sums[label] = cached
            self._mats[label] = [mat, len(examples)]
            self._stamp(label)
        self.prototypes[label] = (cached[0] / len(examples)).to(torch.float32)
        if label in self.label_to_index:
            self._dirty.add(label)           # row refreshed in place at the next search

    def _prune_examples(self, label: str):
        """Keep the max_examples_per_class examples closest to the class mean, in ascending-distance
        order like the reference (memory.py:196-217).  One vectorised pass over the class matrix instead of
        re-stacking every stored embedding."""
        examples = self.examples[label]
        if not examples:
            return
        n = len(examples)
        ent = self._mats.get(label)
        trusted = self._mirror_valid(label, n)
        if ent is not None and ent[1] == n and trusted:
            emb = ent[0][:n]
        else:
            emb = torch.stack([ex.embedding for ex in examples])
        cached = self._sums.get(label)
        total = cached[0] if cached is not None and cached[1] == n and trusted else emb.double().sum(0)
        mean = (total / n).to(torch.float32)
        dist = torch.linalg.vector_norm(emb - mean, dim=1).numpy()
        order = np.argsort(dist)
        keep = order[: self.config.max_examples_per_class]
        self.examples[label] = [examples[i] for i in keep]
        kept = emb[torch.from_numpy(np.ascontiguousarray(keep))]
        dropped = order[self.config.max_examples_per_class:]
        if len(dropped) <= 8:                       # the usual case: one over the cap
            new_sum = total - emb[torch.from_numpy(np.ascontiguousarray(dropped))].double().sum(0)
        else:
            new_sum = kept.double().sum(0)
        self._sums[label] = (new_sum, len(keep))
        if ent is not None and ent[0].shape[0] >= len(keep):
            ent[0][: len(keep)] = kept
            ent[1] = len(keep)
            self._mats[label] = ent
        else:
            self._mats.pop(label, None)
        self._stamp(label)
        assert len(self.examples[label]) <= self.config.max_examples_per_class

    # ------------------------------------------------------------------ index maintenance
    def _rebuild_index(self):
        """Dense [C, D] matrix in sorted-label order, one upload (memory.py:161-177)."""
        with self._lock:
            self.index = self._new_index()
            self.label_to_index.clear()
            self.index_to_label.clear()
            labels = sorted(self.prototypes.keys())
            if labels:
                self.index.add(torch.stack([self.prototypes[l].detach().float().cpu() for l in labels]))
            for i, label in enumerate(labels):
                self.label_to_index[label] = i
                self.index_to_label[i] = label
            self._dirty.clear()
            self._row_labels = None
            self._row_label_names = None
            self.updates_since_rebuild = 0

    def _restore_from_save(self):
        """memory.py:179-194: same as a rebuild (prototypes were assigned directly by the loader)."""
        self._rebuild_index()

    def _flush_dirty(self):
        if self._dirty:
            labels = [l for l in self._dirty if l in self.label_to_index]
            if labels:
                rows = torch.tensor([self.label_to_index[l] for l in labels], dtype=torch.int64)
                vals = torch.stack([self.prototypes[l].float() for l in labels])
                self.index.update_rows(rows, vals)
            self._dirty.clear()

    def load_rows(self, rows: torch.Tensor, row_labels: torch.Tensor, label_names: List[str], sharded=None,
                  prepare: bool = False):
        """Generalised store: arbitrary device rows + int32 row->class map behind the same search.
        prepare=True builds the store's fp16 plane + row norms now (`index.prepare_store`: two passes over the rows, 2 B per
        element) instead of at the first many-query / second few-query search, so that every search of a static store --
        the first included -- proposes from the plane (half the bytes of the fp32 sweep).

        With `sharded` (adaptive_classifier.sharded.ShardedSearch) `rows` is this rank's row shard and
        `row_labels` the GLOBAL (replicated) row->class map; search_batch() then all-gathers the ranks'
        query blocks, searches the local shard and merges the all-gathered per-shard top-k."""
        self.index = self._new_index()
        self.index.add_device_rows(rows)
        self._row_labels = row_labels.to(device=rows.device, dtype=torch.int32).contiguous()
        self._row_label_names = list(label_names)
        self._sharded = sharded
        self._row_class_cache = None
        self.updates_since_rebuild = 0
        self._dirty.clear()
        if prepare and sharded is None:
            from .index import BATCH_MIN_ROWS, prepare_store
            if self.index.ntotal >= BATCH_MIN_ROWS:
                self.index._prepared = prepare_store(self.index._store, self.index.ntotal, self.index.d)
                self.index._searches_since_change = 1

    # ------------------------------------------------------------------ search
    def get_nearest_prototypes(self, query_embedding: torch.Tensor, k: int = 5,
                               min_similarity: Optional[float] = None) -> List[Tuple[str, float]]:
        """[(label, score)] ascending in distance; scores = softmax(exp(-d^2)) (memory.py:85-136)."""
        with self._lock:
            if self._row_labels is None and self.updates_since_rebuild >= self.config.prototype_update_frequency:
                self._rebuild_index()
            if self.index.ntotal == 0:
                return []
            self._flush_dirty()
            k = min(k, self.index.ntotal)
            D, I = self.index.search_device(query_embedding.detach().reshape(1, -1), k)
            S = proto_scores(D, I)
            ids = I[0].cpu().numpy()
            scores = S[0].cpu().numpy()
        results = []
        for idx, score in zip(ids, scores):
            if idx >= 0:
                results.append((self._label_of_row(int(idx)), float(score)))
        return results

    def _label_of_row(self, idx):
        if self._row_labels is not None:
            return self._row_label_names[int(self._row_labels[idx].item())]
        return self.index_to_label[idx]

    def search_batch(self, queries: torch.Tensor, k: int):
        """Device-resident batch search: (scores [b,k], row ids [b,k], dist [b,k]) CUDA tensors.

        Same arithmetic as get_nearest_prototypes per row; no host synchronisation."""
        with self._lock:
            if self._row_labels is None and self.updates_since_rebuild >= self.config.prototype_update_frequency:
                self._rebuild_index()
            self._flush_dirty()
            D, I = self._search_locked(queries, k)
            return proto_scores(D, I), I, D

    def search_raw(self, queries: torch.Tensor, k: int):
        """search_batch without the score kernel: (dist [b,k] f32, row ids [b,k] i64) on the device -- the inputs of
        ac_predict_post, which derives the scores itself."""
        with self._lock:
            if self._row_labels is None and self.updates_since_rebuild >= self.config.prototype_update_frequency:
                self._rebuild_index()
            self._flush_dirty()
            return self._search_locked(queries, k)

    def _search_locked(self, queries, k):
        sharded = getattr(self, "_sharded", None)
        if sharded is not None and (sharded.world > 1 or sharded.force_collectives):
            return sharded.search_block(queries, k)           # this rank's queries against every rank's row shard
        k = min(k, max(self.index.ntotal, 1))
        return self.index.search_device(queries, k)

    def class_map(self, label_to_id: Dict[str, int], dev):
        """(row_class int32 tensor or None, nrows, lut int64 tensor, nlut): hit row id -> classifier class id, the arguments
        of ac_rows_to_class / ac_predict_post (None row_class: row id == index of the label in index_to_label)."""
        if self._row_labels is not None:
            names, row_class, nrows = self._row_label_names, self._row_labels, int(self._row_labels.numel())
        else:
            n = self.index.ntotal
            names, row_class, nrows = [self.index_to_label[i] for i in range(n)], None, n
        key = tuple(label_to_id.get(nm, -1) for nm in names)
        cached = getattr(self, "_class_lut_cache", None)
        if cached is None or cached[0] != key or cached[1].device != dev:
            cached = (key, torch.tensor(key if key else (-1,), dtype=torch.int64, device=dev))
            self._class_lut_cache = cached
        return row_class, nrows, cached[1], len(key)

    def hit_class_ids(self, I: torch.Tensor, label_to_id: Dict[str, int]) -> torch.Tensor:
        """Classifier class id of every hit row id in I [b, k] (-1 = padding / unknown label), on device,
        one native launch (`ac_rows_to_class`)."""
        dev = I.device
        row_class, nrows, lut, nlut = self.class_map(label_to_id, dev)
        I = I.contiguous()
        out = torch.empty_like(I)
        with torch.cuda.device(dev):
            nv.check(nv.lib().ac_rows_to_class(nv.ptr(I), I.numel(), nv.ptr(row_class), nrows, nv.ptr(lut),
                                               nlut, nv.ptr(out), nv.stream_ptr(dev)), "ac_rows_to_class")
        return out

    # ------------------------------------------------------------------ misc API
    def get_stats(self) -> Dict[str, Any]:
        return {
            "num_classes": len(self.prototypes),
            "examples_per_class": {label: len(ex) for label, ex in self.examples.items()},
            "total_examples": sum(len(ex) for ex in self.examples.values()),
            "prototype_dimensions": self.embedding_dim,
            "updates_since_rebuild": self.updates_since_rebuild,
        }

    def clear(self):
        with self._lock:
            self.examples.clear()
            self.prototypes.clear()
            self._sums.clear()
            self._mats.clear()
            self._dmats.clear()
            self._fps.clear()
            self._fp_refs.clear()
            self._dm_refs.clear()
            self._dirty.clear()
            self.index = self._new_index()
            self.label_to_index.clear()
            self.index_to_label.clear()
            self._row_labels = None
            self._row_label_names = None
            self.updates_since_rebuild = 0

    def drop_label(self, label):
        """Forget cached sums when the classifier deletes a label's examples (classifier.py:1396-1399)."""
        self._sums.pop(label, None)
        self._mats.pop(label, None)
        self._dmats.pop(label, None)
        self._fps.pop(label, None)
        self._fp_refs.pop(label, None)
        self._dm_refs.pop(label, None)
        self._dirty.discard(label)
