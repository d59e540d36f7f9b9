"""Row-sharded prototype search across the GPUs of one node.

  partition   prototype rows P[g*N/G : (g+1)*N/G] live on rank g (contiguous; global id = local id +
              row_offset); the head and encoder weights are replicated; query batches are data parallel.
  exchange    1. all_gather the ranks' query blocks  [b/G, D] -> [b, D]        (RCCL over xGMI)
              2. local `ac_knn_l2_topk_x` of all b queries against the rank's shard
              3. all_to_all of the per-shard (EXACT fp64 dist, id int64) lists: rank j receives, from every shard, the
                 [b/G, k] candidates of ITS OWN query block (SURVEY 8e "alternative": 1/G of the bytes an all_gather of
                 all [b, k] lists moves, and every rank merges b/G queries instead of all b)
              4. `ac_topk_merge_f64` -> global top-k by (exact distance, id) of the rank's own queries, rounded to fp32
              (fp64 on the wire: two candidates on different shards whose exact distances differ but round to the same
              fp32 value must be ordered by distance, not by id, or the sharded result differs from the unsharded one --
              measured: ~40 of 131k neighbour pairs at 10M x 768, 4096 queries)
              `search()` (every rank wants the result of ALL queries) keeps step 3 as an all_gather.
The messages are small (cfg2 at G = 8: 2.1 MB of queries in, 0.26 MB of candidates per peer), i.e. latency bound; there
is no other collective on the data path.  One process per GPU, torch.distributed backend "nccl" (= RCCL).
Query blocks may differ in size between ranks (an uneven last batch) and may be empty: the ranks first agree on the sizes
(an 8-byte all_gather, skipped when the caller passes `block_sizes` or constructs with `equal_blocks=True`), queries travel
padded to the largest block, candidates by an all_to_all with split sizes.  A `block_sizes` list that contradicts the
rank's own block raises ValueError before any collective is entered.
No multi-GPU scaling curve has been measured by the builder (one GPU per gpurun box); the RCCL calls themselves are
exercised on a world-size-1 nccl group (tests/test_sharded_gpu.py::test_rccl_branch_world1), the orchestration at world
2 / 4 / 8 over gloo on CPU (tests/test_sharded_gloo.py) and at world 2 over gloo with the real kernels on one GPU.

The local search and the merge are injected so that the orchestration (offsets, gather layout,
merge semantics) can be exercised with world_size-2 `gloo` groups on CPU in tests, where the
callables are the oracle; the product wires the HIP kernels (default).
"""
import torch
import torch.distributed as dist


def shard_bounds(N, world, rank):
    """Contiguous row range [lo, hi) of `rank` (first N % world ranks get one extra row)."""
    q, r = divmod(N, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _pack(D, I):
    """(fp64 distances [.., k], int64 ids [.., k]) -> one int64 tensor [.., 2k] (the exchanges are latency bound)"""
    return torch.cat([D.contiguous().view(torch.int64), I], dim=-1)


def _unpack(both):
    k = both.shape[-1] // 2
    return both[..., :k].contiguous().view(torch.float64), both[..., k:].contiguous()


class ShardedSearch:
    def __init__(self, local_rows, n_local, dim, row_offset, group=None, local_search=None, merge=None,
                 force_collectives=False, equal_blocks=False, block_rows=None):
        """force_collectives: run the collectives even on a one-rank group (tests: executes the RCCL calls on one GPU).
        equal_blocks: the caller guarantees that every rank passes the same number of queries to every search_block /
        gather_queries call (a fixed per-rank batch): the ranks then skip the exchange of their block sizes.
        block_rows: the FIXED-BATCH path (what a serving loop and bench.py use): every rank's query block has AT MOST this many
        rows; blocks travel padded to exactly block_rows, so no sizes are exchanged and no value is read back to the host
        between the collectives (a short or empty block only costs its padding rows of local search), every message has a
        fixed shape and lives in buffers allocated once (`search_block`), and the query all-gather of the NEXT batch can run
        under the local sweep of the current one (`prefetch_queries` / `search_blocks`)."""
        self.rows, self.n_local, self.dim, self.row_offset = local_rows, n_local, dim, row_offset
        self.group = group
        self.force_collectives = bool(force_collectives)
        self.equal_blocks = bool(equal_blocks)
        self.block_rows = None if block_rows is None else int(block_rows)
        if self.block_rows is not None and self.block_rows < 1:
            raise ValueError("block_rows must be >= 1")
        self._bufs = {}                # fixed-batch path: (name, shape, dtype) -> tensor, allocated once
        self.stats = {"size_exchanges": 0, "buffer_allocations": 0, "prefetched_gathers": 0}
        self._prepared = None          # bf16 planes + norms of the local shard (batched searches), built on first use
        if local_search is None or merge is None:
            from . import index as ix

            def _hip_search(P, n, D, Q, k, off):
                # many queries x a big shard: the prepared-store GEMM-form sweep (same exact result)
                if ix.batch_applies(n, Q.shape[0], k, auto=True):
                    if self._prepared is None:
                        self._prepared = ix.prepare_store(P, n, D)
                    return ix.knn_l2_topk_exact(P, n, D, Q, k, row_offset=off, prepared=self._prepared)
                return ix.knn_l2_topk_exact(P, n, D, Q, k, row_offset=off)

            local_search = local_search or _hip_search
            merge = merge or ix.topk_merge
        self._search, self._merge = local_search, merge
        self._ws = None

    @property
    def rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def _collective(self):
        return self.world > 1 or (self.force_collectives and dist.is_initialized())

    def _staged(self, t):
        """device tensors under a non-RCCL backend (gloo: CPU tests, two processes on one GPU) travel through the host"""
        return t.is_cuda and dist.get_backend(self.group) != "nccl"

    def _all_gather(self, t):
        """all_gather along a new leading dim -> [world, ...].  RCCL gathers device tensors directly."""
        t = t.contiguous()
        if self._staged(t):
            parts = [torch.empty_like(t, device="cpu") for _ in range(self.world)]
            dist.all_gather(parts, t.cpu(), group=self.group)
            return torch.stack(parts).to(t.device)
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.group) if t.is_cuda else \
            dist.all_gather(list(out.unbind(0)), t, group=self.group)
        return out

    def _all_to_all(self, t):
        """t [world, m, ...]: block j goes to rank j; returns [world, m, ...] with block i = what rank i sent to this rank."""
        t = t.contiguous()
        if self._staged(t):
            src = t.cpu()
            out = torch.empty_like(src)
            dist.all_to_all_single(out, src, group=self.group)
            return out.to(t.device)
        out = torch.empty_like(t)
        dist.all_to_all_single(out, t, group=self.group)
        return out

    def block_sizes(self, m):
        """Every rank's query-block size [world] (one tiny all_gather + a host read).  Callers that know the sizes by
        construction (e.g. `shard_bounds(batch, world, r)`) pass them to search_block / gather_queries and skip this."""
        if not self._collective:
            return [int(m)]
        if self.equal_blocks:
            return [int(m)] * self.world
        self.stats["size_exchanges"] += 1
        if dist.get_backend(self.group) == "nccl":
            t = torch.tensor([int(m)], dtype=torch.int64, device=self.rows.device if self.rows.is_cuda else "cuda")
            out = torch.empty(self.world, dtype=torch.int64, device=t.device)
            dist.all_gather_into_tensor(out, t, group=self.group)
            return [int(x) for x in out.tolist()]
        parts = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        dist.all_gather(parts, torch.tensor([int(m)], dtype=torch.int64), group=self.group)
        return [int(p.item()) for p in parts]

    def _check_sizes(self, sizes, m):
        sizes = [int(x) for x in sizes]
        if len(sizes) != self.world or any(x < 0 for x in sizes):
            raise ValueError(f"block_sizes must hold one non-negative size per rank ({self.world}), got {sizes}")
        if sizes[self.rank] != int(m):
            raise ValueError(f"block_sizes[{self.rank}] = {sizes[self.rank]} but this rank's query block has {int(m)} rows")
        return sizes

    def gather_queries(self, q_local, block_sizes=None):
        """Data-parallel query blocks -> the full [sum(sizes), D] block on every rank, rank order.  Blocks may differ in
        size (an uneven last batch): they travel padded to the largest one and are compacted on arrival."""
        if not self._collective:
            return q_local
        sizes = self._check_sizes(self.block_sizes(q_local.shape[0]) if block_sizes is None else block_sizes, q_local.shape[0])
        br = max(sizes)
        if br == 0:
            return q_local
        if q_local.shape[0] < br:
            pad = torch.zeros((br - q_local.shape[0], q_local.shape[1]), dtype=q_local.dtype, device=q_local.device)
            q_local = torch.cat([q_local, pad])
        g = self._all_gather(q_local)                                  # [world, br, D]
        if all(x == br for x in sizes):
            return g.reshape(-1, q_local.shape[-1])
        return torch.cat([g[r, :sizes[r]] for r in range(self.world)])

    def search(self, queries, k):
        """queries [b, D] (identical on all ranks) -> global (dist fp32 [b,k], ids [b,k]) on EVERY rank.
        The local search returns EXACT fp64 distances; they are what travels and what the merge orders by."""
        D_loc, I_loc = self._search(self.rows, self.n_local, self.dim, queries, k, self.row_offset)
        if not self._collective:
            return D_loc.to(torch.float32), I_loc
        both = self._all_gather(_pack(D_loc, I_loc))            # ONE message per peer: fp64 bits and ids side by side
        return self._merge(*_unpack(both))

    # ---- the fixed-batch path (block_rows given): fixed shapes, buffers allocated once, no host read-back ----------------
    def _buf(self, name, shape, dtype, device):
        key = (name, tuple(shape), dtype, str(device))
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = torch.empty(shape, dtype=dtype, device=device)
            self.stats["buffer_allocations"] += 1
        return t

    def prefetch_queries(self, q_local):
        """Start the all-gather of a query block (padded to block_rows) WITHOUT waiting for it: under RCCL the collective runs on
        the communicator's own stream, so it proceeds while the caller's stream is still busy with the previous batch's sweep.
        Returns a handle for `search_block(..., gathered=handle)`.  Two buffers alternate, so one batch may be in flight while
        the previous one is being searched."""
        if self.block_rows is None:
            raise ValueError("prefetch_queries needs the fixed-batch path (construct with block_rows=...)")
        m, br, G = int(q_local.shape[0]), self.block_rows, self.world
        if m > br:
            raise ValueError(f"query block of {m} rows exceeds block_rows = {br}")
        slot = self.stats["prefetched_gathers"] & 1
        self.stats["prefetched_gathers"] += 1
        dev = q_local.device
        pad = self._buf(f"qpad{slot}", (br, self.dim), q_local.dtype, dev)
        pad[:m].copy_(q_local)
        if m < br:
            pad[m:].zero_()
        if not self._collective:
            return {"m": m, "all": pad, "work": None}
        if self._staged(pad):
            parts = [torch.empty((br, self.dim), dtype=pad.dtype) for _ in range(G)]
            dist.all_gather(parts, pad.cpu(), group=self.group)
            out = self._buf(f"qall{slot}", (G, br, self.dim), pad.dtype, dev)
            out.copy_(torch.stack(parts))
            return {"m": m, "all": out.view(G * br, self.dim), "work": None}
        out = self._buf(f"qall{slot}", (G, br, self.dim), pad.dtype, dev)
        if pad.is_cuda:
            work = dist.all_gather_into_tensor(out, pad, group=self.group, async_op=True)
        else:
            work = dist.all_gather(list(out.unbind(0)), pad, group=self.group, async_op=True)
        return {"m": m, "all": out.view(G * br, self.dim), "work": work}

    def _search_block_fixed(self, q_local, k, gathered=None):
        h = gathered if gathered is not None else self.prefetch_queries(q_local)
        m, br, G = h["m"], self.block_rows, self.world
        if h["work"] is not None:
            h["work"].wait()                       # (RCCL: the caller's STREAM waits for the collective; the host does not)
        D_loc, I_loc = self._search(self.rows, self.n_local, self.dim, h["all"], k, self.row_offset)
        if not self._collective:
            return D_loc[:m].to(torch.float32), I_loc[:m]
        dev = D_loc.device
        send = self._buf("cand_send", (G, br, 2 * k), torch.int64, dev)
        send.view(G * br, 2 * k)[:, :k].copy_(D_loc.contiguous().view(torch.int64))
        send.view(G * br, 2 * k)[:, k:].copy_(I_loc)
        recv = self._buf("cand_recv", (G, br, 2 * k), torch.int64, dev)
        if self._staged(send):
            src = send.cpu()
            out = torch.empty_like(src)
            dist.all_to_all_single(out, src, group=self.group)
            recv.copy_(out)
        else:
            dist.all_to_all_single(recv, send, group=self.group)
        if m == 0:
            return torch.empty((0, k), dtype=torch.float32, device=dev), torch.empty((0, k), dtype=torch.int64, device=dev)
        return self._merge(*_unpack(recv[:, :m]))

    def search_blocks(self, blocks, k):
        """The pipelined serving loop over an iterable of query blocks (fixed-batch path): while batch i is swept locally, the
        query all-gather of batch i + 1 is already running.  Yields (dist fp32 [m_i, k], ids [m_i, k]) per block."""
        it = iter(blocks)
        try:
            cur = self.prefetch_queries(next(it))
        except StopIteration:
            return
        for nxt in it:
            ahead = self.prefetch_queries(nxt)          # (issued BEFORE the current block's sweep is enqueued)
            yield self._search_block_fixed(None, k, gathered=cur)
            cur = ahead
        yield self._search_block_fixed(None, k, gathered=cur)

    def search_block(self, q_local, k, block_sizes=None, gathered=None):
        """The data-parallel step: this rank's query block [m, D] -> the global (dist fp32 [m, k], ids [m, k]) of THOSE
        queries.  all_gather(queries) -> local search of all of them -> all_to_all of the candidate lists -> this rank
        merges only its own block.  With `block_rows` (constructor) this is the fixed-batch path: blocks padded to block_rows,
        pre-allocated messages, nothing read back to the host; `gathered` = a handle of prefetch_queries() for this block.
        Otherwise -- block_sizes: every rank's m when the caller knows them (validated against this rank's
        block; a wrong list raises here instead of hanging in the collective); None = the ranks exchange their sizes first
        (one extra 8-byte all_gather).  Sizes may differ and may be 0."""
        if self.block_rows is not None and block_sizes is None:
            return self._search_block_fixed(q_local, k, gathered)     # fixed shapes: no size exchange, no host read-back
        if not self._collective:
            D_loc, I_loc = self._search(self.rows, self.n_local, self.dim, q_local, k, self.row_offset)
            return D_loc.to(torch.float32), I_loc
        G, m = self.world, q_local.shape[0]
        sizes = self._check_sizes(self.block_sizes(m) if block_sizes is None else block_sizes, m)
        total = sum(sizes)
        dev = q_local.device
        if total == 0:
            return torch.empty((0, k), dtype=torch.float32, device=dev), torch.empty((0, k), dtype=torch.int64, device=dev)
        q_all = self.gather_queries(q_local, sizes)
        D_loc, I_loc = self._search(self.rows, self.n_local, self.dim, q_all, k, self.row_offset)
        both = _pack(D_loc, I_loc)                                 # [total, 2k] int64: one message per peer
        if all(x == m for x in sizes):
            both = self._all_to_all(both.reshape(G, m, both.shape[1]))   # -> [shard, own query, 2k]
        else:
            both = self._all_to_all_v(both, sizes, m).reshape(G, m, both.shape[1])
        if m == 0:
            return torch.empty((0, k), dtype=torch.float32, device=dev), torch.empty((0, k), dtype=torch.int64, device=dev)
        return self._merge(*_unpack(both))

    def _all_to_all_v(self, t, in_sizes, m):
        """t [sum(in_sizes), w]: rows of block j go to rank j; returns [world * m, w] (every peer sends this rank's m rows)."""
        t = t.contiguous()
        staged = self._staged(t)
        src = t.cpu() if staged else t
        out = torch.empty((self.world * m, t.shape[1]), dtype=t.dtype, device=src.device)
        dist.all_to_all_single(out, src, output_split_sizes=[m] * self.world, input_split_sizes=list(in_sizes), group=self.group)
        return out.to(t.device) if staged else out
