"""CPU suite: the multi-GPU orchestration (row shards, query all-gather, top-k all-gather + merge)
with world_size-2 / 4 / 8 gloo groups (uneven row shards, uneven and empty query blocks).  The HIP kernels cannot run here, so the local search and the merge
are the ORACLE (injected) -- what is under test is the sharding / collective / offset logic of
adaptive_classifier.sharded, which is identical under RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, N, D, sizes, k, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "adaptive-classifier_amd")]
    from adaptive_classifier.sharded import ShardedSearch, shard_bounds
    from oracle import knn_oracle, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(N, world, rank)
    rows = torch.from_numpy(synth.synth_unit_rows(hi - lo, D, 1, row_offset=lo)) if hi > lo else torch.zeros((0, D))
    q0 = sum(sizes[:rank])
    q_local = torch.from_numpy(synth.synth_unit_rows(sizes[rank], D, 2, row_offset=q0)) if sizes[rank] else torch.zeros((0, D))

    def local_search(P, n, Dd, Q, kk, off):
        d, i = knn_oracle.knn_l2_topk(P.numpy()[:n], Q.numpy(), kk, row_offset=off, return_exact=True)   # fp64 on the wire
        return torch.from_numpy(d), torch.from_numpy(i)

    def merge(Ds, Is):
        d, i = knn_oracle.topk_merge(Ds.numpy(), Is.numpy(), Ds.shape[2])
        return torch.from_numpy(d), torch.from_numpy(i)

    ss = ShardedSearch(rows, hi - lo, D, lo, local_search=local_search, merge=merge)
    Q = ss.gather_queries(q_local)                         # sizes agreed by the ranks themselves
    Dg, Ig = ss.search(Q, k)
    Db, Ib = ss.search_block(q_local, k)                   # all_to_all: this rank merges only its own query block
    Db2, Ib2 = ss.search_block(q_local, k, block_sizes=sizes)      # sizes known by construction: no size exchange
    wrong = None
    try:                                                   # a list that contradicts this rank's block: loud, before any collective
        bad = list(sizes); bad[rank] += 1
        ss.search_block(q_local, k, block_sizes=bad)
    except ValueError as e:
        wrong = str(e)
    ret[rank] = (Q.numpy(), Dg.numpy(), Ig.numpy(), Db.numpy(), Ib.numpy(), Db2.numpy(), Ib2.numpy(), wrong)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, N, k, sizes, D=64):
    from oracle import knn_oracle, synth
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), N, D, sizes, k, ret), nprocs=world, join=True)
    nq = sum(sizes)
    P = synth.synth_unit_rows(N, D, 1)
    Q = synth.synth_unit_rows(nq, D, 2)
    oD, oI = knn_oracle.knn_l2_topk(P, Q, k)
    for r in range(world):
        Qr, Dg, Ig, Db, Ib, Db2, Ib2, wrong = ret[r]
        lo = sum(sizes[:r])
        assert np.array_equal(Qr, Q)                       # gathered query block is the global batch, rank order
        assert np.array_equal(Ig, oI) and np.array_equal(Dg, oD)
        assert Ib.shape == (sizes[r], k)
        assert np.array_equal(Ib, oI[lo:lo + sizes[r]]) and np.array_equal(Db, oD[lo:lo + sizes[r]])
        assert np.array_equal(Ib2, Ib) and np.array_equal(Db2, Db)
        assert wrong is not None and "block_sizes" in wrong


@pytest.mark.parametrize("N,k", [(1001, 8), (37, 16)])
def test_sharded_search_world2(N, k):
    _run(2, N, k, [3, 3])


@pytest.mark.parametrize("world,N,k,sizes", [
    (4, 1003, 8, [2, 2, 2, 2]),                # equal query blocks, uneven row shards (251, 251, 251, 250)
    (4, 1003, 8, [3, 1, 0, 2]),                # uneven last batch incl. an empty block
    (8, 1003, 8, [2, 2, 2, 2, 2, 2, 2, 1]),    # world 8, ragged tail
    (8, 70, 8, [1] * 8),                       # shards of 9 / 8 rows: k = 8 fills from every shard
])
def test_sharded_search_world_4_8_uneven(world, N, k, sizes):
    _run(world, N, k, sizes)


def test_shard_bounds_cover_rows():
    from adaptive_classifier.sharded import shard_bounds
    for N in (0, 1, 7, 8, 1000, 10_000_000):
        for w in (1, 2, 4, 8):
            b = [shard_bounds(N, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == N and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def _worker_fixed(rank, world, port, N, D, batches, br, k, ret):
    """The fixed-batch path: blocks padded to `br`, pre-allocated messages, no size exchange, the query all-gather of batch i + 1
    issued before batch i is searched (search_blocks)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "adaptive-classifier_amd")]
    from adaptive_classifier.sharded import ShardedSearch, shard_bounds
    from oracle import knn_oracle, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(N, world, rank)
    rows = torch.from_numpy(synth.synth_unit_rows(hi - lo, D, 1, row_offset=lo)) if hi > lo else torch.zeros((0, D))

    def local_search(P, n, Dd, Q, kk, off):
        d, i = knn_oracle.knn_l2_topk(P.numpy()[:n], Q.numpy(), kk, row_offset=off, return_exact=True)
        return torch.from_numpy(d), torch.from_numpy(i)

    def merge(Ds, Is):
        d, i = knn_oracle.topk_merge(Ds.numpy(), Is.numpy(), Ds.shape[2])
        return torch.from_numpy(d), torch.from_numpy(i)

    ss = ShardedSearch(rows, hi - lo, D, lo, local_search=local_search, merge=merge, block_rows=br)
    blocks, q0 = [], 0
    for sizes in batches:                                   # batch b: rank r owns sizes[r] queries of the global query stream
        mine0 = q0 + sum(sizes[:rank])
        blocks.append(torch.from_numpy(synth.synth_unit_rows(sizes[rank], D, 2, row_offset=mine0)) if sizes[rank] else torch.zeros((0, D)))
        q0 += sum(sizes)
    outs = [(d.numpy(), i.numpy()) for d, i in ss.search_blocks(blocks, k)]
    allocs_after_loop = ss.stats["buffer_allocations"]
    one = ss.search_block(blocks[0], k)                     # the un-pipelined call of the same path: same buffers, same result
    too_big = None
    try:
        ss.search_block(torch.zeros((br + 1, D)), k)
    except ValueError as e:
        too_big = str(e)
    ret[rank] = (outs, one[0].numpy(), one[1].numpy(), dict(ss.stats), allocs_after_loop, too_big)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,batches,br", [
    (2, [[4, 4], [4, 4], [4, 2]], 4),                 # a fixed per-rank batch with a short last one
    (4, [[3, 3, 3, 3], [3, 0, 1, 3], [2, 2, 2, 2]], 3),   # an empty and a short block in the middle of the stream
])
def test_fixed_batch_path_pipelined_no_size_exchange(world, batches, br):
    from oracle import knn_oracle, synth
    N, D, k = 1003, 64, 8
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_fixed, args=(world, _free_port(), N, D, batches, br, k, ret), nprocs=world, join=True)
    P = synth.synth_unit_rows(N, D, 1)
    nq = sum(sum(b) for b in batches)
    Q = synth.synth_unit_rows(nq, D, 2)
    oD, oI = knn_oracle.knn_l2_topk(P, Q, k)
    for r in range(world):
        outs, d1, i1, stats, allocs_after_loop, too_big = ret[r]
        q0 = 0
        for b, sizes in enumerate(batches):
            lo = q0 + sum(sizes[:r])
            d, i = outs[b]
            assert i.shape == (sizes[r], k)
            assert np.array_equal(i, oI[lo:lo + sizes[r]]) and np.array_equal(d, oD[lo:lo + sizes[r]])
            q0 += sum(sizes)
        lo = sum(batches[0][:r])
        assert np.array_equal(i1, oI[lo:lo + batches[0][r]]) and np.array_equal(d1, oD[lo:lo + batches[0][r]])
        assert stats["size_exchanges"] == 0                                   # nothing agreed through the host
        assert stats["prefetched_gathers"] == len(batches) + 1
        assert stats["buffer_allocations"] == allocs_after_loop <= 6          # two query pads, two gathered blocks, one send, one receive
        assert too_big is not None and "block_rows" in too_big
