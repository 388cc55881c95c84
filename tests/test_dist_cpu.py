"""N > 1 control flow on CPU: world_size-2 gloo.  The user-row routing of item-sharded VBPR (dist.UserRowExchange) is
pure tensor + collective logic, so it is exercised here without a GPU; the kernels it feeds are covered by
tests/test_gpu_dist.py (two ranks on one GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _exchange_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fashionvisualexpl_recommend_amd.dist import UserRowExchange, shard_size
        U, k = 37, 6                                               # 37 users over 2 ranks: shards of 19 and 18
        ush = shard_size(U, world)
        table = torch.arange(U * k, dtype=torch.float32).reshape(U, k)          # the GLOBAL table, same on all ranks
        shard = table[rank * ush:min(U, (rank + 1) * ush)].clone()
        x = UserRowExchange(rank, world, U)
        rs = np.random.RandomState(10 + rank)
        u = torch.as_tensor(rs.randint(U, size=50 + 7 * rank))                  # ragged batch sizes, duplicates
        u[:5] = 36                                                              # several hits on one remote/local row
        order, sc, rc, ridx = x.plan(u)
        assert sum(sc) == u.numel() and sum(rc) == ridx.numel()
        (rows,) = x.fetch([shard], ridx, sc, rc)
        assert torch.equal(rows, table[u[order]])                               # every requested row arrived, in order
        # send back "gradients" = ones; owners accumulate -> each owned row ends up with minus its global use count
        (back,) = x.give_back([torch.ones_like(rows)], sc, rc)
        acc = torch.zeros_like(shard)
        acc.index_add_(0, ridx.long(), back, alpha=-1.0)
        counts = [torch.zeros(U, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(counts, torch.bincount(u, minlength=U))
        total = sum(counts)[rank * ush:min(U, (rank + 1) * ush)]
        assert torch.equal(acc[:, 0], -total.float())
        # the dense all-reduce of the shared parameters' gradient
        g = torch.full((8,), float(rank + 1))
        dist.all_reduce(g)
        assert torch.equal(g, torch.full((8,), float(sum(range(1, world + 1)))))
        # empty batch on one rank must not dead-lock the collectives
        order, sc, rc, ridx = x.plan(u[:0] if rank == 0 else u)
        (rows,) = x.fetch([shard], ridx, sc, rc)
        assert rows.shape[0] == (0 if rank == 0 else u.numel())
    finally:
        dist.destroy_process_group()


def test_user_row_exchange_world2_gloo():
    mp.spawn(_exchange_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_shard_size():
    from fashionvisualexpl_recommend_amd.dist import shard_size
    assert shard_size(100, 8) == 13 and shard_size(8, 8) == 1 and shard_size(5_000_000, 8) == 625_000


def _ingest_worker(rank, world, port, path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fashionvisualexpl_recommend_amd.sharded import item_range, load_feature_shard
        full = np.load(path)
        I = full.shape[0]
        lo, hi = item_range(I, rank, world)
        part, gmax = load_feature_shard(path, lo, hi)
        want = full / np.max(np.abs(full))                     # visual_loader_mixin.py:30: ONE global scalar
        assert gmax == float(np.max(np.abs(full)))
        np.testing.assert_array_equal(part, want[lo:hi].astype(np.float32))
        assert part.dtype == np.float32 and part.shape == (hi - lo, full.shape[1])
    finally:
        dist.destroy_process_group()


def test_sharded_feature_ingestion_world2_gloo(tmp_path):
    """every rank reads only its item rows of cnn_features_*.npy; the reference's global max-abs
    normalisation becomes max-abs per shard + all-reduce(MAX).  The largest value sits in rank 1's shard."""
    rs = np.random.RandomState(0)
    f = np.abs(rs.standard_normal((37, 16)))                   # float64, like np.empty in the reference's extractor
    f[30, 3] = 9.5                                             # the global maximum is NOT in rank 0's rows
    path = str(tmp_path / "cnn_features_vgg19_fc2.npy")
    np.save(path, f)
    mp.spawn(_ingest_worker, args=(2, _free_port(), path), nprocs=2, join=True)


def test_item_range_and_local_lists():
    from fashionvisualexpl_recommend_amd.sharded import item_range, local_positive_lists
    assert [item_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert item_range(8, 3, 8) == (3, 4) and item_range(5, 7, 8) == (5, 5)
    lists = local_positive_lists([[9, 0, 4, 5], [3], []], 4, 3, 6)
    assert lists == [[1, 2], [0], [], []]                       # shard-local ids, sorted; missing users -> empty


def _fixed_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fashionvisualexpl_recommend_amd.dist import UserRowExchange, shard_size
        U, k, cap = 37, 6, 40
        ush = shard_size(U, world)
        table = torch.arange(U * k, dtype=torch.float32).reshape(U, k) + 1.0
        shard = table[rank * ush:min(U, (rank + 1) * ush)].clone()
        x = UserRowExchange(rank, world, U)
        rs = np.random.RandomState(20 + rank)
        u = torch.as_tensor(rs.randint(U, size=50 + 7 * rank))                  # ragged batches, duplicates
        u[:5] = 36
        order, slot, valid, ridx = x.plan_fixed(u, cap)
        assert bool(valid.all()) and ridx.numel() == world * cap
        (rows,) = x.fetch_fixed([shard], ridx, slot, valid)
        assert torch.equal(rows, table[u[order]])                               # every requested row arrived, in order
        (back,) = x.give_back_fixed([torch.ones_like(rows)], slot, valid, cap)
        acc = torch.zeros_like(shard)
        ok = ridx >= 0
        acc.index_add_(0, ridx[ok].long(), back[ok], alpha=-1.0)
        assert float(back[~ok].abs().sum()) == 0.0                              # empty slots carry nothing
        counts = [torch.zeros(U, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(counts, torch.bincount(u, minlength=U))
        total = sum(counts)[rank * ush:min(U, (rank + 1) * ush)]
        assert torch.equal(acc[:, 0], -total.float())
        assert not x.overflowed()
        # a bucket larger than the capacity is reported, the rest of the step stays well-formed
        order, slot, valid, ridx = x.plan_fixed(torch.zeros(30, dtype=torch.int64), 8)
        (rows,) = x.fetch_fixed([shard], ridx, slot, valid)
        assert int(valid.sum()) == 8 and torch.equal(rows[:8], table[:1].expand(8, k)) and float(rows[8:].abs().sum()) == 0.0
        assert x.overflowed() and not x.overflowed()
    finally:
        dist.destroy_process_group()


def test_fixed_capacity_row_exchange_world2_gloo():
    """The all-to-all exchange with equal, fixed-capacity splits: no split sizes are read back to the host."""
    mp.spawn(_fixed_worker, args=(2, _free_port()), nprocs=2, join=True)
