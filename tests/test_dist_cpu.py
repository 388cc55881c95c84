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
