#!/usr/bin/env python3
"""Where the item-sharded step spends its time with ONE rank over RCCL (no wire traffic): host enqueue time vs GPU time."""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
# (the host-side phase timers this script used were a scratch monkey-patch, removed in round 2)
from fashionvisualexpl_recommend_amd.dist import ItemShardedVBPR
from fashionvisualexpl_recommend_amd.engine import EpochWalkSampler

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
w = dict(bench.WORKLOADS["c2"]); B = w["B"]
t = bench.make_state(w, dev, 1, torch)
sh = ItemShardedVBPR(0, 1, w["U"], t["Gu"], t["Tu"], t["Gi"], t["Bi"], t["F"], t["E"], t["Bp"], lr=1e-4, reg=1e-4, max_batch=B,
                     feat_dtype="bf16", device=0)
g = torch.Generator(device=dev); g.manual_seed(1)
npu = 20
items = torch.randint(w["I"], (w["U"], npu), generator=g, device=dev, dtype=torch.int32).sort(dim=1).values
indptr = torch.arange(w["U"] + 1, device=dev, dtype=torch.int64) * npu
pos_user = torch.arange(w["U"], device=dev, dtype=torch.int32).repeat_interleave(npu)
smp = EpochWalkSampler.from_csr(indptr, items.reshape(-1), pos_user, w["I"], seed=5)
bufs = tuple(torch.empty(B, dtype=torch.int32, device=dev) for _ in range(3))
for _ in range(5):
    sh.step(*smp.sample(B, out=bufs))
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter(); host = 0.0
for _ in range(K):
    h0 = time.perf_counter()
    sh.step(*smp.sample(B, out=bufs))
    host += time.perf_counter() - h0
torch.cuda.synchronize()
print("ms/step %.3f  host enqueue ms/step %.3f" % ((time.perf_counter() - t0) / K * 1e3, host / K * 1e3))
if hasattr(sh, "timings"):
    print({k: round(v / (K + 5) * 1e3, 3) for k, v in sh.timings.items()})
dist.destroy_process_group()
