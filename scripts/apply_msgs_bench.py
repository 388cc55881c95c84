#!/usr/bin/env python3
"""GPU cost of the replicated-user step's message handling for N ranks (no wire): pack + apply of N messages."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fashionvisualexpl_recommend_amd.dist import ReplicatedUserVBPR
from fashionvisualexpl_recommend_amd.engine import EpochWalkSampler

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
w = dict(bench.WORKLOADS["c2"]); B = w["B"]
for N in (1, 2, 4, 8):
    t = bench.make_state(w, dev, 1, torch)
    U = w["U"] * N
    g = torch.Generator(device=dev); g.manual_seed(1)
    glo = lambda r, c: (torch.rand((r, c), generator=g, device=dev) * 2 - 1) * (6.0 / (r + c)) ** 0.5
    cap = B // 20 + 256
    m = ReplicatedUserVBPR(0, 1, glo(U, w["k"]), glo(U, w["d"]), t["Gi"], t["Bi"], t["F"], t["E"], t["Bp"], lr=1e-4, reg=1e-4,
                           max_batch=B, user_cap=cap, feat_dtype="bf16", device=0)
    items = torch.randint(w["I"], (U, 20), generator=g, device=dev, dtype=torch.int32).sort(dim=1).values
    indptr = torch.arange(U + 1, device=dev, dtype=torch.int64) * 20
    pos_user = torch.arange(U, device=dev, dtype=torch.int32).repeat_interleave(20)
    smp = EpochWalkSampler.from_csr(indptr, items.reshape(-1), pos_user, w["I"], seed=5)
    bufs = tuple(torch.empty(B, dtype=torch.int32, device=dev) for _ in range(3))
    msgsN = torch.zeros(N * m.msg.numel(), dtype=torch.float32, device=dev)

    def step():
        u, i, j = smp.sample(B, out=bufs)
        m.eng.step_begin(u, i, j)
        m.eng.pack_user_msg(u, m.cap, m.msg)
        for r in range(N):                                   # stand-in for the all-gather: N copies of the own message
            msgsN[r * m.msg.numel():(r + 1) * m.msg.numel()].copy_(m.msg)
        m.eng.apply_user_msgs(msgsN, N, m.cap, -m.lr / N)
        m.eng.step_end(want_loss=False)
    for _ in range(5):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    print("N=%d  ms/step %.3f  (message %.2f MB per rank)" % (N, (time.perf_counter() - t0) / 30 * 1e3, m.msg.numel() * 4 / 1e6), flush=True)
    del m, t
