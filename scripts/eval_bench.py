#!/usr/bin/env python3
"""Device-side evaluation throughput on the C2 shape (bprx_score_block fp32-MFMA GEMM + bprx_eval_users)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fashionvisualexpl_recommend_amd.engine import Engine  # noqa: E402

w = dict(bench.WORKLOADS["c2"])
dev = torch.device("cuda", 0)
t = bench.make_state(w, dev, 1, torch)
e = Engine(model="vbpr", num_users=w["U"], num_items=w["I"], embed_k=w["k"], embed_d=w["d"], feat_dim=w["D"],
           feat_dtype=w["dtype"], optimizer="sgd", max_batch=1024).bind(**t)
g = torch.Generator(device=dev); g.manual_seed(3)
U, I = w["U"], w["I"]
tr_items = torch.randint(I, (U, 20), generator=g, device=dev, dtype=torch.int32).reshape(-1)
tr_ptr = torch.arange(U + 1, device=dev, dtype=torch.int64) * 20
te_items = torch.randint(I, (U,), generator=g, device=dev, dtype=torch.int32)
te_ptr = torch.arange(U + 1, device=dev, dtype=torch.int64)
nblk = int(os.environ.get("EVAL_USERS", "16384"))
blk = 4096
sc = e.score_block(0, blk)
torch.cuda.synchronize()
t0 = time.perf_counter()
tg = 0.0
for u0 in range(0, nblk, blk):
    a = time.perf_counter()
    sc = e.score_block(u0, u0 + blk, out=sc)
    torch.cuda.synchronize()
    tg += time.perf_counter() - a
    r = e.eval_users(u0, u0 + blk, sc, (tr_ptr, tr_items), (te_ptr, te_items), 10)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
flops = 2.0 * nblk * I * (w["k"] + w["d"])
print(json.dumps({"users": nblk, "items": I, "seconds": dt, "users_per_s": nblk / dt, "score_gemm_s": tg,
                  "score_gemm_tflops": flops / tg / 1e12, "hr": float(r[:, 0].mean())}))
