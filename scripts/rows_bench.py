"""Micro-benchmark of the row-list forward projection (k_proj_fwd_rows) through bprx_score_pairs: HIP-event time of the
proj_fwd phase for n distinct rows of the C2 feature table.  Env knobs: BPRX_ROWS_MT, BPRX_ROWS_W16, BPRX_ROWS_DBG."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fashionvisualexpl_recommend_amd.engine import Engine

w = dict(bench.WORKLOADS[os.environ.get("WL", "c2")])
dev = torch.device("cuda", 0)
t = bench.make_state(w, dev, 1, torch)
for n in [int(x) for x in sys.argv[1:]] or [512, 2048, 8192]:
    e = Engine(model="vbpr", num_users=w["U"], num_items=w["I"], embed_k=w["k"], embed_d=w["d"], feat_dim=w["D"],
               feat_dtype=w["dtype"], optimizer="sgd", max_batch=max(n, 16)).bind(**t)
    g = torch.Generator(device=dev); g.manual_seed(n)
    items = torch.randperm(w["I"], generator=g, device=dev)[:n].to(torch.int32)
    users = torch.zeros(n, dtype=torch.int32, device=dev)
    for _ in range(5):
        e.score_pairs(users, items)
    e.profile(True)
    for _ in range(50):
        e.score_pairs(users, items)
    torch.cuda.synchronize()
    p = e.profile_read()
    print("rows %6d  proj_fwd %.2f us (incl. ~6 us event overhead)" % (n, p["proj_fwd"][0] / p["proj_fwd"][1] * 1e3), flush=True)
    e.close()
