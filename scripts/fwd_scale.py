#!/usr/bin/env python3
"""Forward projection time against the item count (workgroups per CU): per-CU or chip-wide limit?"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fashionvisualexpl_recommend_amd.engine import Engine  # noqa: E402

dev = torch.device("cuda", 0)
out = {}
for I in [int(x) for x in os.environ.get("ITEMS", "16384,32768,40960,49152,50000,57344,65536,98304,131072").split(",")]:
    w = dict(bench.WORKLOADS["c2"], I=I, U=1024)
    t = bench.make_state(w, dev, 1, torch)
    e = Engine(model="vbpr", num_users=w["U"], num_items=I, embed_k=w["k"], embed_d=w["d"], feat_dim=w["D"],
               feat_dtype="bf16", optimizer="sgd", lr=0.05, reg=1e-4, max_batch=1024).bind(**t)
    for _ in range(3):
        e.step_project()
    torch.cuda.synchronize()
    e.profile(True)
    for _ in range(10):
        e.step_project()
    torch.cuda.synchronize()
    p = e.profile_read()
    us = p["proj_fwd"][0] / p["proj_fwd"][1] * 1e3
    out[I] = round(us, 1)
    print(I, "items", round(us, 1), "us", round(I * 8192 / us / 1e6, 2), "TB/s of F", flush=True)
    e.close()
    del e, t
print(json.dumps(out))
