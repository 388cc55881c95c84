#!/usr/bin/env python3
"""Random shapes: occurrence-segment path (BPRX_ITEM_MODE=2) against the atomic staging path (0), sgd and adam."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fashionvisualexpl_recommend_amd import synth
from fashionvisualexpl_recommend_amd.engine import Engine

rs = np.random.RandomState(int(os.environ.get("FUZZ_SEED", "1")))
bad = 0
for trial in range(int(os.environ.get("FUZZ_N", "40"))):
    model = rs.choice(["bprmf", "vbpr"])
    k = int(rs.choice([4, 8, 16, 32, 64, 128, 200]))
    d, D = (int(rs.choice([4, 20, 64, 128])), int(rs.choice([128, 256, 512]))) if model == "vbpr" else (0, 0)
    U, I = int(rs.randint(3, 400)), int(rs.randint(3, 400))
    B = int(rs.choice([1, 2, 7, 64, 65, 255, 256, 1000, 3000]))
    opt = rs.choice(["sgd", "adam_tf23"])
    dtype = rs.choice(["bf16", "fp32"]) if model == "vbpr" else "fp32"
    t = dict(Gu=synth.glorot_uniform(rs, U, k), Gi=synth.glorot_uniform(rs, I, k), Bi=(rs.standard_normal(I) * 0.01).astype(np.float32))
    if d:
        F = np.abs(rs.standard_normal((I, D))).astype(np.float32); F /= F.max()
        t.update(Tu=synth.glorot_uniform(rs, U, d), F=F, E=synth.glorot_uniform(rs, D, d), Bp=synth.glorot_uniform(rs, D, 1).reshape(-1))
    u = rs.randint(U, size=B).astype(np.int32); i = rs.randint(I, size=B).astype(np.int32); j = rs.randint(I, size=B).astype(np.int32)
    if B > 100:
        i[:B // 3] = i[0]; j[B // 3: B // 2] = i[0]           # a hot item, both roles
    res = []
    for mode in (("2", "2") if os.environ.get("FUZZ_SELF") == "1" else ("0", "2")):   # FUZZ_SELF=1: run-to-run spread of one mode
        os.environ["BPRX_ITEM_MODE"] = mode
        kw = dict(embed_d=d, feat_dim=D, feat_dtype=dtype) if d else {}
        e = Engine(model=model, num_users=U, num_items=I, embed_k=k, optimizer=opt, lr=0.05 if opt == "sgd" else 0.01, reg=1e-3,
                   max_batch=B, **kw).bind(**{n: v.copy() for n, v in t.items()})
        dev = lambda a: torch.as_tensor(a, device="cuda")
        losses = [e.step(dev(u), dev(i), dev(j)).item() for _ in range(2)]
        e.sync_check()
        res.append((losses, {n: e.t[n].float().cpu().numpy().copy() for n in e.params()}))
        e.close()
    (l0, t0), (l1, t1) = res
    ok = np.allclose(l0, l1, rtol=2e-4)
    # (B > 100 plants a hot item with B/3 + B/6 occurrences: the two modes sum those fp32 terms in different orders)
    tol = 5e-3 if (dtype == "bf16" or opt != "sgd") else 2e-5 * max(1.0, B / 150.0)
    for n in t0:
        diff = np.abs(t0[n] - t1[n]).max()
        if diff > tol * max(1.0, np.abs(t0[n]).max()):
            ok = False
    if not ok:
        bad += 1
        print("MISMATCH", trial, model, k, d, D, U, I, B, opt, dtype, l0, l1, {n: float(np.abs(t0[n] - t1[n]).max()) for n in t0}, flush=True)
print("fuzz done: %d mismatches" % bad)
