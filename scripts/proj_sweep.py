#!/usr/bin/env python3
"""A/B sweep of kernel variants in ONE process on a bench workload (interleaved rounds, HIP-event timing)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fashionvisualexpl_recommend_amd.engine import Engine  # noqa: E402


def main():
    wl = os.environ.get("SWEEP_WORKLOAD", "c2")
    w = dict(bench.WORKLOADS[wl])
    B = int(os.environ.get("SWEEP_B", w["B"]))
    dev = torch.device("cuda", 0)
    tables = bench.make_state(w, dev, 1234, torch)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    batches = [tuple(torch.randint(n, (B,), generator=g, device=dev, dtype=torch.int32) for n in (w["U"], w["I"], w["I"]))
               for _ in range(4)]
    fwd = [int(x) for x in os.environ.get("SWEEP_FWD", "0,8,2,10,3,11").split(",")]
    bwd = [int(x) for x in os.environ.get("SWEEP_BWD", "0,8,9,12").split(",")]
    sks = [int(x) for x in os.environ.get("SWEEP_SK", "32").split(",")]
    rounds = int(os.environ.get("SWEEP_ROUNDS", "3"))
    steps = int(os.environ.get("SWEEP_STEPS", "8"))
    configs = [(f, bwd[0], sks[0]) for f in fwd] + [(fwd[0], b, s) for b in bwd for s in sks if (b, s) != (bwd[0], sks[0])]
    res = {c: {} for c in configs}
    ref = None
    for rnd in range(rounds):
        for c in configs:
            os.environ["BPRX_FWD_VARIANT"], os.environ["BPRX_BWD_VARIANT"], os.environ["BPRX_SK"] = map(str, c)
            t = {n: (v.clone() if n != "F" else v) for n, v in tables.items()}
            e = Engine(model=w["model"], num_users=w["U"], num_items=w["I"], embed_k=w["k"], embed_d=w["d"],
                       feat_dim=w["D"], feat_dtype=w["dtype"], optimizer="sgd", lr=0.05, reg=1e-4, max_batch=B).bind(**t)
            for s in range(2):
                e.step(*batches[s % 4], want_loss=False)
            torch.cuda.synchronize()
            e.profile(True)
            for s in range(steps):
                e.step(*batches[s % 4], want_loss=False)
            torch.cuda.synchronize()
            p = e.profile_read()
            e.profile(False)
            for k, (ms, n) in p.items():
                res[c].setdefault(k, []).append(ms / n)
            if rnd == 0:
                names = [n for n in ("E", "Tu", "Gu") if n in e.t]
                chk = [float(e.t[n].double().abs().sum().item()) for n in names]
                if ref is None:
                    ref = chk
                res[c]["_chk"] = [abs(a - b) / b for a, b in zip(chk, ref)]
            e.close()
            del e, t
    for c in configs:
        r = res[c]
        line = {"fwd": c[0], "bwd": c[1], "SK": c[2]}
        for k in ("proj_fwd", "proj_bwd", "reduce_parts", "triplet_grad", "apply"):
            if k in r:
                line[k] = round(float(np.median(r[k])) * 1e3, 1)       # microseconds, median over rounds
        line["chk"] = ["%.1e" % x for x in r["_chk"]]
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
