"""Randomised parity sweep (one-off confidence run, also usable as a soak): random small shapes -- model, k, d, feature width and
dtype, item count on both sides of 65 536, batch sizes around the 16-triplet plane granule, optimizer, sampler (epoch walk with
byte planes / i.i.d. Philox / caller-made batches), segment or list or atomic mode (bf16 / fp32 features; fp8 has its own tests) -- three steps each on the engine and on the
CPU oracle, tables and losses compared.  python scripts/fuzz_parity.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fashionvisualexpl_recommend_amd import synth                      # noqa: E402
from fashionvisualexpl_recommend_amd.engine import Engine, EpochWalkSampler, PhiloxSampler   # noqa: E402
from oracle import oracle as orc                                      # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    model = rs.choice(["bprmf", "vbpr"])
    k = int(rs.choice([4, 8, 16, 32, 64, 128, 256]))
    d = int(rs.choice([4, 8, 20, 64, 128, 256])) if model == "vbpr" else 0
    D = int(rs.choice([128, 256, 512])) if d else 0
    dtype = str(rs.choice(["bf16", "fp32"])) if d else "fp32"          # (fp8: its own tests -- outlier-fraction tolerances)
    I = int(rs.choice([37, 300, 1000, 5000, 70000, 140000]))
    if d and I > 5000:
        I = int(rs.choice([300, 1000, 5000]))                           # (keep the oracle's projections quick)
    U = int(rs.choice([50, 400, 3000]))
    B = int(rs.choice([16, 48, 250, 256, 1000, 1024, 2048]))
    opt = str(rs.choice(["sgd", "sgd", "adam_tf23"]))
    mode = str(rs.choice(["auto", "seg", "noseg"]))
    smp_kind = str(rs.choice(["epoch", "philox", "caller"]))
    os.environ.pop("BPRX_ITEM_MODE", None)
    if mode == "seg":
        os.environ["BPRX_ITEM_MODE"] = "2"
    elif mode == "noseg":
        os.environ["BPRX_ITEM_MODE"] = "0"
    tag = "case %d: %s k=%d d=%d D=%d %s I=%d U=%d B=%d %s mode=%s sampler=%s" % (case, model, k, d, D, dtype, I, U, B, opt, mode, smp_kind)
    r2 = np.random.RandomState(1000 + case)
    t = dict(Gu=synth.glorot_uniform(r2, U, k), Gi=synth.glorot_uniform(r2, I, k), Bi=(r2.standard_normal(I) * 0.01).astype(np.float32))
    quant = 0
    kw = {}
    if d:
        F = synth.make_features(I, D, seed=case)
        F = (F / np.abs(F).max()).astype(np.float32)
        if dtype == "bf16":
            F = orc.bf16_round(F); quant = 1
        t.update(Tu=synth.glorot_uniform(r2, U, d), F=F, E=synth.glorot_uniform(r2, D, d), Bp=synth.glorot_uniform(r2, D, 1).reshape(-1))
        kw = dict(embed_d=d, feat_dim=D, feat_dtype=dtype)
    lr, reg = (0.05, 1e-3) if opt == "sgd" else (0.01, 1e-3)
    try:
        e = Engine(model=model, num_users=U, num_items=I, embed_k=k, optimizer=opt, lr=lr, reg=reg, max_batch=B, **kw).bind(**t)
        o = orc.OracleModel(**t, quant=quant)
        t0 = {n: np.array(v, dtype=np.float32).reshape(-1).copy() for n, v in t.items() if n != "F"}
        per = int(min(I - 1, rs.choice([3, 9, 20])))
        lists = [sorted(r2.choice(I, size=per, replace=False).tolist()) for _ in range(U)]
        smp = None
        if smp_kind == "epoch":
            smp = EpochWalkSampler(lists, I, seed=case).feeds(e)
        elif smp_kind == "philox":
            smp = PhiloxSampler(lists, I, seed=case).feeds(e)
        for step in range(3):
            if smp is not None:
                u, i, j = smp.sample(B)
            else:
                u, i, j = (torch.as_tensor(r2.randint(n, size=B).astype(np.int32), device="cuda") for n in (U, I, I))
            loss = e.step(u, i, j).item()
            want = o.step(u.cpu().numpy(), i.cpu().numpy(), j.cpu().numpy(), opt, lr, reg)
            rt, at = (2e-5, 2e-6) if not d or dtype == "fp32" else (3e-3, 2e-4)
            if opt != "sgd":
                at = max(at, 2e-3 * lr)
            assert abs(loss - want) <= 2e-4 * abs(want) + 1e-5, ("loss", step, loss, want)
            for n in (("Gu", "Gi", "Bi", "Tu", "E", "Bp") if d else ("Gu", "Gi", "Bi")):
                got, ref = e.t[n].cpu().numpy().reshape(-1), getattr(o, n).reshape(-1)
                at_n = at
                if quant == 1:
                    # bf16 W: an item's gradient sum that lands next to a bf16 rounding boundary rounds either way depending on
                    # the (unordered) summation order -- one flipped entry is 2^-9 of that item's share of dE / dBp; the table
                    # tolerance therefore scales with how far the table has moved (aggressive random shapes move Bp by 0.4 a step)
                    at_n = max(at, 2e-3 * float(np.abs(ref - t0[n]).max()))
                if opt == "sgd":
                    np.testing.assert_allclose(got, ref, rtol=rt, atol=at_n, err_msg="%s step %d" % (n, step))
                else:
                    # adam_tf23: m / (sqrt(v) + eps) is sign-like for an element whose gradient is rounding noise around zero -- a
                    # last-bit difference there moves the element by up to lr per step in either direction (TensorFlow's Adam does the
                    # same; tests/test_gpu_parity.py::test_bprmf_steps_match_oracle).  A handful of elements may exceed the
                    # tolerance, none by more than what such sign flips explain.
                    diff = np.abs(got - ref)
                    over = diff > at_n + rt * np.abs(ref)
                    assert over.sum() <= max(2, 5e-5 * over.size) and diff.max() <= 2.5 * lr * (step + 1), (n, step, int(over.sum()), float(diff.max()))
        e.sync_check()
        print("ok  ", tag, "index kind", e.lib.bprx_index_pass_kind(e.h))
    except AssertionError as ex:
        bad += 1
        print("FAIL", tag, str(ex)[:300])
print("%d cases, %d failed" % (n_cases, bad))
sys.exit(1 if bad else 0)
