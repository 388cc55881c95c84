"""bprx_step replayed as hipGraphs (BPRX_GRAPH=1 / 2; measured slower than plain launches on MI355X, hence opt-in): a
training loop that reuses its index buffers must give the tables the plain
launches give.  The captured launch sequences depend on host-side state that alternates from step to step (list cursors,
fp8 absmax slot, validity of the derived images): covered here by loops long enough to alternate, by reads between steps
(bprx_score_block changes what the next step has to recompute) and by outside writes (bprx_tables_dirty).
Both engines run the same kernels on the same inputs; what differs is the order of fp32 atomics: 2e-5 relative."""
import os

import numpy as np
import pytest
import torch

from fashionvisualexpl_recommend_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _make(graph, **kw):
    from fashionvisualexpl_recommend_amd.engine import Engine
    old = os.environ.get("BPRX_GRAPH")
    os.environ["BPRX_GRAPH"] = str(graph)                  # read at bprx_create
    try:
        return Engine(**kw)
    finally:
        if old is None:
            del os.environ["BPRX_GRAPH"]
        else:
            os.environ["BPRX_GRAPH"] = old


def _tables(model, U, I, k, d, D, dtype, seed=3):
    rs = np.random.RandomState(seed)
    t = dict(Gu=synth.glorot_uniform(rs, U, k), Gi=synth.glorot_uniform(rs, I, k), Bi=(rs.standard_normal(I) * 0.01).astype(np.float32))
    if model == "vbpr":
        F = synth.make_features(I, D, seed=seed)
        F = (F / np.abs(F).max()).astype(np.float32)
        if dtype == "bf16":
            F = orc.bf16_round(F)
        elif dtype == "fp8":
            F = orc.e4m3_round(F * np.float32(448.0)) / np.float32(448.0)
        t.update(Tu=synth.glorot_uniform(rs, U, d), F=F, E=synth.glorot_uniform(rs, D, d), Bp=synth.glorot_uniform(rs, D, 1).reshape(-1))
    return t


CASES = [
    # model, dtype, I, B, graph mode of the engine under test
    ("vbpr", "bf16", 6000, 256, 2),        # list mode (2B < I), small-steps policy
    ("vbpr", "fp32", 3000, 300, 2),        # list mode, fp32 features (W left dirty / re-zeroed)
    ("vbpr", "fp8", 6000, 256, 2),         # list mode, fp8: the absmax slot alternates as well
    ("vbpr", "bf16", 400, 512, 1),         # streaming form (2B >= I: occurrence segments), graphs forced
    ("vbpr", "fp8", 400, 512, 1),
    ("bprmf", "fp32", 5000, 512, 2),       # BPRMF, exclusive-row fast path + shared-row list (alternating cursor)
]


@pytest.mark.parametrize("model,dtype,I,B,gmode", CASES)
def test_graph_replay_equals_plain_launches(model, dtype, I, B, gmode):
    U, k, d, D = 900, 16, 12, 256
    t = _tables(model, U, I, k, d, D, dtype)
    kw = dict(model=model, num_users=U, num_items=I, embed_k=k, optimizer="sgd", lr=0.05, reg=1e-3, max_batch=B, device=0)
    if model == "vbpr":
        kw.update(embed_d=d, feat_dim=D, feat_dtype=dtype)
    c = lambda a: torch.as_tensor(a.copy())
    eg = _make(gmode, **kw).bind(**{n: c(v) for n, v in t.items()})
    ep = _make(0, **kw).bind(**{n: c(v) for n, v in t.items()})
    st = torch.cuda.Stream()                               # a capturable (non-default) stream
    bufs = [torch.zeros(B, dtype=torch.int32, device="cuda") for _ in range(3)]
    with torch.cuda.stream(st):
        for step in range(9):
            rs = np.random.RandomState(100 + step)
            u, i, j = rs.randint(U, size=B), rs.randint(I, size=B), rs.randint(I, size=B)
            u[:5] = 7; i[8:11] = 11; j[12] = i[13]
            for buf, a in zip(bufs, (u, i, j)):
                buf.copy_(torch.as_tensor(a.astype(np.int32)), non_blocking=False)
            lg = eg.step(*bufs)
            lp = ep.step(bufs[0].clone(), bufs[1].clone(), bufs[2].clone())      # new pointers every step: never captured
            if step == 4 and model == "vbpr":              # a read between steps: P becomes valid for all items
                sg, sp = eg.score_block(0, 64), ep.score_block(0, 64)
                np.testing.assert_allclose(sg.cpu().numpy(), sp.cpu().numpy(), rtol=2e-3 if dtype != "fp32" else 2e-5, atol=1e-5)
            if step == 6:                                   # an outside write to a bound table
                for e in (eg, ep):
                    e.t["Gu"][3].mul_(1.5)
                    e.tables_dirty()
            st.synchronize()
            np.testing.assert_allclose(float(lg), float(lp), rtol=1e-5)
    eg.sync_check(); ep.sync_check()
    rt, at = (2e-5, 2e-6) if dtype == "fp32" else (2e-4, 2e-6)
    for n in t:
        if n == "F":
            continue
        np.testing.assert_allclose(eg.t[n].cpu().numpy(), ep.t[n].cpu().numpy(), rtol=rt, atol=at, err_msg=n)
