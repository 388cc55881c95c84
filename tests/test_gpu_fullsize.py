"""BASELINE.json configs[1], [3] (per-GPU shard) and [4] at FULL size -- VBPR k=d=64 bf16 100K x 50K; VBPR k=d=128 bf16
250K x 62.5K (the c4shard shape: the 9-tile forward, the 8-wave backward with three tiles in flight); VBPR k=d=256 fp8
(17 column tiles, the one-pass scaled-fp8 forward) at the cache-resident size (c5small: 100K x 50K, 205-MB table), at HBM
scale (c5: 1M x 500K, a 2-GB table streamed with `nt` loads, B = 262 144) and in list mode at that scale (c5list: B = 65 536,
the projections run over the batch's distinct items) -- and the configs[2] per-GPU shard (BPRMF k=128,
625K x 1M): the oracle cannot run these in seconds, so the HIP path is checked through size-independent properties and
against an independent torch fp32 recomputation of the SAME step on the device with the same operand rounding (torch is
the checker here, never the product path).  Reference: VBPR.py:59-144, BPRMF.py:55-125."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _state(workload):
    import bench
    w = dict(bench.WORKLOADS[workload])
    dev = torch.device("cuda", 0)
    return w, dev, bench.make_state(w, dev, 77, torch)


def _sampler_properties(U, I, B, dev, g):
    """positives are training interactions, negatives are not; the stream is stateless"""
    from fashionvisualexpl_recommend_amd.engine import PhiloxSampler
    npu = 20
    items = torch.randint(I, (U, npu), generator=g, device=dev, dtype=torch.int32).sort(dim=1).values
    indptr = torch.arange(U + 1, device=dev, dtype=torch.int64) * npu
    pos_user = torch.arange(U, device=dev, dtype=torch.int32).repeat_interleave(npu)
    s = PhiloxSampler.from_csr(indptr, items.reshape(-1), pos_user, I, seed=9)
    u, i, j = s.sample(B)
    rows = items[u.long()]
    assert bool((rows == i[:, None]).any(dim=1).all())
    assert not bool((rows == j[:, None]).any(dim=1).any())
    assert int(u.min()) >= 0 and int(u.max()) < U and int(j.min()) >= 0 and int(j.max()) < I
    u2, i2, j2 = s.sample(B, first=0)
    assert torch.equal(u, u2) and torch.equal(i, i2) and torch.equal(j, j2)          # stateless: same slice, same triplets
    return u, i, j


@pytest.mark.parametrize("workload", ["c2", "c4shard", "c5small", "c5", "c5list"])
def test_vbpr_full_size_step_against_torch_fp32(workload):
    from fashionvisualexpl_recommend_amd.engine import Engine
    w, dev, t = _state(workload)
    U, I, k, d, D, B = w["U"], w["I"], w["k"], w["d"], w["D"], w["B"]
    fp8 = w["dtype"] == "fp8"
    lr, reg = 1e-3, 1e-4
    g = torch.Generator(device=dev); g.manual_seed(5)
    t["Bi"] = torch.randn(I, generator=g, device=dev) * 0.01
    before = {n: v.clone() for n, v in t.items() if n != "F"}
    eng = Engine(model="vbpr", num_users=U, num_items=I, embed_k=k, embed_d=d, feat_dim=D, feat_dtype=w["dtype"],
                 optimizer="sgd", lr=lr, reg=reg, max_batch=B).bind(**t)
    u, i, j = _sampler_properties(U, I, B, dev, g)

    # ---- independent fp32 recomputation of the step: the SAME operand rounding (bf16 / e4m3 operands of the two
    # projections), fp32 (matmuls) and fp64 (scatter sums) everywhere else ----
    ul, il, jl = u.long(), i.long(), j.long()
    F = t["F"]
    EB = torch.cat([before["E"], before["Bp"][:, None]], 1)                                       # [D, d+1]
    if fp8:
        sE = torch.tensor(448.0, device=dev) / EB.abs().max()                                     # k_absmax / k_cast_Et8
        Eq = (EB * sE).to(torch.float8_e4m3fn).float() / sE
        frow = lambda idx: F[idx].float() / 448.0                                                 # codes of f * 448
    else:
        Eq = EB.to(torch.bfloat16).float()
        frow = lambda idx: F[idx].float()
    touched = torch.unique(torch.cat([il, jl]))
    Pt = torch.cat([frow(touched[s0:s0 + 8192]) @ Eq for s0 in range(0, touched.numel(), 8192)])  # [nT, d+1]
    slot = torch.full((I,), -1, device=dev, dtype=torch.long); slot[touched] = torch.arange(touched.numel(), device=dev)
    Pi, Pj = Pt[slot[il]], Pt[slot[jl]]
    gu, tu = before["Gu"][ul], before["Tu"][ul]
    gi, gj = before["Gi"][il], before["Gi"][jl]
    xp = before["Bi"][il] + (gu * gi).sum(1) + (tu * Pi[:, :d]).sum(1) + Pi[:, d]
    xn = before["Bi"][jl] + (gu * gj).sum(1) + (tu * Pj[:, :d]).sum(1) + Pj[:, d]
    got_xp = eng.score_pairs(u, i)
    torch.testing.assert_close(got_xp, xp, rtol=2e-4, atol=2e-4)
    got_blk = eng.score_block(1000, 1064)                      # predict_all rows through the full-table projection
    blk_want = before["Bi"][None, :] + before["Gu"][1000:1064] @ before["Gi"].T
    Pall_d = torch.cat([frow(torch.arange(s0, min(I, s0 + 8192), device=dev)) @ Eq for s0 in range(0, I, 8192)])
    blk_want = blk_want + before["Tu"][1000:1064] @ Pall_d[:, :d].T + Pall_d[:, d][None, :]
    torch.testing.assert_close(got_blk, blk_want, rtol=2e-4, atol=2e-4)
    del Pall_d, blk_want, got_blk
    diff = xp - xn
    gg = -torch.sigmoid(-diff)
    loss_want = torch.nn.functional.softplus(-diff).double().sum() + reg * (
        (gu.double() ** 2).sum() + (gi.double() ** 2).sum() + (gj.double() ** 2).sum() + (tu.double() ** 2).sum()
        + (before["Bi"][il].double() ** 2).sum() + (before["Bi"][jl].double() ** 2).sum() / 10
        + (before["E"].double() ** 2).sum() + (before["Bp"].double() ** 2).sum())
    loss = float(eng.step(u, i, j).item())
    eng.sync_check()
    assert loss == pytest.approx(float(loss_want), rel=2e-4)

    def scatter(n_rows, idx, vals):
        out = torch.zeros((n_rows, vals.shape[1]), device=dev, dtype=torch.float64)
        return out.index_add_(0, idx, vals.double())
    dGu = scatter(U, ul, gg[:, None] * (gi - gj) + 2 * reg * gu)
    dTu = scatter(U, ul, gg[:, None] * (Pi[:, :d] - Pj[:, :d]) + 2 * reg * tu)
    dGi = scatter(I, il, gg[:, None] * gu + 2 * reg * gi) + scatter(I, jl, -gg[:, None] * gu + 2 * reg * gj)
    dBi = scatter(I, il, (gg + 2 * reg * before["Bi"][il])[:, None]) + \
        scatter(I, jl, (-gg + 0.2 * reg * before["Bi"][jl])[:, None])
    gth = torch.cat([gg[:, None] * tu, gg[:, None]], 1)
    W = (scatter(I, il, gth) - scatter(I, jl, gth)).float().to(torch.bfloat16).float()             # bf16 like the MFMA operand
    del gth
    dEq = torch.zeros((D, d + 1), device=dev, dtype=torch.float32)
    for s0 in range(0, I, 8192):                                                                   # F^T W in fp32 chunks
        dEq += frow(torch.arange(s0, min(I, s0 + 8192), device=dev)).T @ W[s0:s0 + 8192]
    want = {"Gu": before["Gu"] - lr * dGu.float(), "Tu": before["Tu"] - lr * dTu.float(),
            "Gi": before["Gi"] - lr * dGi.float(), "Bi": before["Bi"] - lr * dBi.float()[:, 0],
            "E": before["E"] - lr * (dEq[:, :d] + 2 * reg * before["E"]),
            "Bp": before["Bp"] - lr * (dEq[:, d] + 2 * reg * before["Bp"])}
    for n, wv in want.items():
        delta_scale = float((wv - before[n]).abs().max()) + 1e-12
        err = float((eng.t[n] - wv).abs().max())
        assert err <= 5e-3 * delta_scale + 1e-7, (workload, n, err, delta_scale)      # error relative to the size of the update
    # conservation: with the +g / -g bias gradients, sum(dBi) carries only the regularisation terms
    assert abs(float(dBi.sum()) - float((2 * reg * before["Bi"][il].double()).sum()
                                         + (0.2 * reg * before["Bi"][jl].double()).sum())) < 1e-6 * B
    # a second step from the updated state must keep every table finite and move the loss only slightly
    loss2 = float(eng.step(u, i, j).item())
    eng.sync_check()
    assert np.isfinite(loss2) and abs(loss2 - loss) < 0.05 * abs(loss)


def test_c3_shard_full_size_bprmf_against_torch_fp32():
    from fashionvisualexpl_recommend_amd.engine import Engine
    w, dev, t = _state("c3shard")
    U, I, k, B = w["U"], w["I"], w["k"], w["B"]
    lr, reg = 0.05, 1e-4
    before = {n: v.clone() for n, v in t.items()}
    eng = Engine(model="bprmf", num_users=U, num_items=I, embed_k=k, optimizer="sgd", lr=lr, reg=reg, max_batch=B).bind(**t)
    g = torch.Generator(device=dev); g.manual_seed(6)
    u = torch.randint(U, (B,), generator=g, device=dev, dtype=torch.int32)
    i = torch.randint(I, (B,), generator=g, device=dev, dtype=torch.int32)
    j = torch.randint(I, (B,), generator=g, device=dev, dtype=torch.int32)
    u[:64] = 12345                                           # force some heavily shared rows next to the exclusive majority
    j[100:110] = i[100:110]
    ul, il, jl = u.long(), i.long(), j.long()
    gu, gi, gj = before["Gu"][ul], before["Gi"][il], before["Gi"][jl]
    diff = (before["Bi"][il] + (gu * gi).sum(1)) - (before["Bi"][jl] + (gu * gj).sum(1))
    gg = -torch.sigmoid(-diff)
    eng.step(u, i, j)
    eng.sync_check()

    def scatter(n_rows, idx, vals):
        return torch.zeros((n_rows, vals.shape[1]), device=dev, dtype=torch.float64).index_add_(0, idx, vals.double())
    dGu = scatter(U, ul, gg[:, None] * (gi - gj) + 2 * reg * gu)
    dGi = scatter(I, il, gg[:, None] * gu + 2 * reg * gi) + scatter(I, jl, -gg[:, None] * gu + 2 * reg * gj)
    torch.testing.assert_close(eng.t["Gu"], before["Gu"] - lr * dGu.float(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(eng.t["Gi"], before["Gi"] - lr * dGi.float(), rtol=1e-5, atol=1e-6)
    untouched = torch.ones(U, dtype=torch.bool, device=dev); untouched[ul] = False
    assert torch.equal(eng.t["Gu"][untouched], before["Gu"][untouched])        # rows outside the batch are bit-identical
