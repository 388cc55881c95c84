"""Sparse-batch VBPR step ("list mode", 2B < I): both projections run over the batch's DISTINCT items only -- the
reference gathers the 2B feature rows of the batch (VBPR.py:78) and its own default is --batch_size 256
(train_rec.py:23) -- instead of streaming the whole feature table twice per step.  Parity against the CPU oracle at
I = 50 000 with B in {256, 4096} (all three feature dtypes, both optimizers), equivalence with the dense form on the
same inputs, mode switches from step to step, and the projection cache of bprx_score_block / bprx_score_pairs.
Tolerances: fp32 features 1e-5 relative on scores (north_star); bf16 / fp8: operand rounding is identical to the
oracle's quant twin, what remains is fp32 summation order (2e-3 relative + 1e-4 absolute on the updated tables)."""
import numpy as np
import pytest
import torch

from fashionvisualexpl_recommend_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _engine(**kw):
    from fashionvisualexpl_recommend_amd.engine import Engine
    return Engine(**kw)


def _tables(U, I, k, d, D, seed, dtype):
    rs = np.random.RandomState(seed)
    F = synth.make_features(I, D, seed=seed)
    F = (F / np.abs(F).max()).astype(np.float32)
    if dtype == "bf16":
        F = orc.bf16_round(F)
    elif dtype == "fp8":
        F = orc.e4m3_round(F * np.float32(448.0)) / np.float32(448.0)
    return dict(Gu=synth.glorot_uniform(rs, U, k), Gi=synth.glorot_uniform(rs, I, k),
                Bi=(rs.standard_normal(I) * 0.01).astype(np.float32), Tu=synth.glorot_uniform(rs, U, d), F=F,
                E=synth.glorot_uniform(rs, D, d), Bp=synth.glorot_uniform(rs, D, 1).reshape(-1))


def _batch(U, I, B, seed):
    rs = np.random.RandomState(seed)
    u, i, j = rs.randint(U, size=B), rs.randint(I, size=B), rs.randint(I, size=B)
    u[:6] = 17                                         # one user six times
    i[10:13] = 4242 % I                                # one item three times as positive
    j[20] = i[21]                                      # an item as negative of one triplet and positive of another
    j[30] = i[30]                                      # degenerate i == j
    return u.astype(np.int32), i.astype(np.int32), j.astype(np.int32)


def _dev(a):
    return torch.as_tensor(a, device="cuda")


def _close(got, want, rtol, atol, msg, outlier_frac=0.0, outlier_abs=0.0):
    if outlier_frac:
        bad = np.abs(got - want) > atol + rtol * np.abs(want)
        assert bad.mean() <= outlier_frac, "%s: %.4f%% outside tolerance" % (msg, 100 * bad.mean())
        assert np.abs(got - want).max() <= outlier_abs, "%s: max abs diff %g" % (msg, np.abs(got - want).max())
        return
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=msg)


QUANT = {"fp32": 0, "bf16": 1, "fp8": 2}


def _resync(o, e, opt):
    """Reduced-precision feature paths: every step starts from IDENTICAL state (the oracle takes over the device's
    tables and Adam slots).  The W rows are rounded to bf16 (and [E|Bp] to e4m3) every step; an element whose fp32 atomic
    sum lands on a rounding boundary takes the neighbouring code on one side only, and carried over several steps that
    drift -- not the kernels -- would dominate the comparison (test_gpu_parity.test_vbpr_fp8_features_match_oracle)."""
    for n in ("Gu", "Gi", "Bi", "Tu", "E", "Bp"):
        getattr(o, n)[...] = e.t[n].cpu().numpy().reshape(getattr(o, n).shape)
        if opt != "sgd":
            for sl in ("m", "v"):
                o.slots[sl + n][...] = e.t[sl + "_" + n].cpu().numpy().reshape(o.slots[sl + n].shape)


@pytest.mark.parametrize("B", [256, 4096])
@pytest.mark.parametrize("dtype,opt", [("bf16", "sgd"), ("bf16", "adam_tf23"), ("fp32", "sgd"), ("fp32", "adam_tf23"),
                                       ("fp8", "sgd")])
def test_list_mode_steps_match_oracle_at_50k_items(B, dtype, opt):
    U, I, k, d = 3000, 50_000, 32, 20
    D = 256 if dtype == "fp32" else 1024
    t = _tables(U, I, k, d, D, seed=21, dtype=dtype)
    lr = 0.05 if opt == "sgd" else 0.01
    e = _engine(model="vbpr", num_users=U, num_items=I, embed_k=k, embed_d=d, feat_dim=D, feat_dtype=dtype, optimizer=opt,
                lr=lr, reg=1e-3, max_batch=B).bind(**t)
    assert 2 * B < I                                    # the per-step policy picks list mode here
    o = orc.OracleModel(**t, quant=QUANT[dtype])
    rt, at = (2e-5, 2e-6) if dtype == "fp32" else (2e-3, 1e-4)
    if opt != "sgd":
        at = max(at, 2e-3 * lr)
    for step in range(3):
        if dtype != "fp32":
            _resync(o, e, opt)
        u, i, j = _batch(U, I, B, 300 + step)
        loss = e.step(_dev(u), _dev(i), _dev(j)).item()
        want = o.step(u, i, j, opt, lr, 1e-3)
        assert loss == pytest.approx(want, rel=1e-4 if dtype != "fp32" else 2e-5)
        of, oa = (0.0, 0.0)
        if dtype == "fp8":
            of, oa = 3e-2, 1e-2 * lr
        elif dtype == "bf16" and opt != "sgd":
            of, oa = 1e-3, 3 * lr                       # Adam's first steps are sign-like (test_gpu_parity._close)
        for n in ("Gu", "Gi", "Bi", "Tu", "E", "Bp"):
            _close(e.t[n].cpu().numpy().reshape(-1), getattr(o, n).reshape(-1), rt, at, "%s step %d" % (n, step), of, oa)
    # scores after the steps (projection of the listed pairs) and the whole-table projection agree with the oracle
    u, i, _ = _batch(U, I, 1000 if B >= 1000 else B, 9)
    if dtype != "fp32":
        _resync(o, e, opt)
    st, sa = (1e-5, 1e-6) if dtype == "fp32" else ((1e-4, 2e-5) if dtype == "bf16" else (2e-4, 5e-5))
    _close(e.score_pairs(u, i).cpu().numpy(), o.score_pairs(u, i), st, sa, "score_pairs")
    e.sync_check()


@pytest.mark.parametrize("k,d,D,dtype,opt", [(5, 3, 100, "fp32", "sgd"), (7, 1, 64, "fp32", "adam_tf23"),     # odd widths: scalar lanes
                                             (16, 20, 128, "bf16", "sgd"),                                     # ONE k-chunk: 7 idle waves
                                             (16, 20, 256, "fp8", "sgd"),                                      # fp8: one 256-byte chunk
                                             (8, 143, 384, "bf16", "sgd"), (8, 200, 512, "bf16", "adam_tf23"),   # 9 / 13 column tiles
                                             (8, 256, 512, "fp8", "sgd")])                                     # 17 column tiles
def test_list_mode_odd_shapes_match_oracle(k, d, D, dtype, opt):
    """The row-list kernels at the edges of their templates: widths that are no multiple of 4, a single k-chunk (fewer
    chunks than waves), every column-tile count class of k_proj_fwd_rows (split / full, 1..17 tiles)."""
    U, I, B = 300, 1500, 96
    t = _tables(U, I, k, d, D, seed=31, dtype=dtype)
    lr = 0.05 if opt == "sgd" else 0.01
    e = _engine(model="vbpr", num_users=U, num_items=I, embed_k=k, embed_d=d, feat_dim=D, feat_dtype=dtype, optimizer=opt,
                lr=lr, reg=1e-3, max_batch=B).bind(**t)
    o = orc.OracleModel(**t, quant=QUANT[dtype])
    rt, at = (2e-5, 2e-6) if dtype == "fp32" else (2e-3, 1e-4)
    if opt != "sgd":
        at = max(at, 2e-3 * lr)
    for step in range(3):
        if dtype != "fp32":
            _resync(o, e, opt)
        u, i, j = _batch(U, I, B, 500 + step)
        loss = e.step(_dev(u), _dev(i), _dev(j)).item()
        want = o.step(u, i, j, opt, lr, 1e-3)
        assert loss == pytest.approx(want, rel=1e-4 if dtype != "fp32" else 2e-5)
        # adam: lr_t*m/(sqrt(v)+eps) is steep in g around |g| ~ eps -- an element whose gradient nearly cancels (an item that
        # is positive in one triplet and negative in another) can move by a visible fraction of lr on a last-bit difference
        of, oa = ((1e-3, 3 * lr) if opt != "sgd" else (0.0, 0.0)) if dtype == "fp32" else \
            ((1e-3, 3 * lr) if opt != "sgd" else (3e-2, 1e-2 * lr))
        for n in ("Gu", "Gi", "Bi", "Tu", "E", "Bp"):
            _close(e.t[n].cpu().numpy().reshape(-1), getattr(o, n).reshape(-1), rt, at, "%s step %d" % (n, step), of, oa)
    if dtype != "fp32" or opt != "sgd":
        _resync(o, e, opt)
    u, i, _ = _batch(U, I, 64, 9)
    st, sa = (1e-5, 1e-6) if dtype == "fp32" else ((1e-4, 2e-5) if dtype == "bf16" else (2e-4, 5e-5))
    _close(e.score_pairs(u, i).cpu().numpy(), o.score_pairs(u, i), st, sa, "score_pairs")
    _close(e.score_block(0, 64).cpu().numpy(), o.predict_all()[:64], st * 10, sa * 10, "predict_all rows")
    e.sync_check()


@pytest.mark.parametrize("dtype", ["bf16", "fp32", "fp8"])
def test_list_mode_equals_dense_mode(monkeypatch, dtype):
    """Same inputs through BPRX_LIST_MODE=0 (both projections stream every item) and =2 (distinct items only)."""
    U, I, k, d, D, B = 500, 3000, 64, 64, 512, 300
    t = _tables(U, I, k, d, D, seed=5, dtype=dtype)
    res = {}
    for mode in (0, 2):
        monkeypatch.setenv("BPRX_LIST_MODE", str(mode))
        e = _engine(model="vbpr", num_users=U, num_items=I, embed_k=k, embed_d=d, feat_dim=D, feat_dtype=dtype,
                    optimizer="sgd", lr=0.05, reg=1e-3, max_batch=B).bind(**t)
        losses = []
        for step in range(3):
            u, i, j = _batch(U, I, B, 50 + step)
            losses.append(e.step(_dev(u), _dev(i), _dev(j)).item())
        e.sync_check()
        res[mode] = (losses, {n: e.t[n].cpu().numpy().copy() for n in ("Gu", "Gi", "Bi", "Tu", "E", "Bp")})
        e.close()
    assert res[0][0] == pytest.approx(res[2][0], rel=1e-5)
    for n in res[0][1]:
        # fp8: the two forms sum in different orders, so [E|Bp] can differ in the last bits before it is re-quantised
        # bf16: a W element whose fp32 atomic sum lands on a bf16 rounding boundary may flip from run to run (atomic order)
        rt, at = {"fp32": (1e-5, 1e-7), "bf16": (1e-3, 1e-5), "fp8": (2e-3, 1e-4)}[dtype]
        np.testing.assert_allclose(res[2][1][n], res[0][1][n], rtol=rt, atol=at, err_msg=n)


@pytest.mark.parametrize("dtype,opt", [("bf16", "sgd"), ("fp32", "sgd"), ("fp32", "adam_tf23"), ("bf16", "adam_tf23")])
def test_mode_switches_from_step_to_step(dtype, opt):
    """One handle, batches of changing size: list mode (2B < I; forced off for some steps to keep the atomic-staging form covered) and occurrence segments
    (2B >= I) in turn; the staging tables, multiplicity counters, W rows and the item list must be clean after each."""
    U, I, k, d, D = 200, 1000, 32, 20, 256
    t = _tables(U, I, k, d, D, seed=8, dtype=dtype)
    lr = 0.05 if opt == "sgd" else 0.01
    e = _engine(model="vbpr", num_users=U, num_items=I, embed_k=k, embed_d=d, feat_dim=D, feat_dtype=dtype, optimizer=opt,
                lr=lr, reg=1e-3, max_batch=1024).bind(**t)
    o = orc.OracleModel(**t, quant=QUANT[dtype])
    rt, at = (2e-5, 2e-6) if dtype == "fp32" else (2e-3, 1e-4)
    if opt != "sgd":
        at = max(at, 2e-3 * lr)
    for step, B in enumerate([128, 1024, 64, 400, 250, 1, 1024, 200]):
        if dtype != "fp32":
            _resync(o, e, opt)
        u, i, j = _batch(U, I, max(B, 32), 70 + step)
        u, i, j = u[:B], i[:B], j[:B]
        loss = e.step(_dev(u), _dev(i), _dev(j)).item()
        want = o.step(u, i, j, opt, lr, 1e-3)
        assert loss == pytest.approx(want, rel=1e-4 if dtype != "fp32" else 2e-5), (step, B)
        # bf16: W elements on a bf16 rounding boundary may flip (fp32 summation order; a flipped W[t, d] moves every Bp
        # element a little, and eight steps with B up to 1024 accumulate it): <= 3 % of the elements may miss, by <= 0.01 lr
        of, oa = (0.0, 0.0) if dtype == "fp32" else ((1e-3, 3 * lr) if opt != "sgd" else (3e-2, 1e-2 * lr))
        for n in ("Gu", "Gi", "Bi", "Tu", "E", "Bp"):
            _close(e.t[n].cpu().numpy().reshape(-1), getattr(o, n).reshape(-1), rt, at, "%s step %d B %d" % (n, step, B), of, oa)
    e.sync_check()


def test_projection_cache_of_score_block_and_score_pairs():
    """bprx_score_block projects the feature table once per parameter state, not once per user block; a step or
    bprx_tables_dirty() invalidates it (Evaluator.py:174 calls predict_all once per epoch; the blocked form calls
    bprx_score_block per 4096 users)."""
    U, I, k, d, D = 300, 700, 32, 20, 256
    t = _tables(U, I, k, d, D, seed=4, dtype="bf16")
    e = _engine(model="vbpr", num_users=U, num_items=I, embed_k=k, embed_d=d, feat_dim=D, feat_dtype="bf16",
                optimizer="sgd", lr=0.05, reg=1e-3, max_batch=256).bind(**t)
    o = orc.OracleModel(**t, quant=1)
    want = o.predict_all()
    e.profile(True)
    a = torch.cat([e.score_block(0, 100), e.score_block(100, U)]).cpu().numpy()
    u, i, _ = _batch(U, I, 200, 3)
    x = e.score_pairs(u, i).cpu().numpy()
    prof = e.profile_read()
    assert prof["proj_fwd"][1] == 1, prof                 # ONE projection launch for two blocks and the pair scores
    _close(a, want, 1e-4, 2e-5, "predict_all")
    _close(x, o.score_pairs(u, i), 1e-4, 2e-5, "score_pairs from the cached projections")
    # outside write to E + tables_dirty -> the next call re-projects
    e.t["E"].mul_(0.5)
    o.E[...] = o.E * np.float32(0.5)
    e.tables_dirty()
    b = e.score_block(0, U).cpu().numpy()
    assert e.profile_read()["proj_fwd"][1] == 1
    _close(b, o.predict_all(), 1e-4, 2e-5, "predict_all after tables_dirty")
    # a step moves E/Bp -> re-projected as well
    u, i, j = _batch(U, I, 256, 6)
    e.step(_dev(u), _dev(i), _dev(j))
    o.step(u, i, j, "sgd", 0.05, 1e-3)
    e.profile_read()
    c = e.score_block(0, U).cpu().numpy()
    assert e.profile_read()["proj_fwd"][1] == 1
    _close(c, o.predict_all(), 2e-3, 1e-4, "predict_all after a step")
    e.profile(False)
    e.sync_check()
