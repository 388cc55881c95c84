"""Lazy-exact adam_tf23: TF-2.3's Adam moves EVERY row of every table each step (non-lazy sparse apply, BPRMF.py:52,123 /
VBPR.py:56,142); libbprx replays the skipped steps of a row when the row is next read (bprx_sparse.hip).  The replay must
be BIT-IDENTICAL to the whole-table sweeps (BPRX_ADAM_LAZY=0) -- shown on batches without duplicate rows, where the
gradients themselves are free of fp32 atomic-order noise -- and agree with the CPU oracle's non-lazy rule otherwise."""
import numpy as np
import pytest
import torch

from fashionvisualexpl_recommend_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.as_tensor(a, device="cuda")


def _tables(U, I, k, d, D, seed):
    rs = np.random.RandomState(seed)
    t = dict(Gu=synth.glorot_uniform(rs, U, k), Gi=synth.glorot_uniform(rs, I, k),
             Bi=(rs.standard_normal(I) * 0.01).astype(np.float32))
    if d:
        F = synth.make_features(I, D, seed=seed)
        t.update(Tu=synth.glorot_uniform(rs, U, d), F=(F / np.abs(F).max()).astype(np.float32),
                 E=synth.glorot_uniform(rs, D, d), Bp=synth.glorot_uniform(rs, D, 1).reshape(-1))
    return t


def _unique_batches(U, I, B, steps, seed):
    """No row twice inside a batch (users distinct, the 2B items distinct); rows recur across batches after varying gaps."""
    rs = np.random.RandomState(seed)
    out = []
    for s in range(steps):
        hotU, hotI = (U // 4, I // 4) if s % 3 else (U, I)          # every third batch touches the whole range: long gaps
        u = rs.choice(hotU, B, replace=False)
        it = rs.choice(hotI, 2 * B, replace=False)
        out.append((u.astype(np.int32), it[:B].astype(np.int32), it[B:].astype(np.int32)))
    return out


@pytest.mark.parametrize("model", ["bprmf", "vbpr"])
def test_lazy_replay_is_bit_identical_to_the_sweeps(monkeypatch, model):
    from fashionvisualexpl_recommend_amd.engine import Engine
    U, I, k, B = 3000, 4000, 32, 200
    d, D = (20, 64) if model == "vbpr" else (0, 0)
    t = _tables(U, I, k, d, D, seed=3)
    batches = _unique_batches(U, I, B, 14, seed=9)
    res = {}
    for lazy in (0, 1):
        monkeypatch.setenv("BPRX_ADAM_LAZY", str(lazy))
        kw = dict(embed_d=d, feat_dim=D, feat_dtype="fp32") if d else {}
        e = Engine(model=model, num_users=U, num_items=I, embed_k=k, optimizer="adam_tf23", lr=0.01, reg=1e-3, max_batch=B,
                   **kw).bind(**t)
        losses, mid = [], None
        for s, (u, i, j) in enumerate(batches):
            losses.append(e.step(_dev(u), _dev(i), _dev(j)).item())
            if s == 6:                                               # a mid-run predict_all must not disturb the run
                mid = e.score_block(0, 64).cpu().numpy().copy()
        names = ["Gu", "Gi", "Bi", "m_Gu", "v_Gu", "m_Gi", "v_Gi", "m_Bi", "v_Bi"] + \
                (["Tu", "m_Tu", "v_Tu"] if d else [])
        res[lazy] = (losses, mid, {n: e.t[n].cpu().numpy().copy() for n in names})     # e.t: bprx_sync_adam first
        e.sync_check()
        e.close()
    assert res[0][0] == res[1][0]                                   # identical losses, step by step
    np.testing.assert_array_equal(res[1][1], res[0][1])
    for n in res[0][2]:
        a, b = res[0][2][n], res[1][2][n]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "%s: %d of %d words differ" % (
            n, int((a.view(np.uint32) != b.view(np.uint32)).sum()), a.size)


def test_lazy_adam_survives_the_lr_ring_wrap():
    """More steps than the device ring of lr_t values holds (8192): the library must catch everything up before a slot
    is overwritten.  A row touched at step 1 and again only at the end is compared with the oracle's non-lazy rule."""
    from fashionvisualexpl_recommend_amd.engine import Engine
    U, I, k = 40, 60, 8
    t = _tables(U, I, k, 0, 0, seed=5)
    e = Engine(model="bprmf", num_users=U, num_items=I, embed_k=k, optimizer="adam_tf23", lr=0.01, reg=0.0, max_batch=4).bind(**t)
    o = orc.OracleModel(**t)
    rs = np.random.RandomState(1)
    first = (np.array([0, 1], np.int32), np.array([0, 1], np.int32), np.array([2, 3], np.int32))
    steps = [first] + [tuple(rs.randint(10, n, size=2).astype(np.int32) for n in (U, I, I)) for _ in range(8300)] + [first]
    for u, i, j in steps:
        e.step(_dev(u), _dev(i), _dev(j), want_loss=False)
        o.step(u, i, j, "adam_tf23", 0.01, 0.0)
    for n in ("Gu", "Gi", "Bi"):
        np.testing.assert_allclose(e.t[n].cpu().numpy().reshape(-1), getattr(o, n).reshape(-1), rtol=2e-4, atol=2e-5, err_msg=n)
    e.sync_check()


def test_snapshot_and_restore_with_pending_rows(tmp_path):
    """state snapshots read the tensors (sync first) and a restore declares every row current (bprx_tables_dirty)."""
    from fashionvisualexpl_recommend_amd.engine import Engine
    U, I, k, B = 300, 400, 16, 64
    t = _tables(U, I, k, 0, 0, seed=7)
    mk = lambda: Engine(model="bprmf", num_users=U, num_items=I, embed_k=k, optimizer="adam_tf23", lr=0.01, reg=1e-3,
                        max_batch=B).bind(**t)
    rs = np.random.RandomState(2)
    bs = [tuple(rs.randint(n, size=B).astype(np.int32) for n in (U, I, I)) for _ in range(8)]
    a = mk()
    for u, i, j in bs[:4]:
        a.step(_dev(u), _dev(i), _dev(j), want_loss=False)
    snap = {n: v.clone() for n, v in a.t.items()}
    step = a.adam_step
    for u, i, j in bs[4:]:
        a.step(_dev(u), _dev(i), _dev(j), want_loss=False)
    want = {n: v.cpu().numpy().copy() for n, v in a.t.items()}
    b = mk()
    for u, i, j in bs[:2]:                                            # leave b with pending rows of another history
        b.step(_dev(u), _dev(i), _dev(j), want_loss=False)
    for n, v in snap.items():
        b._t[n].copy_(v)                                              # outside write, as models.load_state_dict does
    b.adam_step = step
    b.tables_dirty()
    for u, i, j in bs[4:]:
        b.step(_dev(u), _dev(i), _dev(j), want_loss=False)
    for n in want:
        np.testing.assert_allclose(b.t[n].cpu().numpy(), want[n], rtol=1e-5, atol=1e-7, err_msg=n)


def test_restore_on_a_non_default_stream_with_a_smaller_step_counter():
    """the lazy-Adam bookkeeping reset of bprx_tables_dirty / bprx_set_adam_step is ordered on the CALLER's stream
    (it used to run on the NULL stream, unordered with respect to the sync / copies a restore had just enqueued on a
    non-blocking stream): a restore to an EARLIER step inside torch.cuda.stream(s) must not leave last[row] > adam_t."""
    import torch
    from fashionvisualexpl_recommend_amd.engine import Engine
    U, I, k, B = 300, 400, 16, 64
    t = _tables(U, I, k, 0, 0, seed=11)
    e = Engine(model="bprmf", num_users=U, num_items=I, embed_k=k, optimizer="adam_tf23", lr=0.01, reg=1e-3, max_batch=B).bind(**t)
    rs = np.random.RandomState(4)
    bs = [tuple(rs.randint(n, size=B).astype(np.int32) for n in (U, I, I)) for _ in range(10)]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for u, i, j in bs[:3]:
            e.step(_dev(u), _dev(i), _dev(j), want_loss=False)
        snap = {n: v.clone() for n, v in e.t.items()}
        step = e.adam_step
        for u, i, j in bs[3:8]:
            e.step(_dev(u), _dev(i), _dev(j), want_loss=False)
        want = None
        for n, v in snap.items():                                     # restore the EARLIER state (step 3 < 8)
            e._t[n].copy_(v)
        e.adam_step = step
        e.tables_dirty()
        for u, i, j in bs[3:8]:                                       # the same five steps again
            e.step(_dev(u), _dev(i), _dev(j), want_loss=False)
        got = {n: v.cpu().numpy().copy() for n, v in e.t.items()}
    side.synchronize()
    o = orc.OracleModel(**t)
    for u, i, j in bs[:8]:
        o.step(u, i, j, "adam_tf23", 0.01, 1e-3)
    for n in ("Gu", "Gi", "Bi"):
        np.testing.assert_allclose(got[n].reshape(-1), getattr(o, n).reshape(-1), rtol=2e-4, atol=2e-5, err_msg=n)


def test_model_attributes_are_current_under_lazy_adam(tmp_path):
    """model.Gu / .Gi / .Bi (the reference's attribute surface) go through engine.t, i.e. rows that lazy
    adam_tf23 has not replayed yet are brought up to date before they are handed out."""
    from argparse import Namespace
    from fashionvisualexpl_recommend_amd import configs, synth
    from fashionvisualexpl_recommend_amd.dataset import DataLoader
    from fashionvisualexpl_recommend_amd.models import BPRMF
    tr, va, te = synth.make_interactions_clustered(200, 300, per_user=12, clusters=6, p_in=0.9, seed=3)
    synth.write_dataset(str(tmp_path), "attr", tr, va, te, 300)
    configs.set_roots(str(tmp_path), str(tmp_path / "results"))
    params = Namespace(dataset="attr", validation=True, batch_size=64, epochs=1, batch_eval=128, embed_k=16, lr=0.01, reg=1e-3,
                       top_k=10, verbose=-1, restore_epochs=1, rec="bprmf", best_metric="ndcg", optimizer="adam_tf23", init_seed=0)
    model = BPRMF(DataLoader(params), params)
    assert model.engine.adam_is_lazy()                                # (conftest sets BPRX_ADAM_LAZY=1)
    o = orc.OracleModel(**{n: v.cpu().numpy().copy() for n, v in model.engine.params().items()})
    rs = np.random.RandomState(8)
    for _ in range(6):
        u, i, j = (rs.randint(n, size=64).astype(np.int32) for n in (200, 300, 300))
        model.train_step((u, i, j))
        o.step(u, i, j, "adam_tf23", 0.01, 1e-3)
    for n in ("Gu", "Gi", "Bi"):                                      # most rows were NOT in the last batch: they need the replay
        np.testing.assert_allclose(getattr(model, n).cpu().numpy().reshape(-1), getattr(o, n).reshape(-1), rtol=2e-4, atol=2e-5,
                                   err_msg=n)


def test_adam_policy_picks_sweeps_for_small_batches_and_lazy_for_large_ones(monkeypatch):
    """bprx_create chooses the form from 20 U / B replayed steps against the bytes of a sweep: the reference's own defaults
    (batch 256 on small tables) get sweeps, bench-sized batches the lazy replay; either way the step is the same step
    (bit-identical tables on a duplicate-free run)."""
    from fashionvisualexpl_recommend_amd.engine import Engine
    monkeypatch.delenv("BPRX_ADAM_LAZY", raising=False)
    mk = lambda U, I, k, B, **kw: Engine(model="bprmf", num_users=U, num_items=I, embed_k=k, optimizer="adam_tf23", lr=0.01, reg=1e-3,
                                         max_batch=B, **kw)
    e = mk(20000, 10000, 128, 256)
    assert not e.adam_is_lazy()                                     # train_rec.py defaults on a small catalogue
    e.close()
    e = mk(100000, 50000, 64, 65536)
    assert e.adam_is_lazy()                                         # BASELINE.json configs[1] shape
    e.close()
    U, I, k, B = 3000, 4000, 32, 200
    t = _tables(U, I, k, 0, 0, seed=4)
    batches = _unique_batches(U, I, B, 10, seed=2)
    res = []
    for force in (None, "1"):
        if force:
            monkeypatch.setenv("BPRX_ADAM_LAZY", force)
        e = mk(U, I, k, B).bind(**t)
        if force is None:
            assert not e.adam_is_lazy()
        for u, i, j in batches:
            e.step(_dev(u), _dev(i), _dev(j))
        res.append({n: e.t[n].cpu().numpy().copy() for n in ("Gu", "Gi", "Bi", "m_Gu", "v_Gi")})
        e.close()
    for n in res[0]:
        assert np.array_equal(res[0][n].view(np.uint32), res[1][n].view(np.uint32)), n
