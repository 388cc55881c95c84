"""Device top-K for store_recommendation (bprx_topk, Evaluator.py:225-239) and the ADVICE edge cases of the device
evaluator: the TSV written through the device path must be BYTE-IDENTICAL to what the reference's own
Evaluator.store_recommendation wrote for the same score matrices (tests/golden/eval_tiny_recs.tsv, the sha256 of the
C1-shaped run in golden.json; tests/golden/gen_golden.py ran the reference), ties included: rows whose list depends on
the order of equal scores are flagged by the kernel and redone with numpy."""
import hashlib
import json
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from fashionvisualexpl_recommend_amd import configs, synth
from fashionvisualexpl_recommend_amd.dataset import DataLoader
from fashionvisualexpl_recommend_amd.evaluator import Evaluator, _eval_block

pytestmark = pytest.mark.gpu


class _ScoreModel:
    """A model whose predict_all() rows are a given matrix: BPRMF engine with Gu = one-hot rows, Gi = scores^T would need
    k = U; simpler and exact: k = 1 factors of zero and the scores injected per block through score_block's output."""

    def __init__(self, data, scores):
        from fashionvisualexpl_recommend_amd.engine import Engine
        U, I = scores.shape
        self.data, self.scores = data, torch.as_tensor(scores, device="cuda")
        self.engine = Engine(model="bprmf", num_users=U, num_items=I, embed_k=4, optimizer="sgd", max_batch=8)
        self.engine.bind(Gu=np.zeros((U, 4), np.float32), Gi=np.zeros((I, 4), np.float32), Bi=np.zeros(I, np.float32))
        real = self.engine.score_block
        self.engine.score_block = lambda u0, u1, out=None: self.scores[u0:u1].clone()   # the kernel masks it in place
        self._real = real

    def predict_block(self, u0, u1):
        return self.scores[u0:u1].cpu().numpy().copy()


def _golden(golden_dir):
    return json.load(open(os.path.join(golden_dir, "golden.json")))


def test_store_recommendation_tiny_is_byte_identical(golden_dir, tmp_path):
    want = _golden(golden_dir)["eval_tiny"]
    ds = json.load(open(os.path.join(golden_dir, "dataset_tiny.json")))
    sc = np.load(os.path.join(golden_dir, "eval_tiny_scores.npy"))
    data = Namespace(training_list=ds["loaded_train"], validation_list=ds["loaded_val"], test_list=ds["loaded_test"],
                     num_users=sc.shape[0], num_items=sc.shape[1], params=Namespace(batch_eval=128))
    ev = Evaluator(_ScoreModel(data, sc), data, want["K"], user_block=4)
    p = tmp_path / "recs.tsv"
    ev.store_recommendation(str(p))
    assert p.read_text() == open(os.path.join(golden_dir, "eval_tiny_recs.tsv")).read()


def test_store_recommendation_c1_matches_reference_sha256(golden_dir, tmp_path):
    want = _golden(golden_dir)["eval_c1"]
    tr, va, te = synth.make_interactions(1000, 2000, per_user=22, seed=2024)
    sc = np.random.RandomState(want["score_seed"]).standard_normal((1000, 2000)).astype(np.float32)
    data = Namespace(training_list=tr, validation_list=va, test_list=te, num_users=1000, num_items=2000,
                     params=Namespace(batch_eval=128))
    m = _ScoreModel(data, sc)
    ev = Evaluator(m, data, want["K"], user_block=300)
    p = tmp_path / "recs.tsv"
    ev.store_recommendation(str(p))
    txt = p.read_text()
    assert txt.splitlines()[:20] == want["recs_head"]
    assert hashlib.sha256(txt.encode()).hexdigest() == want["recs_sha256"]
    # the kernel did the work: no row of this real-valued matrix needs the numpy redo
    idx, val, flag = m.engine.topk(0, 300, torch.as_tensor(sc[:300], device="cuda").clone(), ev._csr["train"], want["K"])
    assert int(flag.sum()) == 0


@pytest.mark.parametrize("K", [1, 10, 100, 1000])
def test_topk_matches_numpy_on_distinct_scores(K):
    from fashionvisualexpl_recommend_amd.engine import Engine
    U, I = 64, 5000
    rs = np.random.RandomState(K)
    sc = rs.permutation(U * I).reshape(U, I).astype(np.float32)           # all distinct, exactly representable
    sc -= sc.mean()
    tr = [sorted(rs.choice(I, rs.randint(0, 40), replace=False).tolist()) for _ in range(U)]
    tr[3] = tr[3] + tr[3][:2]                                             # a duplicated train row
    e = Engine(model="bprmf", num_users=U, num_items=I, embed_k=4, optimizer="sgd", max_batch=8)
    e.bind(Gu=np.zeros((U, 4), np.float32), Gi=np.zeros((I, 4), np.float32), Bi=np.zeros(I, np.float32))
    indptr = np.zeros(U + 1, np.int64)
    for u, l in enumerate(tr):
        indptr[u + 1] = indptr[u] + len(l)
    items = np.fromiter((i for l in tr for i in l), np.int32, int(indptr[-1]))
    csr = (torch.as_tensor(indptr, device="cuda"), torch.as_tensor(items, device="cuda"))
    S = torch.as_tensor(sc, device="cuda").clone()
    idx, val, flag = e.topk(0, U, S, csr, K)
    idx, val, flag = idx.cpu().numpy(), val.cpu().numpy(), flag.cpu().numpy()
    assert flag.sum() == 0
    for u in range(U):
        row = sc[u].copy()
        row[tr[u]] = -np.inf
        want = row.argsort()[-K:][::-1]
        assert np.array_equal(idx[u], want), u
        assert np.array_equal(val[u], row[want]), u
    assert torch.isinf(S[0, tr[0]]).all() if tr[0] else True               # masked in place, like the reference (:233)
    e.sync_check()


def test_topk_flags_rows_that_depend_on_tie_order():
    from fashionvisualexpl_recommend_amd.engine import Engine
    U, I, K = 6, 40, 5
    sc = np.arange(U * I, dtype=np.float32).reshape(U, I)
    sc[1, 10] = sc[1, 39]                       # tie INSIDE the list
    sc[2, :] = 1.0                              # everything ties
    sc[3, 34] = sc[3, 35]                       # tie at the boundary (5th and 6th largest)
    tr = [[], [], [], [], list(range(37)), []]  # user 4: only 3 unmasked items < K
    e = Engine(model="bprmf", num_users=U, num_items=I, embed_k=4, optimizer="sgd", max_batch=8)
    e.bind(Gu=np.zeros((U, 4), np.float32), Gi=np.zeros((I, 4), np.float32), Bi=np.zeros(I, np.float32))
    indptr = np.cumsum([0] + [len(l) for l in tr]).astype(np.int64)
    items = np.array([i for l in tr for i in l] or [0], np.int32)
    csr = (torch.as_tensor(indptr, device="cuda"), torch.as_tensor(items, device="cuda"))
    idx, val, flag = e.topk(0, U, torch.as_tensor(sc, device="cuda").clone(), csr, K)
    assert flag.cpu().numpy().tolist() == [0, 1, 1, 1, 1, 0]
    assert idx[0].cpu().numpy().tolist() == [39, 38, 37, 36, 35]
    e.sync_check()


def test_device_eval_counts_a_duplicated_train_row_once(tmp_path):
    """ADVICE: the reference masks with set(training_list[user]) (Evaluator.py:41); a train file with a repeated (u, i) row
    must not change nneg / position / auc on the device path.  Device metrics == host metrics (boolean mask)."""
    U, I, K = 40, 60, 5
    rs = np.random.RandomState(4)
    sc = rs.standard_normal((U, I)).astype(np.float32)
    tr = [rs.choice(I, 8, replace=False).tolist() for _ in range(U)]
    for u in range(0, U, 3):
        tr[u] = tr[u] + [tr[u][0], tr[u][1], tr[u][0]]                     # repeated interactions
    te = [[int(rs.choice([i for i in range(I) if i not in tr[u]]))] for u in range(U)]
    va = [[int(rs.choice([i for i in range(I) if i not in tr[u] and i != te[u][0]]))] for u in range(U)]
    data = Namespace(training_list=tr, validation_list=va, test_list=te, num_users=U, num_items=I,
                     params=Namespace(batch_eval=128))
    ev = Evaluator(_ScoreModel(data, sc), data, K, user_block=16)
    got = ev.metrics()
    ev.force_host = True
    want = ev.metrics()
    for k in want:
        assert got[k] == pytest.approx(want[k], abs=1e-12), k
    ev.model.engine.sync_check()


def test_device_eval_reports_out_of_range_held_out_item():
    from fashionvisualexpl_recommend_amd import _ffi
    from fashionvisualexpl_recommend_amd.engine import Engine
    U, I = 4, 10
    e = Engine(model="bprmf", num_users=U, num_items=I, embed_k=4, optimizer="sgd", max_batch=8)
    e.bind(Gu=np.zeros((U, 4), np.float32), Gi=np.zeros((I, 4), np.float32), Bi=np.zeros(I, np.float32))
    dev = lambda a: torch.as_tensor(a, device="cuda")
    trc = (dev(np.zeros(U + 1, np.int64)), dev(np.zeros(1, np.int32)))
    evc = (dev(np.arange(U + 1, dtype=np.int64)), dev(np.array([1, 2, 99, 3], np.int32)))   # item 99 >= I
    e.eval_users(0, U, dev(np.zeros((U, I), np.float32)), trc, evc, 3)
    with pytest.raises(_ffi.BprxError) as ei:
        e.sync_check()
    assert ei.value.code == _ffi.E_RANGE
