"""Evaluator: mirror of the reference's src/recommender/Evaluator.py:131-239 (metric definitions :82-128).

Same constructor, `eval(epoch, results, epoch_text, start_time)` and `store_recommendation(path)`; same result
keys (hr_v ... ndcg_t, including the reference's 'auc_t': auc_v aliasing at :220).  Scores come from the
model's predict_all() (libbprx bprx_score_block); the ranking arithmetic below is host NumPy, evaluated on
blocks of users so that U x I is never resident at once.  No multiprocessing Pool (the reference forks one in
the model constructor, Evaluator.py:21; forking after HIP initialisation is not safe).

Metric definitions restated (Evaluator.py:82-128), for user u with train set T, eval items e_1..e_n:
  negatives = all items - T - {e}; position = sum_t #(neg >= score(e_t));
  auc = 1 - position/(|neg| n); top-K = stable descending order of (negatives by ascending id, then e_1..e_n);
  hr = any e in top-K; prec = hits/min(K,|cand|); rec = hits/n; ndcg = ln2/ln(position+2) if position < K else 0.
"""
import datetime
import math
from time import time

import numpy as np


def _eval_block(scores, u0, train, evl, K):
    """Rows of one user block -> list of (hr, prec, rec, auc, ndcg) tuples (users with empty eval lists skipped)."""
    nb, I = scores.shape
    out = []
    cand = np.ones((nb, I), dtype=bool)
    for r in range(nb):
        cand[r, train[u0 + r]] = False
    single = all(len(evl[u0 + r]) <= 1 for r in range(nb))
    if single:
        rows = np.array([r for r in range(nb) if len(evl[u0 + r]) == 1], dtype=np.int64)
        if rows.size == 0:
            return out
        ev = np.array([evl[u0 + r][0] for r in rows], dtype=np.int64)
        c = cand[rows]
        c[np.arange(rows.size), ev] = False
        sp = scores[rows, ev]
        position = ((scores[rows] >= sp[:, None]) & c).sum(axis=1)
        nneg = c.sum(axis=1)
        for p, n in zip(position.tolist(), nneg.tolist()):
            topn = min(K, n + 1)
            hit = 1 if p < topn else 0
            auc = 1 - (p / (n * 1))
            out.append((float(hit), hit / topn, hit / 1, auc, math.log(2) / math.log(p + 2) if p < K else 0))
        return out
    for r in range(nb):
        ev = list(evl[u0 + r])
        if len(ev) == 0:
            continue
        c = cand[r].copy()
        c[ev] = False
        s = scores[r]
        neg = s[c]
        sp = s[ev]
        position = int(sum((neg >= sp[t]).sum() for t in range(len(ev))))
        auc = 1 - (position / (len(neg) * len(ev)))
        topn = min(K, len(neg) + len(ev))
        hits = 0
        for t in range(len(ev)):
            rank = int((neg >= sp[t]).sum()) + sum(1 for q in range(len(ev))
                                                   if q != t and (sp[q] > sp[t] or (sp[q] == sp[t] and q < t)))
            hits += 1 if rank < topn else 0
        out.append((1. if hits > 0 else 0., hits / topn, hits / len(ev), auc,
                    math.log(2) / math.log(position + 2) if position < K else 0))
    return out


class Evaluator:
    def __init__(self, model, data, k, user_block=4096):
        self.data = data
        self.batch_eval = getattr(data.params, "batch_eval", 128)
        self.k = k
        self.model = model
        self.user_block = user_block

    def _score_blocks(self):
        U = self.model.data.num_users
        for u0 in range(0, U, self.user_block):
            u1 = min(U, u0 + self.user_block)
            yield u0, self.model.predict_block(u0, u1)

    # ---- device path (libbprx bprx_score_block + bprx_eval_users): used whenever the model runs on the engine ----
    def _device_csr(self, lists, device, dedup=False):
        import torch
        if dedup:       # the reference masks with set(training_list[user]) (Evaluator.py:41): a repeated row counts once
            lists = [list(dict.fromkeys(l)) for l in lists]
        indptr = np.zeros(len(lists) + 1, dtype=np.int64)
        for u, l in enumerate(lists):
            indptr[u + 1] = indptr[u] + len(l)
        items = np.fromiter((i for l in lists for i in l), dtype=np.int32, count=int(indptr[-1]))
        if items.size == 0:
            items = np.zeros(1, np.int32)
        return torch.as_tensor(indptr, device=device), torch.as_tensor(items, device=device)

    def _metrics_device(self):
        import torch
        eng = self.model.engine
        self._metrics_device_csr()
        U = self.model.data.num_users
        rows = {"test": [], "val": []}
        for u0 in range(0, U, self.user_block):
            u1 = min(U, u0 + self.user_block)
            sc = eng.score_block(u0, u1)
            for key in ("test", "val"):
                if self._csr[key] is not None:
                    rows[key].append(eng.eval_users(u0, u1, sc, self._csr["train"], self._csr[key], self.k))
        out = {}
        for key, suf in (("test", "_t"), ("val", "_v")):
            if not rows[key]:
                continue
            r = torch.cat(rows[key]).cpu().numpy()
            if (r[:, 0] == -2).any():
                return None                                # > 32 held-out items for some user: host path
            r = r[r[:, 0] >= 0]
            hr, p, rr, auc, ndcg = r.mean(axis=0).tolist()
            out.update({"hr" + suf: hr, "p" + suf: p, "r" + suf: rr, "auc" + suf: auc, "ndcg" + suf: ndcg})
        return out

    def metrics(self):
        """The ten means of Evaluator.py:189-193 with the TRUE auc_t (eval() applies the reference's aliasing)."""
        if getattr(self.model, "engine", None) is not None and not getattr(self, "force_host", False):
            m = self._metrics_device()
            if m is not None:
                return m
        res_t, res_v = [], []
        val = bool(self.data.validation_list)
        for u0, sc in self._score_blocks():
            res_t += _eval_block(sc, u0, self.data.training_list, self.data.test_list, self.k)
            if val:
                res_v += _eval_block(sc, u0, self.data.training_list, self.data.validation_list, self.k)
        hr_t, p_t, r_t, auc_t, ndcg_t = np.array(res_t).mean(axis=0).tolist()
        out = {"hr_t": hr_t, "p_t": p_t, "r_t": r_t, "auc_t": auc_t, "ndcg_t": ndcg_t}
        if val:
            hr_v, p_v, r_v, auc_v, ndcg_v = np.array(res_v).mean(axis=0).tolist()
            out.update({"hr_v": hr_v, "p_v": p_v, "r_v": r_v, "auc_v": auc_v, "ndcg_v": ndcg_v})
        return out

    def eval(self, epoch=0, results=None, epoch_text='', start_time=0):
        """Evaluator.py:149-223."""
        if results is None:
            results = {}
        eval_start_time = time()
        m = self.metrics()
        z = lambda k_: m.get(k_, 0.0)     # the reference crashes here without a validation set (:179,:195)
        print_results = \
            "%s \tTrain Time: %s \tEvaluation Time: %s" \
            "\nMetrics@%d (Validation)\n\t\tHR\tPrec\tRec\tAUC\tnDCG\n\t\t%f\t%f\t%f\t%f\t%f" \
            "\nMetrics@%d (Test)\n\t\tHR\tPrec\tRec\tAUC\tnDCG\n\t\t%f\t%f\t%f\t%f\t%f\n" % (
                epoch_text,
                datetime.timedelta(seconds=(time() - start_time)),
                datetime.timedelta(seconds=(time() - eval_start_time)),
                self.k, z("hr_v"), z("p_v"), z("r_v"), z("auc_v"), z("ndcg_v"),
                self.k, z("hr_t"), z("p_t"), z("r_t"), z("auc_t"), z("ndcg_t"))
        print(print_results)
        results[epoch] = {
            'hr_v': z("hr_v"), 'auc_v': z("auc_v"), 'p_v': z("p_v"), 'r_v': z("r_v"), 'ndcg_v': z("ndcg_v"),
            'hr_t': z("hr_t"), 'auc_t': z("auc_v"), 'p_t': z("p_t"), 'r_t': z("r_t"), 'ndcg_t': z("ndcg_t")
        }                                   # 'auc_t': auc_v is the reference's own aliasing (Evaluator.py:220)
        return print_results

    def _store_recommendation_device(self, out):
        """bprx_score_block + bprx_topk per user block; only rows whose list depends on the order of EQUAL scores (flagged
        by the kernel: the reference's order there is numpy's unstable argsort) are redone on the host, from the row the
        kernel has already masked."""
        eng = self.model.engine
        self._metrics_device_csr()
        U = self.model.data.num_users
        for u0 in range(0, U, self.user_block):
            u1 = min(U, u0 + self.user_block)
            sc = eng.score_block(u0, u1)
            idx, val, flag = eng.topk(u0, u1, sc, self._csr["train"], self.k)
            idx, val, flag = idx.cpu().numpy(), val.cpu().numpy(), flag.cpu().numpy()
            redo = np.nonzero(flag)[0]
            rows = {int(r): sc[int(r)].cpu().numpy() for r in redo}          # few: ties are rare in real-valued scores
            for r in range(u1 - u0):
                u = u0 + r
                if r in rows:
                    row = rows[r]
                    top_k_id = row.argsort()[-self.k:][::-1]
                    top_k_score = row[top_k_id]
                else:
                    kk = min(self.k, idx.shape[1], sc.shape[1])
                    top_k_id, top_k_score = idx[r, :kk], val[r, :kk]
                for i, value in enumerate(top_k_id):
                    out.write(str(u) + '\t' + str(value) + '\t' + str(top_k_score[i]) + '\n')

    def _metrics_device_csr(self):
        if getattr(self, "_csr", None) is None:
            eng = self.model.engine
            U = self.model.data.num_users
            pad = lambda l: list(l[:U]) + [[] for _ in range(U - len(l))]
            self._csr = {"train": self._device_csr(pad(self.data.training_list), eng.device, dedup=True),
                         "test": self._device_csr(pad(self.data.test_list), eng.device),
                         "val": self._device_csr(pad(self.data.validation_list), eng.device)
                         if self.data.validation_list else None}

    def store_recommendation(self, path=""):
        """Evaluator.py:225-239: per user mask train items, top-k by argsort, 'u\\titem\\tscore' rows."""
        if getattr(self.model, "engine", None) is not None and not getattr(self, "force_host", False) and self.k <= 1024:
            with open(path, 'w') as out:
                self._store_recommendation_device(out)
            return
        with open(path, 'w') as out:
            for u0, sc in self._score_blocks():
                for r in range(sc.shape[0]):
                    u = u0 + r
                    row = sc[r]
                    row[self.data.training_list[u]] = -np.inf
                    top_k_id = row.argsort()[-self.k:][::-1]
                    top_k_score = row[top_k_id]
                    for i, value in enumerate(top_k_id):
                        out.write(str(u) + '\t' + str(value) + '\t' + str(top_k_score[i]) + '\n')
