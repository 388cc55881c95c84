"""Multi-GPU training from the train_rec.py surface: one process per GPU (torch.distributed,
backend "nccl" == RCCL on ROCm; "gloo" for CPU-side tests and one-GPU rehearsals).

--shard item  (VBPR, BASELINE.json configs[3]): items -- Gi, Bi and the rows of the feature table F -- are range-partitioned
  over the ranks and never cross xGMI; every rank holds the full user tables (ReplicatedUserVBPR: one all-gather of the
  batch's distinct users' gradient rows per step; E / beta' summed in rank order inside that message or, with
  --dense_reduce allreduce, by an RCCL all-reduce).  A rank trains on the interactions whose POSITIVE item it owns and
  draws negatives from its own item range (north_star: "negative sampling stays GPU-local"; the reference draws them from
  all items, dataset.py:101, so multi-rank runs follow the reference statistically, not triplet for triplet).
  Sharded FEATURE INGESTION: a rank memory-maps cnn_features_{model}_{layer}.npy and reads only its item rows; the
  reference's GLOBAL max-abs normalisation (visual_loader_mixin.py:30) becomes a max-abs per shard + all-reduce(MAX).
--shard user  (BPRMF, configs[2]): users are range-partitioned, item rows travel by all-to-all (UserShardedBPRMF).

Evaluation gathers the ranks' score columns per user block on rank 0 and runs the reference's metric definitions there
(evaluator._eval_block); outputs (epoch lines, results dict, recs TSV) are written by rank 0 only.
"""
import os
from time import time

import numpy as np
import torch
import torch.distributed as dist

from . import configs
from .synth import glorot_uniform


def item_range(num_items, rank, world):
    """[lo, hi) of rank's item shard: equal shards of ceil(I / world), the last one shorter."""
    sh = (num_items + world - 1) // world
    return min(num_items, rank * sh), min(num_items, (rank + 1) * sh)


def load_feature_shard(path, lo, hi, group=None, features=None):
    """visual_loader_mixin.py:22-31 for ONE item shard: rows [lo, hi) of the .npy (memory-mapped: the other shards' bytes are
    never read), divided by the GLOBAL max-abs -- max over this shard, then all-reduce(MAX) over the ranks.  Returns
    (float32 [hi-lo, D] array, global max-abs).  `features`: an in-memory [I, D] array instead of the file (tests)."""
    src = np.load(path, mmap_mode="r") if features is None else features
    part = np.asarray(src[lo:hi], dtype=np.float64 if src.dtype == np.float64 else np.float32)
    m = torch.tensor([float(np.max(np.abs(part))) if part.size else 0.0], dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
    gmax = float(m.item())
    return (part / gmax).astype(np.float32), gmax              # same operation order as the reference: f / max|f|


def local_positive_lists(training_list, num_users, lo, hi):
    """Per user, the training items inside [lo, hi) as shard-local ids (sorted: the device sampler bisects them)."""
    out = []
    for u in range(num_users):
        l = training_list[u] if u < len(training_list) else []
        out.append(sorted(i - lo for i in l if lo <= i < hi))
    return out


class ShardedVBPR:
    """Item-sharded VBPR behind the reference's model surface (train(), predict_block(), evaluator)."""

    def __init__(self, data, params, features=None, group=None):
        from .dist import ReplicatedUserVBPR
        from .engine import EpochWalkSampler
        self.data, self.params, self.group = data, params, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.num_users, self.num_items = data.num_users, data.num_items
        self.lo, self.hi = item_range(self.num_items, self.rank, self.world)
        k, d = params.embed_k, params.embed_d
        path = configs.cnn_features_path(params.dataset, getattr(params, "cnn_model", "vgg19"),
                                         getattr(params, "output_layer", "fc2"))
        F, self.feat_max = load_feature_shard(path, self.lo, self.hi, group, features)
        D = F.shape[1]
        # identical initial values on every rank (one seeded generator, the creation order of VBPR.py:44-54); a rank keeps
        # its item rows of Gi / Bi and every row of the replicated tables
        rs = np.random.RandomState(getattr(params, "init_seed", 0))
        Gu, Gi = glorot_uniform(rs, self.num_users, k), glorot_uniform(rs, self.num_items, k)
        Bp, Tu, E = glorot_uniform(rs, D, 1).reshape(-1), glorot_uniform(rs, self.num_users, d), glorot_uniform(rs, D, d)
        c = lambda a: torch.as_tensor(np.ascontiguousarray(a))
        self.batch = int(params.batch_size)
        self.m = ReplicatedUserVBPR(self.rank, self.world, c(Gu), c(Tu), c(Gi[self.lo:self.hi]),
                                    c(np.zeros(self.hi - self.lo, np.float32)), c(F), c(E), c(Bp), params.lr, params.reg,
                                    max_batch=self.batch, user_cap=min(self.batch, self.num_users),
                                    feat_dtype=getattr(params, "dtype", "fp32"), group=group,
                                    optimizer=getattr(params, "optimizer", "adam_tf23"),
                                    dense_reduce=getattr(params, "dense_reduce", "gather"))
        self.engine = self.m.eng
        lists = local_positive_lists(data.training_list, self.num_users, self.lo, self.hi)
        self.local_pos = sum(len(l) for l in lists)
        n = torch.tensor([self.local_pos], dtype=torch.int64)
        dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
        self.steps_per_epoch = max(1, int(n.item()) // self.batch)     # every rank steps as often as the fullest shard
        self.sampler = EpochWalkSampler(lists, self.hi - self.lo, device=self.engine.device,
                                        seed=getattr(params, "init_seed", 0) + 7919 * self.rank) if self.local_pos else None
        self.directory_parameters = f'batch_{params.batch_size}-D_{d}-K_{k}-lr_{params.lr}-reg_{params.reg}-W_{self.world}'

    # ---- scores: every rank's item columns, gathered on rank 0 ------------------------------------------------
    def predict_block(self, u0, u1):
        loc = self.engine.score_block(u0, u1).cpu()                    # [nb, I_shard]
        sh = (self.num_items + self.world - 1) // self.world
        pad = torch.zeros((u1 - u0, sh), dtype=torch.float32)
        pad[:, :loc.shape[1]] = loc
        parts = [torch.empty_like(pad) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(pad, parts, dst=0, group=self.group)
        if self.rank != 0:
            return None
        return torch.cat(parts, dim=1)[:, :self.num_items].numpy()

    def metrics(self, K, user_block=4096):
        from .evaluator import _eval_block
        res_t, res_v = [], []
        val = bool(self.data.validation_list)
        for u0 in range(0, self.num_users, user_block):
            u1 = min(self.num_users, u0 + user_block)
            sc = self.predict_block(u0, u1)
            if self.rank == 0:
                res_t += _eval_block(sc, u0, self.data.training_list, self.data.test_list, K)
                if val:
                    res_v += _eval_block(sc, u0, self.data.training_list, self.data.validation_list, K)
        if self.rank != 0:
            return None
        out = dict(zip(("hr_t", "p_t", "r_t", "auc_t", "ndcg_t"), np.array(res_t).mean(axis=0).tolist()))
        if val:
            out.update(zip(("hr_v", "p_v", "r_v", "auc_v", "ndcg_v"), np.array(res_v).mean(axis=0).tolist()))
        return out

    # ---- BPRMF.py:127-165 with one step = one global batch of world x batch_size triplets -----------------------
    def train(self):
        results, dev = {}, self.engine.device
        empty = torch.zeros(0, dtype=torch.int32, device=dev)
        for it in range(1, self.params.epochs + 1):
            start, loss = time(), 0.0
            for _ in range(self.steps_per_epoch):
                u, i, j = self.sampler.sample(self.batch) if self.sampler is not None else (empty, empty, empty)
                loss += float(self.m.step(u, i, j, want_loss=True).item())
            m = self.metrics(self.params.top_k)
            if self.rank == 0:
                results[it] = {"hr_v": m.get("hr_v", 0.0), "auc_v": m.get("auc_v", 0.0), "p_v": m.get("p_v", 0.0),
                               "r_v": m.get("r_v", 0.0), "ndcg_v": m.get("ndcg_v", 0.0), "hr_t": m["hr_t"],
                               "auc_t": m.get("auc_v", 0.0), "p_t": m["p_t"], "r_t": m["r_t"], "ndcg_t": m["ndcg_t"]}
                print('Epoch {0}/{1} \tLoss (rank 0 shard): {2:.3f} \tTrain+Eval Time: {3:.1f}s \tHR@{4} (Test): {5:.4f} '
                      '\tnDCG (Test): {6:.4f}'.format(it, self.params.epochs, loss / self.steps_per_epoch, time() - start,
                                                       self.params.top_k, m["hr_t"], m["ndcg_t"]))
        self.engine.sync_check()
        self.results = results
        return results
