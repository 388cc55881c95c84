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

Evaluation runs on the devices: every rank scores its own item columns per user block and counts there (bprx_eval_pos /
bprx_eval_counts: what the reference's metrics count is additive over item shards); two small all-reduces per block carry the
held-out items' scores and the counts, and every rank finishes the same metrics (bprx_eval_finish) -- equal to the single-GPU
evaluator on the concatenated score row.  Rank 0 writes the reference's outputs (epoch lines, results pickle, weights of the
replicated tables + the gathered item shards, recs TSV of the last and of the best epoch; BPRMF.py:152-183).
"""
import contextlib
import io
import os
import pickle
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
        from .evaluator import Evaluator
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
        self.host_staged = dist.get_backend(group) != "nccl"              # gloo (tests, one-GPU rehearsals): collectives on host copies
        lists = local_positive_lists(data.training_list, self.num_users, self.lo, self.hi)
        self.local_pos = sum(len(l) for l in lists)
        n = torch.tensor([self.local_pos], dtype=torch.int64)
        dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
        self.steps_per_epoch = max(1, int(n.item()) // self.batch)     # every rank steps as often as the fullest shard
        # a rank whose shard holds no positive still takes every step (with an empty batch: dist.ReplicatedUserVBPR.step)
        self.sampler = EpochWalkSampler(lists, self.hi - self.lo, device=self.engine.device,
                                        seed=getattr(params, "init_seed", 0) + 7919 * self.rank) if self.local_pos else None
        if self.sampler is not None:
            self.sampler.feeds(self.engine)                # (the step hands the sampler's index arrays to the engine unchanged)
        self.directory_parameters = f'batch_{params.batch_size}-D_{d}-K_{k}-lr_{params.lr}-reg_{params.reg}-W_{self.world}'
        # device CSRs with GLOBAL item ids for the shard-additive evaluation (the Evaluator's own helpers)
        ev = Evaluator(self, data, params.top_k)
        ev._metrics_device_csr()
        self._csr = ev._csr
        self.evaluator = _ShardedEvaluator(self, data, params.top_k)

    # ---- collectives on device tensors (RCCL), or through host copies under gloo ----------------------------------
    def _all_reduce_sum(self, t):
        if self.world == 1:
            return t
        if self.host_staged:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)
        return t

    # ---- scores: every rank's item columns, gathered on rank 0 (output files only: never inside the training loop) ------
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
        """The ten means of Evaluator.py:189-193 on the devices; every rank returns the same dict."""
        eng, out = self.engine, {}
        rows = {"test": [], "val": []}
        for u0 in range(0, self.num_users, user_block):
            u1 = min(self.num_users, u0 + user_block)
            sc = eng.score_block(u0, u1)                               # this rank's columns only
            for key in ("test", "val"):
                if self._csr[key] is None:
                    continue
                sp = self._all_reduce_sum(eng.eval_pos(u0, u1, sc, self.lo, self.num_items, self._csr[key]))
                cn = self._all_reduce_sum(eng.eval_counts(u0, u1, sc, self.lo, self.num_items, self._csr["train"],
                                                          self._csr[key], sp))
                rows[key].append(eng.eval_finish(u0, u1, self.num_items, self._csr[key], sp, cn, K))
        for key, suf in (("test", "_t"), ("val", "_v")):
            if not rows[key]:
                continue
            r = torch.cat(rows[key]).cpu().numpy()
            if (r[:, 0] == -2).any():
                raise NotImplementedError("more than 32 held-out items per user: not supported by the sharded evaluator")
            r = r[r[:, 0] >= 0]
            hr, p, rr, auc, ndcg = r.mean(axis=0).tolist()
            out.update({"hr" + suf: hr, "p" + suf: p, "r" + suf: rr, "auc" + suf: auc, "ndcg" + suf: ndcg})
        return out

    # ---- snapshots: the reference deep-copies / checkpoints the whole model (BPRMF.py:156-160,177-179) ---------------
    def local_state(self):
        """This rank's tensors (replicated tables, its item rows, Adam slots) -- what `best model` tracking keeps per rank."""
        sd = {n: v.detach().clone() for n, v in self.engine.t.items() if n != "F"}
        sd["adam_step"] = self.engine.adam_step
        return sd

    def load_local_state(self, sd):
        for n, v in sd.items():
            if n == "adam_step":
                self.engine.adam_step = v
            else:
                self.engine.t[n].copy_(v)
        self.engine.tables_dirty()

    def full_state(self, sd=None):
        """The whole model on rank 0 (None elsewhere): replicated tables as they are, item-sharded ones gathered by rows."""
        sd = self.local_state() if sd is None else sd
        sh = (self.num_items + self.world - 1) // self.world
        out = {}
        for n, v in sd.items():
            if n == "adam_step" or not (n.endswith("Gi") or n.endswith("Bi")):
                out[n] = v if n == "adam_step" else v.cpu()
                continue
            loc = v.cpu()
            pad = torch.zeros((sh,) + tuple(loc.shape[1:]), dtype=loc.dtype)
            pad[:loc.shape[0]] = loc
            parts = [torch.empty_like(pad) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(pad, parts, dst=0, group=self.group)
            if self.rank == 0:
                out[n] = torch.cat(parts, dim=0)[:self.num_items]
        return out if self.rank == 0 else None

    def store_recommendation(self, path):
        """Evaluator.py:225-239 on the gathered score blocks (rank 0 writes; every rank takes part in the gathers)."""
        K = self.params.top_k
        out = open(path, 'w') if self.rank == 0 else None
        try:
            for u0 in range(0, self.num_users, 4096):
                u1 = min(self.num_users, u0 + 4096)
                sc = self.predict_block(u0, u1)
                if self.rank != 0:
                    continue
                for r in range(sc.shape[0]):
                    u, row = u0 + r, sc[r]
                    row[self.data.training_list[u]] = -np.inf
                    top_k_id = row.argsort()[-K:][::-1]
                    for i, value in enumerate(top_k_id):
                        out.write(str(u) + '\t' + str(value) + '\t' + str(row[top_k_id][i]) + '\n')
        finally:
            if out is not None:
                out.close()

    # ---- BPRMF.py:127-192 with one step = one global batch of world x batch_size triplets ---------------------------
    def train(self):
        params, dev = self.params, self.engine.device
        max_metrics = {'hr': 0, 'p': 0, 'r': 0, 'auc': 0, 'ndcg': 0}
        best_state, best_epoch, best_epoch_print = None, getattr(params, "restore_epochs", 1), 'No best epoch found!'
        results = {}
        rec = getattr(params, "rec", "vbpr")
        wdir = os.path.join(configs.weight_dir(), params.dataset, rec)
        rdir = os.path.join(configs.results_dir(), params.dataset, rec)
        if self.rank == 0:
            os.makedirs(wdir, exist_ok=True)
            os.makedirs(rdir, exist_ok=True)
        empty = torch.zeros(0, dtype=torch.int32, device=dev)
        loss_buf = torch.zeros(self.steps_per_epoch, dtype=torch.float32, device=dev)      # read once per epoch: no per-step sync
        verbose = getattr(params, "verbose", -1)
        best_metric = getattr(params, "best_metric", "ndcg")
        if self.rank == 0:
            print('Start training...')
        for it in range(1, params.epochs + 1):
            start = time()
            for s in range(self.steps_per_epoch):
                u, i, j = self.sampler.sample(self.batch) if self.sampler is not None else (empty, empty, empty)
                self.m.step(u, i, j, loss_out=loss_buf, loss_index=s)
            loss = float(loss_buf.double().sum().item())
            epoch_text = 'Epoch {0}/{1} \tLoss (rank 0 shard): {2:.3f}'.format(it, params.epochs, loss / self.steps_per_epoch)
            epoch_print = self.evaluator.eval(it, results, epoch_text, start)               # identical on every rank
            for metric in max_metrics.keys():                                               # BPRMF.py:152-156
                if max_metrics[metric] <= results[it][metric + '_v']:
                    max_metrics[metric] = results[it][metric + '_v']
                    if metric == best_metric:
                        best_epoch, best_state, best_epoch_print = it, self.local_state(), epoch_print
            if (it % verbose == 0 or it == 1) and verbose != -1:
                full = self.full_state()
                if self.rank == 0:
                    torch.save(full, os.path.join(wdir, f'weights-{it}-{self.directory_parameters}.pt'))
        self.engine.sync_check()
        last = params.epochs
        if self.rank == 0:
            print('Training end...')
        self.store_recommendation(os.path.join(rdir, f'recs-{last}-{self.directory_parameters}.tsv'))
        if self.rank == 0:
            with open(os.path.join(rdir, f'results-metrics-{self.directory_parameters}') + '.pkl', 'wb') as f:
                pickle.dump(results, f)                                                     # utils/write.py:14-22
            print("Store Best Model at Epoch {0}".format(best_epoch))
            print(best_epoch_print)
        last_state = self.local_state()
        if best_state is not None:
            full = self.full_state(best_state)
            if self.rank == 0:
                torch.save(full, os.path.join(wdir, f'best-weights-{best_epoch}-{self.directory_parameters}.pt'))
            self.load_local_state(best_state)
        self.store_recommendation(os.path.join(rdir, f'best-recs-{best_epoch}-{self.directory_parameters}.tsv'))
        self.load_local_state(last_state)
        if self.rank == 0:
            print('End Store Best Model!')
            print('Best Values for Each Metric:\nHR\tPrec\tRec\tAUC\tnDCG\n{}\t{}\t{}\t{}\t{}\n'.format(
                max_metrics['hr'], max_metrics['p'], max_metrics['r'], max_metrics['auc'], max_metrics['ndcg']))
        self.results = results
        return results


class ShardedBPRMF:
    """User-sharded BPRMF behind the reference's model surface (BASELINE.json configs[2]; train_rec --world_size N --shard user
    --rec bprmf).  Rank r owns the users [r*ush, (r+1)*ush) -- their Gu rows, their training positives, their
    evaluation -- and the item rows [r*ish, (r+1)*ish) of Gi / Bi.  A step fetches the rows of the batch's positive and negative
    items from their owners and returns their gradients by fixed-capacity all-to-alls (dist.UserShardedBPRMF: routing in HIP
    kernels, no host synchronisation); negatives are drawn from ALL items, as the reference does (dataset.py:101).
    Evaluation: the item shards are all-gathered once per epoch, every rank scores and evaluates ITS users on the device
    (bprx_score_block + bprx_eval_users), and the per-user rows are gathered in user order -- the means equal the single-GPU
    evaluator's.  Rank 0 writes the reference's outputs (BPRMF.py:152-183).  --optimizer sgd: the owners add the routed gradient
    rows into their shard; adam_tf23 (the reference's optimizer): they sum them into a gradient table and take the Adam step of
    the whole shard (dist.UserShardedBPRMF)."""

    def __init__(self, data, params, group=None):
        from .dist import UserShardedBPRMF, shard_size
        from .engine import Engine, EpochWalkSampler
        from .evaluator import Evaluator
        self.data, self.params, self.group = data, params, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.num_users, self.num_items = data.num_users, data.num_items
        k = params.embed_k
        self.ush, self.ish = shard_size(self.num_users, self.world), shard_size(self.num_items, self.world)
        self.u0, self.u1 = min(self.num_users, self.rank * self.ush), min(self.num_users, (self.rank + 1) * self.ush)
        self.i0, self.i1 = min(self.num_items, self.rank * self.ish), min(self.num_items, (self.rank + 1) * self.ish)
        rs = np.random.RandomState(getattr(params, "init_seed", 0))             # BPRMF.py:48-50 creation order, every rank alike
        Gu, Gi = glorot_uniform(rs, self.num_users, k), glorot_uniform(rs, self.num_items, k)
        c = lambda a: torch.as_tensor(np.ascontiguousarray(a))
        pad_rows = lambda a, n: np.concatenate([a, np.zeros((n - a.shape[0],) + a.shape[1:], a.dtype)]) if a.shape[0] < n else a
        self.batch = int(params.batch_size)
        nu = max(1, self.u1 - self.u0)
        # (shards are padded to the full shard size: the row routing addresses rows_per_rank = ish rows on every rank)
        self.m = UserShardedBPRMF(self.rank, self.world, self.num_items, c(pad_rows(Gu[self.u0:self.u1], nu)),
                                  c(pad_rows(Gi[self.i0:self.i1], self.ish)), c(np.zeros(self.ish, np.float32)), params.lr, params.reg,
                                  max_batch=self.batch, group=group, optimizer=getattr(params, "optimizer", "sgd"))
        self.engine = self.m.eng
        dev = self.engine.device
        self.host_staged = dist.get_backend(group) != "nccl"
        lists = [sorted(l) for l in (list(data.training_list[u]) if u < len(data.training_list) else [] for u in range(self.u0, self.u1))]
        self.local_pos = sum(len(l) for l in lists)
        n = torch.tensor([self.local_pos], dtype=torch.int64)
        dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
        self.steps_per_epoch = max(1, int(n.item()) // self.batch)
        self.sampler = EpochWalkSampler(lists, self.num_items, device=dev,
                                        seed=getattr(params, "init_seed", 0) + 7919 * self.rank) if self.local_pos else None
        self.directory_parameters = f'batch_{params.batch_size}-K_{k}-lr_{params.lr}-reg_{params.reg}-W_{self.world}'
        # evaluation engine: this rank's users x ALL items (the gathered item shards are copied into its tables once per epoch)
        self.Gi_full = torch.zeros((self.num_items, k), dtype=torch.float32, device=dev)
        self.Bi_full = torch.zeros(self.num_items, dtype=torch.float32, device=dev)
        self.ev_eng = Engine(model="bprmf", num_users=nu, num_items=self.num_items, embed_k=k, optimizer="sgd", max_batch=256,
                             device=dev.index).bind(Gu=self.engine.t["Gu"], Gi=self.Gi_full, Bi=self.Bi_full)
        pad = lambda l: [list(l[u]) if u < len(l) else [] for u in range(self.u0, self.u0 + nu)]
        ev = Evaluator(self, data, params.top_k)
        self._csr = {"train": ev._device_csr(pad(data.training_list), dev, dedup=True), "test": ev._device_csr(pad(data.test_list), dev),
                     "val": ev._device_csr(pad(data.validation_list), dev) if data.validation_list else None}
        self.evaluator = _ShardedEvaluator(self, data, params.top_k)

    def _gather_items(self):
        """Item shards of every rank -> this rank's full Gi / Bi (evaluation, snapshots)."""
        for shard, full in ((self.m.Gi_shard, self.Gi_full), (self.m.Bi_col, self.Bi_full.view(-1, 1))):
            parts = [torch.empty_like(shard.cpu() if self.host_staged else shard) for _ in range(self.world)]
            dist.all_gather(parts, shard.cpu() if self.host_staged else shard, group=self.group)
            full.copy_(torch.cat(parts, dim=0)[:self.num_items].to(full.device))
        self.ev_eng.t["Gu"].copy_(self.engine.t["Gu"]) if self.ev_eng.t["Gu"].data_ptr() != self.engine.t["Gu"].data_ptr() else None
        self.ev_eng.tables_dirty()

    def _user_rows(self, K):
        """[local users, 5] metric rows per eval list, on the device."""
        nu, out = self.u1 - self.u0, {}
        for key in ("test", "val"):
            if self._csr[key] is None or nu <= 0:
                out[key] = torch.zeros((0, 5), dtype=torch.float64)
                continue
            rows = []
            for b0 in range(0, nu, 4096):
                b1 = min(nu, b0 + 4096)
                sc = self.ev_eng.score_block(b0, b1)
                rows.append(self.ev_eng.eval_users(b0, b1, sc, self._csr["train"], self._csr[key], K))
            out[key] = torch.cat(rows).cpu()
        return out

    def metrics(self, K):
        self._gather_items()
        mine, out = self._user_rows(K), {}
        for key, suf in (("test", "_t"), ("val", "_v")):
            if self._csr[key] is None:
                continue
            parts = [None] * self.world
            dist.all_gather_object(parts, mine[key].numpy(), group=self.group)       # (U x 5 doubles in all: small)
            r = np.concatenate(parts, axis=0)
            if (r[:, 0] == -2).any():
                raise NotImplementedError("more than 32 held-out items per user: not supported by the sharded evaluator")
            r = r[r[:, 0] >= 0]
            hr, p, rr, auc, ndcg = r.mean(axis=0).tolist()
            out.update({"hr" + suf: hr, "p" + suf: p, "r" + suf: rr, "auc" + suf: auc, "ndcg" + suf: ndcg})
        return out

    def local_state(self):
        return {"Gu": self.engine.t["Gu"].detach().clone(), "Gi": self.m.Gi_shard.detach().clone(), "Bi": self.m.Bi_col.detach().clone()}

    def load_local_state(self, sd):
        self.engine.t["Gu"].copy_(sd["Gu"])
        self.m.Gi_shard.copy_(sd["Gi"])
        self.m.Bi_col.copy_(sd["Bi"])
        self.engine.tables_dirty()

    def full_state(self, sd=None):
        sd = self.local_state() if sd is None else sd
        out = {}
        for n, total in (("Gu", self.num_users), ("Gi", self.num_items), ("Bi", self.num_items)):
            loc = sd[n].cpu()
            sh = self.ush if n == "Gu" else self.ish
            pad = torch.zeros((sh,) + tuple(loc.shape[1:]), dtype=loc.dtype)
            pad[:min(sh, loc.shape[0])] = loc[:sh]
            parts = [torch.empty_like(pad) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(pad, parts, dst=0, group=self.group)
            if self.rank == 0:
                out[n] = torch.cat(parts, dim=0)[:total]
        if self.rank == 0:
            out["Bi"] = out["Bi"].reshape(-1)
        return out if self.rank == 0 else None

    def store_recommendation(self, path):
        """Evaluator.py:225-239: every rank lists the top-k of ITS users (device top-K; rows that hinge on ties redone with numpy
        exactly as the single-GPU evaluator does), rank 0 writes the ranks' lines in user order."""
        self._gather_items()
        K, nu, lines = self.params.top_k, self.u1 - self.u0, []
        for b0 in range(0, max(nu, 0), 4096):
            b1 = min(nu, b0 + 4096)
            sc = self.ev_eng.score_block(b0, b1)
            idx, val, flag = self.ev_eng.topk(b0, b1, sc, self._csr["train"], K)
            idx, val, flag = idx.cpu().numpy(), val.cpu().numpy(), flag.cpu().numpy()
            for r in range(b1 - b0):
                u = self.u0 + b0 + r
                if flag[r]:
                    row = sc[r].cpu().numpy()
                    ids = row.argsort()[-K:][::-1]
                    vals = row[ids]
                else:
                    kk = min(K, idx.shape[1], sc.shape[1])
                    ids, vals = idx[r, :kk], val[r, :kk]
                lines += [str(u) + '\t' + str(v) + '\t' + str(vals[q]) + '\n' for q, v in enumerate(ids)]
        parts = [None] * self.world
        dist.all_gather_object(parts, "".join(lines), group=self.group)
        if self.rank == 0:
            with open(path, 'w') as f:
                f.write("".join(parts))

    def train(self):
        params, dev = self.params, self.engine.device
        max_metrics = {'hr': 0, 'p': 0, 'r': 0, 'auc': 0, 'ndcg': 0}
        best_state, best_epoch, best_epoch_print = None, getattr(params, "restore_epochs", 1), 'No best epoch found!'
        results = {}
        rec = getattr(params, "rec", "bprmf")
        wdir = os.path.join(configs.weight_dir(), params.dataset, rec)
        rdir = os.path.join(configs.results_dir(), params.dataset, rec)
        if self.rank == 0:
            os.makedirs(wdir, exist_ok=True)
            os.makedirs(rdir, exist_ok=True)
            print('Start training...')
        empty = torch.zeros(0, dtype=torch.int32, device=dev)
        loss_buf = torch.zeros(self.steps_per_epoch, dtype=torch.float32, device=dev)
        verbose, best_metric = getattr(params, "verbose", -1), getattr(params, "best_metric", "ndcg")
        for it in range(1, params.epochs + 1):
            start = time()
            loss_buf.zero_()
            for s in range(self.steps_per_epoch):
                u, i, j = self.sampler.sample(self.batch) if self.sampler is not None else (empty, empty, empty)
                self.m.step(u, i, j, loss_out=loss_buf, loss_index=s)
            loss = float(loss_buf.double().sum().item())
            epoch_text = 'Epoch {0}/{1} \tLoss (rank 0 shard): {2:.3f}'.format(it, params.epochs, loss / self.steps_per_epoch)
            epoch_print = self.evaluator.eval(it, results, epoch_text, start)
            for metric in max_metrics.keys():
                if max_metrics[metric] <= results[it][metric + '_v']:
                    max_metrics[metric] = results[it][metric + '_v']
                    if metric == best_metric:
                        best_epoch, best_state, best_epoch_print = it, self.local_state(), epoch_print
            if (it % verbose == 0 or it == 1) and verbose != -1:
                full = self.full_state()
                if self.rank == 0:
                    torch.save(full, os.path.join(wdir, f'weights-{it}-{self.directory_parameters}.pt'))
        self.engine.sync_check()
        if self.m.x.overflowed():
            raise RuntimeError("a row-routing bucket overflowed (dist.UserShardedBPRMF: raise `slack`)")
        last = params.epochs
        if self.rank == 0:
            print('Training end...')
        self.store_recommendation(os.path.join(rdir, f'recs-{last}-{self.directory_parameters}.tsv'))
        if self.rank == 0:
            with open(os.path.join(rdir, f'results-metrics-{self.directory_parameters}') + '.pkl', 'wb') as f:
                pickle.dump(results, f)
            print("Store Best Model at Epoch {0}".format(best_epoch))
            print(best_epoch_print)
        last_state = self.local_state()
        if best_state is not None:
            full = self.full_state(best_state)
            if self.rank == 0:
                torch.save(full, os.path.join(wdir, f'best-weights-{best_epoch}-{self.directory_parameters}.pt'))
            self.load_local_state(best_state)
        self.store_recommendation(os.path.join(rdir, f'best-recs-{best_epoch}-{self.directory_parameters}.tsv'))
        self.load_local_state(last_state)
        if self.rank == 0:
            print('End Store Best Model!')
            print('Best Values for Each Metric:\nHR\tPrec\tRec\tAUC\tnDCG\n{}\t{}\t{}\t{}\t{}\n'.format(
                max_metrics['hr'], max_metrics['p'], max_metrics['r'], max_metrics['auc'], max_metrics['ndcg']))
        self.results = results
        return results


from .evaluator import Evaluator as _Evaluator                                              # noqa: E402


class _ShardedEvaluator(_Evaluator):
    """The reference's eval() surface (epoch line, results dict with its key aliasing) over the shard-additive device metrics;
    only rank 0 prints."""

    def metrics(self):
        return self.model.metrics(self.k)

    def eval(self, epoch=0, results=None, epoch_text='', start_time=0):
        quiet = contextlib.redirect_stdout(io.StringIO()) if self.model.rank != 0 else contextlib.nullcontext()
        with quiet:
            return super().eval(epoch, results, epoch_text, start_time)
