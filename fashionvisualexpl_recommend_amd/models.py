"""BPRMF and VBPR: host-side mirror of the reference's model classes
(src/recommender/RecommenderModel.py:16-25, models/BPRMF.py:21-192, models/VBPR.py:19-144).

Same constructor `Model(data, params)`, same methods (`call`, `predict_all`, `train_step`, `train`) and the same
attribute names (Bi, Gu, Gi, Tu, F, E, Bp, evaluator, directory_parameters).  The tables are torch-ROCm tensors;
every computation on them runs in libbprx.so through the C ABI (include/bprx.h) -- no autograd, no eager
op chain.  `params` may carry three extra knobs the reference does not have:
    optimizer  'adam_tf23' (default: the reference's tf.optimizers.Adam semantics) | 'sgd'
    dtype      'fp32' (default) | 'bf16'   storage/compute type of the frozen feature table F
    init_seed  seed of the Glorot-uniform initialiser (TF's own RNG stream is not reproducible without TF)
"""
import os
import pickle
from time import time

import numpy as np
import torch

from . import configs
from .engine import Engine, as_index
from .evaluator import Evaluator
from .synth import glorot_uniform


class _Scores:
    """What predict_all() returns: the reference returns a tf.Tensor whose only use is `.numpy()`
    (Evaluator.py:174,231)."""

    def __init__(self, t):
        self.tensor = t

    def numpy(self):
        return self.tensor.cpu().numpy()


class RecommenderModel:
    def __init__(self, data, params):                       # RecommenderModel.py:16-25
        self.data = data
        self.num_items = data.num_items
        self.num_users = data.num_users
        self.params = params
        self.epochs = params.epochs
        self.batch_size = params.batch_size
        self.verbose = getattr(params, "verbose", -1)
        self.restore_epochs = getattr(params, "restore_epochs", 1)
        self.model_name = getattr(params, "rec", None)
        self.dataset_name = getattr(params, "dataset", None)


class BPRMF(RecommenderModel):
    model_kind = "bprmf"

    def __init__(self, data, params, init=None):
        """`init`: optional dict of initial tables (numpy/torch) overriding the seeded Glorot initialiser."""
        super().__init__(data, params)
        self.embed_k = params.embed_k
        self.learning_rate = params.lr
        self.reg = params.reg
        self.optimizer_name = getattr(params, "optimizer", "adam_tf23")
        self.evaluator = Evaluator(self, data, params.top_k)                       # BPRMF.py:40
        self.directory_parameters = f'batch_{params.batch_size}-K_{params.embed_k}-lr_{params.lr}-reg_{params.reg}'
        self._build(init or {})

    # ---- parameters (BPRMF.py:48-52) ---------------------------------------------------------------------------
    def _init_tables(self, init):
        rs = np.random.RandomState(getattr(self.params, "init_seed", 0))
        t = {"Bi": np.zeros(self.num_items, np.float32),
             "Gu": glorot_uniform(rs, self.num_users, self.embed_k),
             "Gi": glorot_uniform(rs, self.num_items, self.embed_k)}
        t.update({k: v for k, v in init.items() if k in t})
        return t, rs

    def _engine_kwargs(self):
        return dict(model="bprmf", num_users=self.num_users, num_items=self.num_items, embed_k=self.embed_k)

    def _adam_form(self):
        """adam_tf23 lazily-exact or by whole-table sweeps (identical arithmetic; include/bprx.h BPRX_FLAG_ADAM_*): a row's replay
        is a serial recurrence over the steps since its last touch -- one epoch = interactions / batch steps in the reference's
        visiting order (~0.45 us each, measured: bprx_api.hip) -- against a sweep that moves every row's (p, m, v) once per step.  The library estimates
        this from max_batch; here the batch size and the interaction count are known."""
        if self.optimizer_name != "adam_tf23" or os.environ.get("BPRX_ADAM_LAZY") is not None:
            return None
        kw = self._engine_kwargs()
        n_pos = sum(len(pos) for pos in self.data.training_list)
        chain_us = 0.45 * n_pos / max(1, self.batch_size)
        elems = kw["num_users"] * (kw["embed_k"] + kw.get("embed_d", 0)) + kw["num_items"] * (kw["embed_k"] + 1)
        return "lazy" if chain_us < elems * 24.0 / 4e6 else "sweep"

    def _build(self, init):
        t, _ = self._init_tables(init)
        self.engine = Engine(optimizer=self.optimizer_name, lr=self.learning_rate, reg=self.reg,
                             max_batch=max(self.batch_size, 4096), adam_form=self._adam_form(), **self._engine_kwargs())
        self.engine.bind(**t)
        self._alias()

    def _alias(self):
        """The reference's attribute surface (model.Gu, .Gi, .Bi, .Tu, .F, .E, .Bp) is served by the class properties below:
        they go through engine.t, which first brings every row up to date when adam_tf23 runs lazily (bprx_sync_adam) -- a
        raw alias of the tensor would show rows that have not been replayed yet."""

    # ---- BPRMF.py:55-76 ------------------------------------------------------------------------------------------
    def call(self, inputs, training=None, mask=None):
        user, item = inputs
        u, i = as_index(user, self.engine.device).long(), as_index(item, self.engine.device).long()
        xui = self.engine.score_pairs(u, i)
        return xui, self.Bi[i], self.Gu[u], self.Gi[i]

    __call__ = call

    # ---- BPRMF.py:78-85 --------------------------------------------------------------------------------------------
    def predict_block(self, u0, u1):
        return self.engine.score_block(u0, u1).cpu().numpy()

    def predict_all(self):
        return _Scores(self.engine.score_block(0, self.num_users))

    # ---- BPRMF.py:87-125 -------------------------------------------------------------------------------------------
    def train_step(self, batch):
        user, pos, neg = (as_index(b, self.engine.device) for b in batch)
        return float(self.engine.step(user, pos, neg).item())      # loss.numpy(): one host sync, like the reference

    # ---- state snapshots (the reference deep-copies the whole model, BPRMF.py:156) --------------------------------
    def state_dict(self):
        sd = {n: v.detach().clone() for n, v in self.engine.t.items() if n != "F"}
        sd["adam_step"] = self.engine.adam_step
        return sd

    def load_state_dict(self, sd):
        for n, v in sd.items():
            if n == "adam_step":
                self.engine.adam_step = v
            else:
                self.engine.t[n].copy_(v)
        self.engine.tables_dirty()                      # the handle caches images derived from E/Bp

    def weights_path(self, epoch):
        rec = getattr(self.params, "rec", self.model_kind)
        return os.path.join(configs.weight_dir(), self.params.dataset, rec, f'weights-{epoch}-{self.directory_parameters}.pt')

    # ---- BPRMF.py:127-192 ------------------------------------------------------------------------------------------
    def train(self, resume=False):
        """The reference's training loop.  `resume=True` (the reference parses --restore_epochs but
        never restores, BPRMF.py:130): continue from the snapshot `weights-{restore_epochs}-...pt` -- tables, Adam slots
        and step counter are reloaded and the deterministic triplet stream is fast-forwarded by restore_epochs epochs,
        so the remaining epochs see exactly the batches an uninterrupted run would."""
        max_metrics = {'hr': 0, 'p': 0, 'r': 0, 'auc': 0, 'ndcg': 0}
        best_state = None
        best_epoch = self.restore_epochs
        best_epoch_print = 'No best epoch found!'
        results = {}
        steps_total = (sum(len(pos) for pos in self.data.training_list) // self.params.batch_size) * self.params.epochs
        if getattr(self.params, "sampler", "ref_stream") == "philox":
            # --sampler philox: the device epoch walk (the reference's visiting order as a stateless Philox stream; bit-exact
            # CPU twin in the oracle) instead of the host MT19937 stream -- no index upload per step
            from .engine import EpochWalkSampler
            smp = EpochWalkSampler(self.data.training_list, self.num_items, device=self.engine.device,
                                   seed=getattr(self.params, "init_seed", 0)).feeds(self.engine)
            next_batch = (smp.sample(self.params.batch_size) for _ in range(steps_total))
        else:
            next_batch = self.data.next_triple_batch(self.engine.device)
        steps = 0
        loss = 0
        it = 1
        steps_per_epoch = sum([len(pos) for pos in self.data.training_list]) // self.params.batch_size
        rec = getattr(self.params, "rec", self.model_kind)
        wdir = os.path.join(configs.weight_dir(), self.params.dataset, rec)
        rdir = os.path.join(configs.results_dir(), self.params.dataset, rec)
        os.makedirs(wdir, exist_ok=True)
        os.makedirs(rdir, exist_ok=True)
        if resume:
            path = self.weights_path(self.restore_epochs)
            self.load_state_dict(torch.load(path, map_location=self.engine.device, weights_only=True))
            for _ in range(self.restore_epochs * steps_per_epoch):
                next(next_batch)
            it = self.restore_epochs + 1
            print('Restored epoch {0} from {1}'.format(self.restore_epochs, path))
        start_ep = time()
        print('Start training...')
        # (the reference reads loss.numpy() after every step, BPRMF.py:125 -- a host synchronisation per step; here the
        #  step losses land in a device buffer and are read once per epoch, so the host runs ahead of the device)
        loss_buf = torch.zeros(max(1, steps_per_epoch), dtype=torch.float32, device=self.engine.device)
        for batch in next_batch:
            steps += 1
            user, pos, neg = (as_index(b, self.engine.device) for b in batch)
            self.engine.step(user, pos, neg, loss_out=loss_buf, loss_index=steps - 1)
            if steps == steps_per_epoch:                                        # epoch is over
                loss = float(loss_buf[:steps].double().sum().item())
                epoch_text = 'Epoch {0}/{1} \tLoss: {2:.3f}'.format(it, self.params.epochs, loss / steps)
                epoch_print = self.evaluator.eval(it, results, epoch_text, start_ep)
                for metric in max_metrics.keys():
                    if max_metrics[metric] <= results[it][metric + '_v']:
                        max_metrics[metric] = results[it][metric + '_v']
                        if metric == self.params.best_metric:
                            best_epoch, best_state, best_epoch_print = it, self.state_dict(), epoch_print
                if (it % self.verbose == 0 or it == 1) and self.verbose != -1:
                    torch.save(self.state_dict(), os.path.join(wdir, f'weights-{it}-{self.directory_parameters}.pt'))
                start_ep = time()
                it += 1
                loss = 0
                steps = 0
        print('Training end...')
        self.evaluator.store_recommendation(path=os.path.join(rdir, f'recs-{it - 1}-{self.directory_parameters}.tsv'))
        with open(os.path.join(rdir, f'results-metrics-{self.directory_parameters}') + '.pkl', 'wb') as f:
            pickle.dump(results, f)                                             # utils/write.py:14-22
        print("Store Best Model at Epoch {0}".format(best_epoch))
        print(best_epoch_print)
        last_state = self.state_dict()
        if best_state is not None:
            torch.save(best_state, os.path.join(wdir, f'best-weights-{best_epoch}-{self.directory_parameters}.pt'))
            self.load_state_dict(best_state)
        self.evaluator.store_recommendation(
            path=os.path.join(rdir, f'best-recs-{best_epoch}-{self.directory_parameters}.tsv'))
        self.load_state_dict(last_state)
        print('End Store Best Model!')
        print('Best Values for Each Metric:\nHR\tPrec\tRec\tAUC\tnDCG\n{}\t{}\t{}\t{}\t{}\n'.format(
            max_metrics['hr'], max_metrics['p'], max_metrics['r'], max_metrics['auc'], max_metrics['ndcg']))
        self.results = results
        return results


def _table_property(name):
    return property(lambda self: self.engine.t[name], doc="bound tensor %s (current: lazy adam_tf23 rows are replayed first)" % name)


for _n in ("Gu", "Gi", "Bi", "Tu", "F", "E", "Bp"):
    setattr(BPRMF, _n, _table_property(_n))


class VBPR(BPRMF):
    model_kind = "vbpr"

    def __init__(self, data, params, init=None, features=None):
        """`features`: optional [I,D] array used instead of the cnn_features .npy (already max-abs normalised
        unless `normalize=True` is left to do it)."""
        self.embed_d = params.embed_d
        self._features = features
        super().__init__(data, params, init)
        self.directory_parameters = f'batch_{params.batch_size}-D_{params.embed_d}-K_{params.embed_k}' \
                                    f'-lr_{params.lr}-reg_{params.reg}'      # VBPR.py:35-39

    def process_cnn_visual_features(self):
        """visual_loader_mixin.py:22-31: np.load, divide by the GLOBAL max-abs, D = shape[1]."""
        f = self._features
        if f is None:
            f = np.load(configs.cnn_features_path(self.params.dataset, getattr(self.params, "cnn_model", "vgg19"),
                                                  getattr(self.params, "output_layer", "fc2")))
        f = np.asarray(f)
        self.cnn_features = f / np.max(np.abs(f))
        self.dim_cnn_features = self.cnn_features.shape[1]

    def _init_tables(self, init):
        t, rs = super()._init_tables(init)
        self.process_cnn_visual_features()                                      # VBPR.py:41
        D, d = self.dim_cnn_features, self.embed_d
        v = {"Bp": glorot_uniform(rs, D, 1).reshape(-1),                        # VBPR.py:44-54, same creation order
             "Tu": glorot_uniform(rs, self.num_users, d),
             "F": self.cnn_features.astype(np.float32),
             "E": glorot_uniform(rs, D, d)}
        v.update({k: val for k, val in init.items() if k in v})
        t.update(v)
        return t, rs

    def _engine_kwargs(self):
        return dict(model="vbpr", num_users=self.num_users, num_items=self.num_items, embed_k=self.embed_k,
                    embed_d=self.embed_d, feat_dim=self.dim_cnn_features,
                    feat_dtype=getattr(self.params, "dtype", "fp32"))

    def call(self, inputs, training=None, mask=None):                           # VBPR.py:59-86
        user, item = inputs
        u, i = as_index(user, self.engine.device).long(), as_index(item, self.engine.device).long()
        xui = self.engine.score_pairs(u, i)
        return xui, self.Gu[u], self.Gi[i], self.F[i], self.Tu[u], self.Bi[i]

    __call__ = call
