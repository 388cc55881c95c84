"""Engine: one libbprx handle + the torch-ROCm tensors it is bound to.

PyTorch is plumbing here (device memory, streams); every computation of the hot path runs in
libbprx.so through the C ABI (include/bprx.h).  A missing library or a missing GPU raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _ffi

PARAM_NAMES = ("Gu", "Gi", "Bi", "Tu", "E", "Bp")


def _ptr(t):
    return None if (t is None or t.numel() == 0) else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def as_index(x, device):
    """int64/any index array -> contiguous int32 device tensor (reference batches are int64, dataset.py:105-107)."""
    if isinstance(x, torch.Tensor):
        t = x.reshape(-1)
        if t.dtype != torch.int32:
            t = t.to(torch.int32)
        return t.to(device, non_blocking=True).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x).reshape(-1), dtype=np.int32), device=device)


class Engine:
    def __init__(self, model, num_users, num_items, embed_k, embed_d=0, feat_dim=0, feat_dtype="fp32",
                 optimizer="adam_tf23", lr=1e-3, reg=0.0, max_batch=256, device=None,
                 beta1=0.9, beta2=0.999, epsilon=1e-7, export_user_grad=False, export_item_grad=False, feat_scale=448.0,
                 dense_allreduce=False, adam_form=None):
        if not torch.cuda.is_available():
            raise RuntimeError("fashionvisualexpl_recommend_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
        self.lib = _ffi.lib()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.model, self.optimizer = model, optimizer
        self.U, self.I, self.k = int(num_users), int(num_items), int(embed_k)
        self.d, self.D = (int(embed_d), int(feat_dim)) if model == "vbpr" else (0, 0)
        self.feat_dtype = feat_dtype
        self.feat_scale = float(feat_scale)
        self.max_batch = int(max_batch)
        cfg = _ffi.Config(_ffi.ABI_VERSION, _ffi.MODEL[model], self.U, self.I, self.k, self.d, self.D,
                          _ffi.FEAT_DTYPE[feat_dtype], _ffi.OPTIMIZER[optimizer], self.device.index, self.max_batch,
                          lr, reg, beta1, beta2, epsilon,
                          (_ffi.FLAG_EXPORT_USER_GRAD if export_user_grad else 0) |
                          (_ffi.FLAG_EXPORT_ITEM_GRAD if export_item_grad else 0) |
                          (_ffi.FLAG_DENSE_ALLREDUCE if dense_allreduce else 0) |
                          {None: 0, "sweep": _ffi.FLAG_ADAM_SWEEP, "lazy": _ffi.FLAG_ADAM_LAZY}[adam_form], self.feat_scale)
        h = C.c_void_p()
        _ffi.check(None, self.lib.bprx_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self._t = {}
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)

    @property
    def t(self):
        """The bound tensors.  adam_tf23 is lazy-exact inside the library: reading the tensors from outside first brings
        every row up to date (bprx_sync_adam; a host call that returns at once when nothing is pending)."""
        if self.optimizer == "adam_tf23" and getattr(self, "h", None) and self._t:
            self.sync_adam()
        return self._t

    @t.setter
    def t(self, v):
        self._t = v

    def adam_is_lazy(self):
        return bool(self.lib.bprx_adam_is_lazy(self.h))

    def sync_adam(self):
        _ffi.check(self.h, self.lib.bprx_sync_adam(self.h, _stream()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.bprx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state -------------------------------------------------------------------------------------------------
    def bind(self, Gu, Gi, Bi, Tu=None, F=None, E=None, Bp=None, slots=None):
        """Bind caller-owned device tensors (fp32, contiguous; F fp32 or bf16).  Adam slots are created here
        (zeros, like tf.optimizers.Adam's m/v) unless given."""
        def prep(x, shape, dtype=torch.float32):
            if x is None:
                return None
            x = torch.as_tensor(x)
            x = x.to(device=self.device, dtype=dtype).reshape(shape).contiguous()
            return x
        t = {"Gu": prep(Gu, (self.U, self.k)), "Gi": prep(Gi, (self.I, self.k)), "Bi": prep(Bi, (self.I,))}
        if self.model == "vbpr":
            if self.feat_dtype == "fp8":
                # e4m3fn codes of f * feat_scale (a float8 tensor is taken as is; anything else is quantised here,
                # round-to-nearest-even, on the device)
                Ft = torch.as_tensor(F)
                if Ft.dtype != torch.float8_e4m3fn:
                    Ft = (Ft.to(device=self.device, dtype=torch.float32) * self.feat_scale).to(torch.float8_e4m3fn)
                Fp = Ft.to(self.device).reshape(self.I, self.D).contiguous()
            else:
                Fp = prep(F, (self.I, self.D), torch.bfloat16 if self.feat_dtype == "bf16" else torch.float32)
            t.update(Tu=prep(Tu, (self.U, self.d)), F=Fp, E=prep(E, (self.D, self.d)), Bp=prep(Bp, (self.D,)))
        if self.optimizer == "adam_tf23":
            for n in PARAM_NAMES:
                if t.get(n) is None:
                    continue
                for s in ("m_", "v_"):
                    given = None if slots is None else slots.get(s + n)
                    t[s + n] = torch.zeros_like(t[n]) if given is None else prep(given, tuple(t[n].shape))
        tb = _ffi.Tables()
        for n in _ffi.TABLE_FIELDS:
            setattr(tb, n, None if t.get(n) is None else t[n].data_ptr())
        torch.cuda.current_stream(self.device).synchronize()   # bind tiles F on the null stream: the tables must be complete
        _ffi.check(self.h, self.lib.bprx_bind_tables(self.h, C.byref(tb)))
        self.t = t
        return self

    def tables_dirty(self):
        """Call after writing any bound table from outside the library (bprx_tables_dirty): the handle reuses images
        derived from E/Bp (their bf16/fp8 copy, the item projections) until a step changes them."""
        _ffi.check(self.h, self.lib.bprx_tables_dirty(self.h, _stream()))

    def params(self):
        return {n: self.t[n] for n in PARAM_NAMES if self.t.get(n) is not None}

    def set_hyper(self, lr, reg):
        _ffi.check(self.h, self.lib.bprx_set_hyper(self.h, lr, reg))

    @property
    def adam_step(self):
        return int(self.lib.bprx_get_adam_step(self.h))

    @adam_step.setter
    def adam_step(self, v):
        _ffi.check(self.h, self.lib.bprx_set_adam_step(self.h, int(v), _stream()))

    # ---- hot path ------------------------------------------------------------------------------------------------
    def score_pairs(self, user, item):
        u, i = as_index(user, self.device), as_index(item, self.device)
        x = torch.empty(u.numel(), dtype=torch.float32, device=self.device)
        for s in range(0, u.numel(), self.max_batch):            # the handle's pair scratch holds max_batch rows
            n = min(self.max_batch, u.numel() - s)
            _ffi.check(self.h, self.lib.bprx_score_pairs(self.h, _ptr(u[s:s + n]), _ptr(i[s:s + n]), n,
                                                         _ptr(x[s:s + n]), _stream()))
        return x

    def step(self, user, pos, neg, want_loss=True, loss_out=None, loss_index=0):
        """One train step on device int32 index tensors.  Returns the device loss scalar (no host sync).
        loss_out / loss_index: write the step's loss to element `loss_index` of this fp32 device tensor instead (a training
        loop that reads its losses once per epoch never waits for a step)."""
        if loss_out is not None:
            lp = C.c_void_p(loss_out.data_ptr() + 4 * int(loss_index))
        else:
            lp = _ptr(self._loss) if want_loss else None
        _ffi.check(self.h, self.lib.bprx_step(self.h, _ptr(user), _ptr(pos), _ptr(neg), user.numel(), lp, _stream()))
        return self._loss if loss_out is None else loss_out

    def step_lr(self):
        """The bias-corrected learning rate of the step begun last (bprx_step_lr; sgd: lr)."""
        v = C.c_float()
        _ffi.check(self.h, self.lib.bprx_step_lr(self.h, C.byref(v)))
        return v.value

    def step_begin(self, user, pos, neg):
        _ffi.check(self.h, self.lib.bprx_step_begin(self.h, _ptr(user), _ptr(pos), _ptr(neg), user.numel(), _stream()))

    def step_begin_sparse(self, user, pos, neg):
        """First half of step_begin (the user-side gradients are final afterwards); the index tensors must stay alive and
        unchanged until step_begin_dense has been called."""
        self._pend_idx = (user, pos, neg)
        _ffi.check(self.h, self.lib.bprx_step_begin_sparse(self.h, _ptr(user), _ptr(pos), _ptr(neg), user.numel(), _stream()))

    def step_begin_dense(self):
        _ffi.check(self.h, self.lib.bprx_step_begin_dense(self.h, _stream()))
        self._pend_idx = None

    def dense_grad(self):
        """fp32 view of the handle-owned dense gradient buffer [D*d + D] (dE then dBp) for the RCCL all-reduce."""
        p, n = C.c_void_p(), C.c_int64()
        _ffi.check(self.h, self.lib.bprx_dense_grad(self.h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return None
        if getattr(self, "_dense_view", None) is None:
            self._dense_view = _DevView(p.value, n.value, self.device).tensor
        return self._dense_view

    def step_project(self):
        _ffi.check(self.h, self.lib.bprx_step_project(self.h, _stream()))

    def user_grad(self):
        """Zero-copy views [U,k], [U,d] of the staging tables that hold the exported user-row gradients."""
        if getattr(self, "_ugrad", None) is None:
            a, b = C.c_void_p(), C.c_void_p()
            _ffi.check(self.h, self.lib.bprx_user_grad(self.h, C.byref(a), C.byref(b)))
            g = _DevView(a.value, self.U * self.k, self.device).tensor.view(self.U, self.k)
            t = _DevView(b.value, self.U * self.d, self.device).tensor.view(self.U, self.d) if self.d else None
            self._ugrad = (g, t)
        return self._ugrad

    # ---- replicated-user multi-GPU step (include/bprx.h: bprx_pack_user_msg / bprx_apply_user_msgs) -------------
    def user_msg_floats(self, cap):
        return int(self.lib.bprx_user_msg_floats(self.h, int(cap)))

    def pack_user_msg(self, user, cap, msg):
        _ffi.check(self.h, self.lib.bprx_pack_user_msg(self.h, _ptr(user), user.numel(), int(cap), _ptr(msg), _stream()))

    def apply_user_msgs(self, msgs, nranks, cap, scale):
        _ffi.check(self.h, self.lib.bprx_apply_user_msgs(self.h, _ptr(msgs), int(nranks), int(cap), float(scale), _stream()))

    def sum_dense_parts(self, parts, nranks):
        _ffi.check(self.h, self.lib.bprx_sum_dense_parts(self.h, _ptr(parts), int(nranks), _stream()))

    def item_grad(self):
        """Zero-copy views [I,k], [I] of the staging tables that hold the exported item-row gradients."""
        if getattr(self, "_igrad", None) is None:
            a, b = C.c_void_p(), C.c_void_p()
            _ffi.check(self.h, self.lib.bprx_item_grad(self.h, C.byref(a), C.byref(b)))
            self._igrad = (_DevView(a.value, self.I * self.k, self.device).tensor.view(self.I, self.k),
                           _DevView(b.value, self.I, self.device).tensor)
        return self._igrad

    def clear_item_grad(self, n_rows):
        _ffi.check(self.h, self.lib.bprx_clear_item_grad(self.h, int(n_rows), 0, _stream()))

    def clear_user_grad(self, n_rows):
        _ffi.check(self.h, self.lib.bprx_clear_user_grad(self.h, int(n_rows), 0, _stream()))

    def clear_item_marks(self, n_rows):
        """After bprx_route_pack (which returns the exported gradient rows to zero): only the touched-row marks are left."""
        _ffi.check(self.h, self.lib.bprx_clear_item_grad(self.h, int(n_rows), 1, _stream()))

    def clear_user_marks(self, n_rows):
        _ffi.check(self.h, self.lib.bprx_clear_user_grad(self.h, int(n_rows), 1, _stream()))

    def step_end(self, want_loss=True, loss_out=None, loss_index=0):
        """loss_out / loss_index: as in step() -- the loss lands in element `loss_index` of a device tensor."""
        if loss_out is not None:
            lp = C.c_void_p(loss_out.data_ptr() + 4 * int(loss_index))
        else:
            lp = _ptr(self._loss) if want_loss else None
        _ffi.check(self.h, self.lib.bprx_step_end(self.h, lp, _stream()))
        return self._loss if loss_out is None else loss_out

    def score_block(self, u0, u1, out=None):
        if out is None:
            out = torch.empty((u1 - u0, self.I), dtype=torch.float32, device=self.device)
        _ffi.check(self.h, self.lib.bprx_score_block(self.h, u0, u1, _ptr(out), _stream()))
        return out

    def eval_users(self, u0, u1, scores, train_csr, eval_csr, K):
        """bprx_eval_users: per-user (hr, prec, rec, auc, ndcg) as a float64 device tensor [(u1-u0), 5]."""
        out = torch.empty((u1 - u0, 5), dtype=torch.float64, device=self.device)
        _ffi.check(self.h, self.lib.bprx_eval_users(self.h, u0, u1, _ptr(scores), _ptr(train_csr[0]), _ptr(train_csr[1]),
                                                    _ptr(eval_csr[0]), _ptr(eval_csr[1]), int(K), _ptr(out), _stream()))
        return out

    # ---- item-sharded evaluation: counts that are additive over item shards (include/bprx.h) ----------------------
    def eval_pos(self, u0, u1, scores, item_lo, items_total, eval_csr):
        sp = torch.empty((u1 - u0, 32), dtype=torch.float32, device=self.device)
        _ffi.check(self.h, self.lib.bprx_eval_pos(self.h, u0, u1, _ptr(scores), int(item_lo), int(items_total),
                                                  _ptr(eval_csr[0]), _ptr(eval_csr[1]), _ptr(sp), _stream()))
        return sp

    def eval_counts(self, u0, u1, scores, item_lo, items_total, train_csr, eval_csr, sp):
        counts = torch.empty((u1 - u0, 65), dtype=torch.int32, device=self.device)
        _ffi.check(self.h, self.lib.bprx_eval_counts(self.h, u0, u1, _ptr(scores), int(item_lo), int(items_total),
                                                     _ptr(train_csr[0]), _ptr(train_csr[1]), _ptr(eval_csr[0]),
                                                     _ptr(eval_csr[1]), _ptr(sp), _ptr(counts), _stream()))
        return counts

    def eval_finish(self, u0, u1, items_total, eval_csr, sp, counts, K):
        out = torch.empty((u1 - u0, 5), dtype=torch.float64, device=self.device)
        _ffi.check(self.h, self.lib.bprx_eval_finish(self.h, u0, u1, int(items_total), _ptr(eval_csr[0]), _ptr(sp), _ptr(counts),
                                                     int(K), _ptr(out), _stream()))
        return out

    def topk(self, u0, u1, scores, train_csr, K):
        """bprx_topk: masks the train items IN `scores` and returns (idx int32 [n,K], val fp32 [n,K], flag int32 [n])."""
        n = u1 - u0
        idx = torch.empty((n, K), dtype=torch.int32, device=self.device)
        val = torch.empty((n, K), dtype=torch.float32, device=self.device)
        flag = torch.empty(n, dtype=torch.int32, device=self.device)
        _ffi.check(self.h, self.lib.bprx_topk(self.h, u0, u1, _ptr(scores), _ptr(train_csr[0]), _ptr(train_csr[1]), int(K),
                                              _ptr(idx), _ptr(val), _ptr(flag), _stream()))
        return idx, val, flag

    def profile(self, on):
        _ffi.check(self.h, self.lib.bprx_profile_enable(self.h, 1 if on else 0))

    def profile_read(self):
        """{phase: (total_ms, launches)} accumulated since the last read (HIP events on the launch stream)."""
        ms = np.zeros(len(_ffi.PHASES), np.float64)
        n = np.zeros(len(_ffi.PHASES), np.int64)
        _ffi.check(self.h, self.lib.bprx_profile_read(self.h, ms.ctypes.data, n.ctypes.data))
        return {p: (float(ms[i]), int(n[i])) for i, p in enumerate(_ffi.PHASES) if n[i]}

    def sync_check(self):
        _ffi.check(self.h, self.lib.bprx_sync_check(self.h, _stream()))


def scatter_add(table, idx, rows, scale):
    """table[idx] += scale * rows on the device (bprx_scatter_add); duplicates in idx are summed."""
    lib = _ffi.lib()
    assert table.is_contiguous() and rows.is_contiguous() and idx.dtype == torch.int32 and table.dtype == torch.float32
    ncols = table.shape[1] if table.dim() == 2 else 1
    rc = lib.bprx_scatter_add(_ptr(table), table.shape[0], ncols, _ptr(idx), _ptr(rows), idx.numel(), float(scale), _stream())
    if rc < 0:
        raise _ffi.BprxError(rc, "bprx_scatter_add failed")


def adam_rows(p, m, v, g, lr_t, beta1=0.9, beta2=0.999, eps=1e-7):
    """One adam_tf23 step of a whole row shard from its summed gradient table g (returned to zero): bprx_adam_rows."""
    lib = _ffi.lib()
    assert all(t.is_contiguous() and t.dtype == torch.float32 and t.numel() == p.numel() for t in (p, m, v, g))
    rc = lib.bprx_adam_rows(_ptr(p), _ptr(m), _ptr(v), _ptr(g), p.numel(), float(lr_t), float(beta1), float(beta2), float(eps),
                            _stream())
    if rc < 0:
        raise _ffi.BprxError(rc, "bprx_adam_rows failed")


class _DevView:
    """Zero-copy torch view of library-owned device memory via __cuda_array_interface__."""

    def __init__(self, ptr, n, device):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}
        self.tensor = torch.as_tensor(self, device=device)


class PhiloxSampler:
    """bprx_sample_philox: device-side stateless throughput sampler over a CSR of training interactions."""

    def __init__(self, train_lists, num_items, device=None, seed=0):
        self.lib = _ffi.lib()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        indptr = np.zeros(len(train_lists) + 1, dtype=np.int64)
        for u, l in enumerate(train_lists):
            indptr[u + 1] = indptr[u] + len(l)
        items = np.fromiter((i for l in train_lists for i in sorted(l)), dtype=np.int32, count=int(indptr[-1]))
        pos_user = np.repeat(np.arange(len(train_lists), dtype=np.int32), np.diff(indptr))
        self.num_pos, self.num_items, self.seed, self.next = int(indptr[-1]), int(num_items), int(seed), 0
        if self.num_pos == 0:
            raise ValueError("no training interactions")
        self.indptr, self.items, self.pos_user = (torch.as_tensor(a, device=self.device) for a in (indptr, items, pos_user))

    @classmethod
    def from_csr(cls, indptr, items_sorted, pos_user, num_items, seed=0):
        """Device tensors: indptr int64 [U+1], items_sorted int32 [N] (ascending inside each user), pos_user int32 [N]."""
        self = cls.__new__(cls)
        self.lib = _ffi.lib()
        self.device = indptr.device
        self.indptr, self.items, self.pos_user = indptr.contiguous(), items_sorted.contiguous(), pos_user.contiguous()
        self.num_pos, self.num_items, self.seed, self.next = int(items_sorted.numel()), int(num_items), int(seed), 0
        return self

    def feeds(self, engine):
        """Name the engine whose steps consume this sampler's batches (bprx_sample_*_h): the sampler then also leaves the byte
        planes of the item ids that engine's index pass scans (segment mode, <= 65 536 items).  Returns self."""
        self._handle = engine.h if engine is not None else None
        return self

    def sample(self, B, first=None, out=None):
        """B triplets starting at stream position `first` (default: continue).  Returns int32 device tensors."""
        if first is None:
            first, self.next = self.next, self.next + B
        u, i, j = out if out is not None else tuple(torch.empty(B, dtype=torch.int32, device=self.device) for _ in range(3))
        rc = self.lib.bprx_sample_philox_h(getattr(self, "_handle", None), _ptr(self.indptr), _ptr(self.items), _ptr(self.pos_user),
                                           self.num_pos, self.num_items, self.seed, first, B, _ptr(u), _ptr(i), _ptr(j), 0, B,
                                           _stream())
        if rc < 0:
            raise _ffi.BprxError(rc, "bprx_sample_philox failed")
        return u, i, j


class EpochWalkSampler(PhiloxSampler):
    """bprx_sample_epoch: the reference's visiting order as a device stream -- per epoch a fresh permutation of the users
    (a keyed Feistel permutation evaluated on the device, bprx_epoch_prepare), every positive of every user exactly once,
    consecutively; negatives by Philox rejection.  Batches are user-grouped like the reference's."""

    def _prepare(self, epoch):
        """Everything epoch `epoch` needs, ENQUEUED without a host synchronisation or an upload: the user order is a keyed Feistel
        permutation evaluated pointwise on the device (bprx_epoch_prepare: slot -> user and the length of its list; CPU twin: the
        oracle's orc_epoch_perm), the prefix sums of the lengths are one device scan, the position -> slot map one more launch
        (bprx_epoch_slots).  Prepared ONE EPOCH AHEAD, so that an epoch switch inside a training loop is a pointer swap.
        (History: until round 3 the permutation was drawn on the host and copied synchronously at every epoch start -- bench.py's
        20-step timed regions cannot hide a host stall; an asynchronous pinned-memory upload behind a deep launch queue was worse:
        intermittent 25-80 ms stalls; then the stable argsort of per-user Philox keys on the device: exact, but a merge sort of 8
        launches inside a preparation of 23 launches and 158 us per epoch = 5 us per C2 step.)"""
        U = self.indptr.numel() - 1
        perm_d = torch.empty(U, dtype=torch.int32, device=self.device)
        lens = torch.empty(U, dtype=torch.int64, device=self.device)
        rc = self.lib.bprx_epoch_prepare(self.seed, epoch, U, _ptr(self.indptr), _ptr(perm_d), _ptr(lens), _stream())
        if rc < 0:
            raise _ffi.BprxError(rc, "bprx_epoch_prepare failed")
        epoch_ptr = torch.zeros(U + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(lens, 0, out=epoch_ptr[1:])
        # position -> slot of its user in the epoch order, once per epoch (4 B per interaction): saves the per-triplet
        # binary search over epoch_ptr in the kernel
        pos_slot = torch.empty(self.num_pos, dtype=torch.int32, device=self.device)
        rc = self.lib.bprx_epoch_slots(_ptr(epoch_ptr), U, _ptr(pos_slot), self.num_pos, _stream())
        if rc < 0:
            raise _ffi.BprxError(rc, "bprx_epoch_slots failed")
        return dict(epoch=epoch, perm=perm_d, epoch_ptr=epoch_ptr, pos_slot=pos_slot)

    def _start_epoch(self, epoch):
        nxt = getattr(self, "_next", None)
        cur = nxt if (nxt is not None and nxt["epoch"] == epoch) else self._prepare(epoch)
        self.perm, self.epoch_ptr, self.pos_slot = cur["perm"], cur["epoch_ptr"], cur["pos_slot"]
        self.epoch, self.pos_in_epoch = epoch, 0
        self._next = self._prepare(epoch + 1)

    def sample(self, B, first=None, out=None):
        if first is not None:
            raise ValueError("the epoch walk is a sequential stream")
        if getattr(self, "perm", None) is None:
            self._start_epoch(0)
        u, i, j = out if out is not None else tuple(torch.empty(B, dtype=torch.int32, device=self.device) for _ in range(3))
        done = 0
        while done < B:
            n = min(B - done, self.num_pos - self.pos_in_epoch)
            rc = self.lib.bprx_sample_epoch_h(getattr(self, "_handle", None), _ptr(self.indptr), _ptr(self.items), _ptr(self.perm),
                                              _ptr(self.epoch_ptr), _ptr(self.pos_slot), self.indptr.numel() - 1, self.num_items,
                                              self.seed, self.epoch, self.pos_in_epoch, n, _ptr(u[done:]), _ptr(i[done:]),
                                              _ptr(j[done:]), done, B, _stream())
            if rc < 0:
                raise _ffi.BprxError(rc, "bprx_sample_epoch failed")
            done += n
            self.pos_in_epoch += n
            if self.pos_in_epoch >= self.num_pos:
                self._start_epoch(self.epoch + 1)
        return u, i, j


class HostSampler:
    """bprx_sampler_*: the reference-compatible host index stream (dataset.py:83-114)."""

    def __init__(self, train_lists, num_items):
        self.lib = _ffi.lib()
        U = len(train_lists)
        indptr = np.zeros(U + 1, dtype=np.int64)
        for u, l in enumerate(train_lists):
            indptr[u + 1] = indptr[u] + len(l)
        items = np.fromiter((i for l in train_lists for i in l), dtype=np.int32, count=int(indptr[-1]))
        self.indptr, self.items = indptr, items
        s = C.c_void_p()
        rc = self.lib.bprx_sampler_create(indptr.ctypes.data, items.ctypes.data, U, num_items, C.byref(s))
        if rc < 0:
            raise _ffi.BprxError(rc, "bprx_sampler_create: invalid training lists (item id out of range?)")
        self.s = s

    def count(self, batch_size, epochs):
        return int(self.lib.bprx_sampler_count(self.s, batch_size, epochs))

    def ref_stream(self, batch_size, epochs, py_seed=0, np_seed=0):
        n = self.count(batch_size, epochs)
        u, i, j = (np.empty(n, np.int32) for _ in range(3))
        got = self.lib.bprx_sampler_ref_stream(self.s, batch_size, epochs, py_seed, np_seed,
                                               u.ctypes.data, i.ctypes.data, j.ctypes.data, n)
        if got < 0:
            raise _ffi.BprxError(int(got), "bprx_sampler_ref_stream failed (a user whose positives cover every item?)")
        return u[:got], i[:got], j[:got]

    def __del__(self):
        try:
            if getattr(self, "s", None):
                self.lib.bprx_sampler_destroy(self.s)
                self.s = None
        except Exception:
            pass
