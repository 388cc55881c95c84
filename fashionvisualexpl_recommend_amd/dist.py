"""Item-sharded multi-GPU VBPR (BASELINE.json configs[3]): one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm).

Partitioning
  items  range-partitioned: rank r owns Gi, Bi and the feature rows F of items [r*Ish, (r+1)*Ish) -- they never
         cross xGMI; negatives are sampled rank-locally; a positive (u, i) is processed on owner(i).
  users  range-partitioned: rank q owns rows [q*Ush, (q+1)*Ush) of Gu / Tu.  A step on rank r needs the rows of the
         users in ITS batch, wherever they live:  all-to-all fetch (row b of the staging tables = user row of
         triplet b) -> local step with BPRX_FLAG_EXPORT_USER_GRAD -> all-to-all return of the per-row gradients ->
         owners add  -lr * grad  into their shard (bprx_scatter_add; duplicates, within and across ranks, sum up).
  E, Bp  replicated; their dense gradient is all-reduced (sum) between bprx_step_begin and bprx_step_end, the only
         all-reduce of the step.  The fetch overlaps the item projection (bprx_step_project) and the gradient return
         overlaps nothing the next step needs before its own fetch.

The global step is exactly the single-GPU batch-synchronous step on the concatenation of all ranks' batches
(tests/test_gpu_dist.py checks that against the CPU oracle with two ranks).  sgd only in this mode.
"""
import torch
import torch.distributed as dist


def shard_size(total, world):
    return (total + world - 1) // world


class UserRowExchange:
    """Routing of table rows (user rows for item-sharded VBPR, item rows for user-sharded BPRMF) between the ranks that
    use them and the ranks that own them.  Pure tensor + collective logic (no kernels): works on CPU tensors with gloo
    (tests) and on device tensors with nccl."""

    def __init__(self, rank, world, users_total, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.ush = shard_size(users_total, world)
        # gloo has no all_to_all for device tensors: stage through the host in that case (test mode only)
        self.host_staged = dist.get_backend(group) != "nccl"

    def _a2a(self, inp, in_splits, out_splits, async_op=False):
        """all_to_all_single with split lists.  async_op: returns (out, work); the collective then runs on the
        communicator's stream beside whatever the caller enqueues next, until work.wait() (nccl only)."""
        out = inp.new_empty((sum(out_splits),) + tuple(inp.shape[1:]))
        if self.host_staged and inp.is_cuda:
            o, i = out.cpu(), inp.cpu()
            dist.all_to_all_single(o, i, out_splits, in_splits, group=self.group)
            out.copy_(o)
            return (out, None) if async_op else out
        work = dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=self.group, async_op=async_op)
        return (out, work) if async_op else out

    def plan(self, u_global):
        """u_global: int tensor [B] of global user ids used by this rank's batch.
        Returns (order, send_counts, recv_counts, recv_local_idx): `order` sorts the batch by owner rank;
        recv_local_idx are the shard-local row ids the other ranks ask this rank for (in rank order)."""
        owner = torch.div(u_global, self.ush, rounding_mode="floor").to(torch.int64)
        order = torch.argsort(owner, stable=True)
        counts = torch.bincount(owner, minlength=self.world)
        if self.host_staged or not counts.is_cuda:
            send = counts.cpu()
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=self.group)
        else:                                            # nccl moves device tensors only
            recv_dev = torch.empty_like(counts)
            dist.all_to_all_single(recv_dev, counts, group=self.group)
            send, recv = counts.cpu(), recv_dev.cpu()
        send_counts, recv_counts = send.tolist(), recv.tolist()
        local = (u_global[order] - owner[order] * self.ush).to(torch.int32)
        recv_local_idx = self._a2a(local, send_counts, recv_counts)
        return order, send_counts, recv_counts, recv_local_idx

    def fetch(self, shard_tables, recv_local_idx, send_counts, recv_counts, async_op=False):
        """Owners gather the requested rows of the shard tables (concatenated column-wise: ONE collective) and send
        them back; returns the fetched rows in batch-sorted order, one tensor per table (views of one buffer).
        async_op: returns (tensors, work) -- call work.wait() (if not None) before using them."""
        idx = recv_local_idx.long()
        widths = [t.shape[1] for t in shard_tables]
        packed = torch.cat([t.index_select(0, idx) for t in shard_tables], dim=1) if len(shard_tables) > 1 \
            else shard_tables[0].index_select(0, idx)
        res = self._a2a(packed, recv_counts, send_counts, async_op=async_op)
        out, work = res if async_op else (res, None)
        parts = list(torch.split(out, widths, dim=1))
        return (parts, work) if async_op else parts

    # ---- fixed-capacity form: no data-dependent split sizes, hence NO host synchronisation ---------------------------
    # Every rank sends exactly `cap` slots to every rank (a slot = one requested row id, -1 = empty), so the collectives
    # take equal splits and nothing has to be read back to size them.  cap = slack x ceil(B / world) (default slack 2: the
    # batch's users spread evenly over the owners up to sampling noise); a bucket that overflows is reported by
    # overflowed() -- a device flag, read when the caller next synchronises anyway -- and its surplus rows take part with
    # zero rows / dropped gradients in that step.
    def _a2a_equal(self, inp):
        out = torch.empty_like(inp)
        if self.host_staged and inp.is_cuda:
            o, i = out.cpu(), inp.cpu()
            dist.all_to_all_single(o, i, group=self.group)
            out.copy_(o)
            return out
        dist.all_to_all_single(out, inp.contiguous(), group=self.group)
        return out

    def plan_fixed(self, u_global, cap):
        """Returns (order, slot, valid, recv_idx): request r sits in send slot slot[r] (owner * cap + its position among the
        requests to that owner, in batch order) when valid[r]; recv_idx [world*cap] are the shard-local row ids the other
        ranks ask this rank for (-1 = empty slot).  `order` is the identity (kept for the callers' signature: rows come back
        in BATCH order).  No sort and nothing that reads a size back to the host: positions are a running count per owner
        (one-hot cumsum over the <= 8 owners), invalid requests are written to a dump slot instead of being masked out
        (boolean-mask indexing, bincount and nonzero all synchronise)."""
        dev = u_global.device
        n, W = u_global.numel(), self.world
        owner = torch.div(u_global, self.ush, rounding_mode="floor").to(torch.int64)
        run = (owner[:, None] == torch.arange(W, device=dev)[None, :]).to(torch.int32).cumsum(0)     # [n, W]
        pos = run.gather(1, owner[:, None]).squeeze(1).to(torch.int64) - 1
        valid = pos < cap
        slot = owner * cap + pos
        local = (u_global.to(torch.int64) - owner * self.ush).to(torch.int32)
        send = torch.full((W * cap + 1,), -1, dtype=torch.int32, device=dev)
        send.scatter_(0, torch.where(valid, slot, torch.full_like(slot, W * cap)), local)             # dump slot: W * cap
        flag = (run[-1] > cap).any()
        self._overflow = flag if getattr(self, "_overflow", None) is None else (self._overflow | flag)
        order = torch.arange(n, device=dev)
        return order, slot, valid, self._a2a_equal(send[:W * cap].contiguous())

    def overflowed(self):
        """True if any bucket of any plan_fixed() since the last call overflowed (synchronises)."""
        f = bool(self._overflow.item()) if getattr(self, "_overflow", None) is not None else False
        self._overflow = None
        return f

    def fetch_fixed(self, shard_tables, recv_idx, slot, valid):
        """Owners gather the requested rows (empty slots: zero rows) and send them back; returns the rows in batch-sorted
        order, one tensor per table."""
        ok = recv_idx >= 0
        idx = recv_idx.clamp(min=0).long()
        widths = [t.shape[1] for t in shard_tables]
        packed = torch.cat([t.index_select(0, idx) for t in shard_tables], dim=1) * ok[:, None].to(shard_tables[0].dtype)
        got = self._a2a_equal(packed)
        rows = got.index_select(0, slot.clamp(max=got.shape[0] - 1)) * valid[:, None].to(got.dtype)
        return list(torch.split(rows, widths, dim=1))

    def give_back_fixed(self, grad_rows, slot, valid, cap):
        """Per-row gradients (batch-sorted order) to the owners; returns the rows aligned with plan_fixed()'s recv_idx
        (rows of empty slots are zero and carry index -1: bprx_scatter_add skips them)."""
        widths = [g.shape[1] for g in grad_rows]
        packed = torch.cat(list(grad_rows), dim=1) if len(grad_rows) > 1 else grad_rows[0]
        send = torch.zeros((self.world * cap + 1, packed.shape[1]), dtype=packed.dtype, device=packed.device)
        send.index_copy_(0, torch.where(valid, slot, torch.full_like(slot, self.world * cap)), packed)   # (dump row: W * cap)
        return list(torch.split(self._a2a_equal(send[:self.world * cap]), widths, dim=1))

    def give_back(self, grad_rows, send_counts, recv_counts):
        """Send per-row gradients (batch-sorted order, concatenated column-wise: ONE collective) to the owners; returns
        the rows each owner received, aligned with the recv_local_idx of plan()."""
        widths = [g.shape[1] for g in grad_rows]
        packed = torch.cat(list(grad_rows), dim=1) if len(grad_rows) > 1 else grad_rows[0]
        return list(torch.split(self._a2a(packed, send_counts, recv_counts), widths, dim=1))


class ItemShardedVBPR:
    """Per-rank driver of the item-sharded VBPR step (see module docstring)."""

    def __init__(self, rank, world, users_total, Gu_shard, Tu_shard, Gi_shard, Bi_shard, F_shard, E, Bp, lr, reg,
                 max_batch, feat_dtype="bf16", group=None, device=None, fixed_cap=True, slack=2.0):
        """fixed_cap (default): the row exchange uses equal, fixed-capacity splits (UserRowExchange.plan_fixed): no host
        synchronisation inside the step; False: exact data-dependent splits (one `.cpu()` of the split sizes per step)."""
        from .engine import Engine, scatter_add
        self._scatter_add = scatter_add
        self.rank, self.world, self.group = rank, world, group
        self.lr = lr
        self.fixed_cap = fixed_cap
        self.cap = int(min(max_batch, -(-max_batch // world) * slack + 8))
        self.x = UserRowExchange(rank, world, users_total, group)
        k, d = Gu_shard.shape[1], Tu_shard.shape[1]
        self.eng = Engine(model="vbpr", num_users=max_batch, num_items=Gi_shard.shape[0], embed_k=k, embed_d=d,
                          feat_dim=F_shard.shape[1], feat_dtype=feat_dtype, optimizer="sgd", lr=lr, reg=reg,
                          max_batch=max_batch, device=device, export_user_grad=True)
        dev = self.eng.device
        self.Gu_shard = Gu_shard.to(dev).contiguous()
        self.Tu_shard = Tu_shard.to(dev).contiguous()
        self.stage_Gu = torch.zeros((max_batch, k), dtype=torch.float32, device=dev)
        self.stage_Tu = torch.zeros((max_batch, d), dtype=torch.float32, device=dev)
        self.eng.bind(Gu=self.stage_Gu, Gi=Gi_shard, Bi=Bi_shard, Tu=self.stage_Tu, F=F_shard, E=E, Bp=Bp)
        self.iota = torch.arange(max_batch, dtype=torch.int32, device=dev)
        self.dense = self.eng.dense_grad()

    def step(self, u_global, i_local, j_local, want_loss=False):
        """One global batch-synchronous step; every rank calls it with its own local batch (int32 device tensors)."""
        B = u_global.numel()
        if self.fixed_cap:
            return self._step_fixed(u_global, i_local, j_local, want_loss)
        order, sc, rc, ridx = self.x.plan(u_global)
        (gu, tu), work = self.x.fetch([self.Gu_shard, self.Tu_shard], ridx, sc, rc, async_op=True)
        self.eng.step_project()                                   # P = F.[E|Bp] runs beside the row fetch (xGMI)
        if work is not None:
            work.wait()
        self.stage_Gu[:B].copy_(gu)
        self.stage_Tu[:B].copy_(tu)
        i_s, j_s = i_local[order].contiguous(), j_local[order].contiguous()
        self.eng.step_begin(self.iota[:B], i_s, j_s)
        if self.world > 1:
            if self.x.host_staged:
                h = self.dense.cpu()
                dist.all_reduce(h, group=self.group)
                self.dense.copy_(h)
            else:
                dist.all_reduce(self.dense, group=self.group)     # RCCL, 4*(D*d + D) bytes
        loss = self.eng.step_end(want_loss=want_loss)
        dG, dT = self.eng.user_grad()
        g_back, t_back = self.x.give_back([dG[:B], dT[:B]], sc, rc)
        self.eng.clear_user_grad(B)
        self._scatter_add(self.Gu_shard, ridx, g_back.contiguous(), -self.lr)
        self._scatter_add(self.Tu_shard, ridx, t_back.contiguous(), -self.lr)
        return loss


    def _step_fixed(self, u_global, i_local, j_local, want_loss):
        """The same global step with fixed-capacity exchanges: every tensor op and collective is enqueued without reading
        anything back to the host."""
        B = u_global.numel()
        order, slot, valid, ridx = self.x.plan_fixed(u_global, self.cap)
        self.eng.step_project()                                   # P = F.[E|Bp]: no user rows needed
        gu, tu = self.x.fetch_fixed([self.Gu_shard, self.Tu_shard], ridx, slot, valid)
        self.stage_Gu[:B].copy_(gu)
        self.stage_Tu[:B].copy_(tu)
        self.eng.step_begin(self.iota[:B], i_local, j_local)     # (rows arrive in batch order: nothing to permute)
        if self.world > 1:
            if self.x.host_staged:
                h = self.dense.cpu()
                dist.all_reduce(h, group=self.group)
                self.dense.copy_(h)
            else:
                dist.all_reduce(self.dense, group=self.group)     # RCCL, 4*(D*d + D) bytes
        loss = self.eng.step_end(want_loss=want_loss)
        dG, dT = self.eng.user_grad()
        g_back, t_back = self.x.give_back_fixed([dG[:B], dT[:B]], slot, valid, self.cap)
        self.eng.clear_user_grad(B)
        self._scatter_add(self.Gu_shard, ridx, g_back.contiguous(), -self.lr)
        self._scatter_add(self.Tu_shard, ridx, t_back.contiguous(), -self.lr)
        return loss


class ReplicatedUserVBPR:
    """Item-sharded VBPR with REPLICATED user tables (SURVEY 8(e), C4 option "users replicated + sparse delta
    all-gather"): rank r owns an item shard (Gi, Bi, F never cross xGMI; negatives are local) and a full copy of Gu / Tu.
    A step:  bprx_step_begin_sparse on the local batch (global user ids; user gradients are summed per user into the
    staging tables instead of being applied) -> bprx_pack_user_msg: one fixed-size message per rank
    [count | ids | dGu rows | dTu rows] -> all_gather_into_tensor, asynchronous, beside bprx_step_begin_dense (item rows, W,
    backward projection) -> the dense gradient's exchange (all-gather + ordered sum, or RCCL all-reduce) ->
    bprx_apply_user_msgs: every replica adds every rank's rows per user in rank order (bit-identical replicas) ->
    bprx_step_end.
    No data-dependent split sizes, hence no host synchronisation and no all-to-all; the message holds `user_cap` distinct
    users per batch (epoch-walk batches of B triplets touch about B / positives-per-user of them; more than user_cap is
    reported by sync_check()).  The global step equals the single-GPU batch-synchronous step on the concatenation of all
    ranks' batches (tests/test_gpu_dist.py)."""

    def __init__(self, rank, world, Gu, Tu, Gi_shard, Bi_shard, F_shard, E, Bp, lr, reg, max_batch, user_cap=None,
                 feat_dtype="bf16", group=None, device=None, optimizer="sgd", dense_reduce="gather", overlap=True):
        """optimizer: 'sgd' | 'adam_tf23' (lazy-exact: every replica sums the ranks' rows per user in rank order and takes
        the same Adam step, so the replicas stay bit-identical).
        dense_reduce: 'gather' = the ranks' dE|dBp are all-gathered and summed in rank order (bit-identical replicas by
        construction); 'allreduce' = an RCCL all-reduce(sum) of the dense gradient, the form north_star names --
        (N-1)/N instead of N-1 message-sized transfers per rank for the dense part.
        overlap: True = the step runs in two halves (bprx_step_begin_sparse / _dense): the user rows are packed and their
        all-gather is started right after the per-triplet gradients, so it travels over xGMI WHILE item rows, W and the
        backward projection dE|dBp = F^T W (about a third of the step) are computed; only the 1-MB dense exchange is
        exposed.  False = the round-1 order: one message [user rows | dE|dBp] after the whole of bprx_step_begin."""
        from .engine import Engine
        self.rank, self.world, self.group, self.lr = rank, world, group, lr
        self.dense_reduce, self.overlap = dense_reduce, bool(overlap)
        if dense_reduce not in ("gather", "allreduce"):
            raise ValueError("dense_reduce: 'gather' | 'allreduce'")
        k, d = Gu.shape[1], Tu.shape[1]
        separate_dense = self.overlap or dense_reduce == "allreduce"      # the message carries the user rows only
        self.eng = Engine(model="vbpr", num_users=Gu.shape[0], num_items=Gi_shard.shape[0], embed_k=k, embed_d=d,
                          feat_dim=F_shard.shape[1], feat_dtype=feat_dtype, optimizer=optimizer, lr=lr, reg=reg,
                          max_batch=max_batch, device=device, export_user_grad=True, dense_allreduce=separate_dense)
        self.eng.bind(Gu=Gu, Gi=Gi_shard, Bi=Bi_shard, Tu=Tu, F=F_shard, E=E, Bp=Bp)
        dev = self.eng.device
        self.dense = self.eng.dense_grad() if separate_dense else None
        self.cap = int(user_cap if user_cap is not None else max_batch)
        n = self.eng.user_msg_floats(self.cap)
        self.msg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.msgs = torch.zeros(world * n, dtype=torch.float32, device=dev)
        self.dparts = (torch.zeros(world * self.dense.numel(), dtype=torch.float32, device=dev)
                       if separate_dense and dense_reduce == "gather" else None)
        self.live = dist.is_initialized()
        self.host_staged = self.live and dist.get_backend(group) != "nccl"     # gloo: test mode

    @property
    def Gu(self):
        return self.eng.t["Gu"]

    @property
    def Tu(self):
        return self.eng.t["Tu"]

    # -- collectives: RCCL (asynchronous: the returned work is waited for where the result is needed), or staged through
    #    the host for the gloo test mode, or a plain copy without a process group
    def _all_gather(self, out, inp):
        if not self.live:
            out.copy_(inp)
            return None
        if self.host_staged:
            h = inp.cpu()
            parts = [torch.empty_like(h) for _ in range(self.world)]
            dist.all_gather(parts, h, group=self.group)
            out.copy_(torch.cat(parts))
            return None
        return dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True)

    def _all_reduce(self, t):
        if not self.live:
            return None
        if self.host_staged:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
            return None
        return dist.all_reduce(t, group=self.group, async_op=True)

    def step(self, u_global, i_local, j_local, want_loss=False, loss_out=None, loss_index=0):
        """One global step.  A rank whose item shard holds no positive of this batch passes EMPTY index tensors: it sends a
        count-0 message and a zero dense gradient and still takes part in every collective (bprx_step_begin_sparse, B = 0).
        loss_out / loss_index: see Engine.step_end."""
        eng = self.eng
        end = lambda: eng.step_end(want_loss=want_loss, loss_out=loss_out, loss_index=loss_index)
        if not self.overlap:
            eng.step_begin(u_global, i_local, j_local)
            eng.pack_user_msg(u_global, self.cap, self.msg)
            wd = self._all_reduce(self.dense) if self.dense is not None and self.world > 1 else None
            wm = self._all_gather(self.msgs, self.msg)
            for w in (wd, wm):
                if w is not None:
                    w.wait()
            eng.apply_user_msgs(self.msgs, self.world, self.cap, -self.lr)
            return end()
        eng.step_begin_sparse(u_global, i_local, j_local)          # ... per-triplet gradients: user rows are final
        eng.pack_user_msg(u_global, self.cap, self.msg)
        wm = self._all_gather(self.msgs, self.msg)                  # in flight beside:
        eng.step_begin_dense()                                      # item rows, W, dE|dBp = F^T W
        if self.dense_reduce == "allreduce":
            wd = self._all_reduce(self.dense) if self.world > 1 else None
        else:
            wd = self._all_gather(self.dparts, self.dense)
        if wm is not None:
            wm.wait()
        eng.apply_user_msgs(self.msgs, self.world, self.cap, -self.lr)
        if wd is not None:
            wd.wait()
        if self.dense_reduce == "gather":
            eng.sum_dense_parts(self.dparts, self.world)
        return end()


RowExchange = UserRowExchange      # the routing is the same whichever table is the remote one


class UserShardedBPRMF:
    """Per-rank driver of user-sharded BPRMF (BASELINE.json configs[2]): rank r owns the Gu rows of its users and samples
    only its own users' positives, so user rows never move; Gi/Bi are range-partitioned by item id and the rows of the
    batch's positive and negative items are fetched from / their gradients returned to the item owners by all-to-all
    (staging row b = positive item of triplet b, row B+b = its negative item; BPRX_FLAG_EXPORT_ITEM_GRAD).
    No dense parameter, hence no all-reduce at all.  The global step equals the single-GPU batch-synchronous step on the
    concatenation of all ranks' batches.  sgd only."""

    def __init__(self, rank, world, items_total, Gu_shard, Gi_shard, Bi_shard, lr, reg, max_batch, group=None, device=None,
                 fixed_cap=True, slack=2.0):
        """fixed_cap (default): the row exchange uses equal, fixed-capacity splits (RowExchange.plan_fixed: 2B requested rows
        spread over `world` owners, `slack` x the even share per owner): no host synchronisation inside the step, an
        overflowing bucket raises a device flag (x.overflowed()); False: exact data-dependent splits (one `.cpu()` of the
        split sizes per step)."""
        from .engine import Engine, scatter_add
        self._scatter_add = scatter_add
        self.rank, self.world, self.group, self.lr = rank, world, group, lr
        self.fixed_cap = fixed_cap
        self.cap = int(min(2 * max_batch, -(-2 * max_batch // world) * slack + 8))
        self.x = RowExchange(rank, world, items_total, group)
        k = Gu_shard.shape[1]
        self.eng = Engine(model="bprmf", num_users=Gu_shard.shape[0], num_items=2 * max_batch, embed_k=k, optimizer="sgd",
                          lr=lr, reg=reg, max_batch=max_batch, device=device, export_item_grad=True)
        dev = self.eng.device
        self.GiBi_shard = torch.cat([Gi_shard.to(dev).float(), Bi_shard.to(dev).float().reshape(-1, 1)], dim=1).contiguous()
        self.stage_Gi = torch.zeros((2 * max_batch, k), dtype=torch.float32, device=dev)
        self.stage_Bi = torch.zeros(2 * max_batch, dtype=torch.float32, device=dev)
        self.eng.bind(Gu=Gu_shard, Gi=self.stage_Gi, Bi=self.stage_Bi)
        self.iota = torch.arange(2 * max_batch, dtype=torch.int32, device=dev)
        self.k = k

    @property
    def Gi_shard(self):
        return self.GiBi_shard[:, :self.k]

    @property
    def Bi_shard(self):
        return self.GiBi_shard[:, self.k]

    def step(self, u_local, i_global, j_global, want_loss=False):
        B, k = u_local.numel(), self.k
        items = torch.cat([i_global, j_global])                                   # 2B requested rows
        if self.fixed_cap:
            order, slot, valid, ridx = self.x.plan_fixed(items, self.cap)
            (rows,) = self.x.fetch_fixed([self.GiBi_shard], ridx, slot, valid)    # [2B, k+1] in batch order
            self.stage_Gi[:2 * B].copy_(rows[:, :k])
            self.stage_Bi[:2 * B].copy_(rows[:, k])
            loss = self.eng.step(u_local, self.iota[:B], self.iota[B:2 * B], want_loss=want_loss)
            dG, dB = self.eng.item_grad()
            g = torch.cat([dG[:2 * B], dB[:2 * B].reshape(-1, 1)], dim=1)
            (back,) = self.x.give_back_fixed([g], slot, valid, self.cap)
            self.eng.clear_item_grad(2 * B)
            self._scatter_add(self.GiBi_shard, ridx, back.contiguous(), -self.lr)  # (index -1 = empty slot: skipped)
            return loss
        order, sc, rc, ridx = self.x.plan(items)
        (rows,) = self.x.fetch([self.GiBi_shard], ridx, sc, rc)                   # [2B, k+1] in owner-sorted order
        inv = torch.empty_like(order)
        inv[order] = torch.arange(order.numel(), device=order.device)             # back to batch order
        rows = rows.index_select(0, inv)
        self.stage_Gi[:2 * B].copy_(rows[:, :k])
        self.stage_Bi[:2 * B].copy_(rows[:, k])
        loss = self.eng.step(u_local, self.iota[:B], self.iota[B:2 * B], want_loss=want_loss)
        dG, dB = self.eng.item_grad()
        g = torch.cat([dG[:2 * B], dB[:2 * B].reshape(-1, 1)], dim=1).index_select(0, order)   # owner-sorted order again
        (back,) = self.x.give_back([g], sc, rc)
        self.eng.clear_item_grad(2 * B)
        self._scatter_add(self.GiBi_shard, ridx, back.contiguous(), -self.lr)
        return loss
