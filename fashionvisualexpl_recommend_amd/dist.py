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
(tests/test_gpu_dist.py checks that against the CPU oracle with two ranks).  sgd, or adam_tf23: the engine steps the rows it
keeps and E|Bp, the owners of the routed rows step their whole shard from the summed returned gradients (bprx_adam_rows).
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
        """True if any bucket of any plan_fixed() / plan_native() since the last call overflowed (synchronises)."""
        f = bool(self._overflow.item()) if getattr(self, "_overflow", None) is not None else False
        self._overflow = None
        if getattr(self, "_nat", None) is not None:
            f = f or bool(self._nat["overflow"].item())
            self._nat["overflow"].zero_()
        return f

    # ---- the same fixed-capacity exchange with the routing in HIP kernels (include/bprx.h: bprx_route_*) ---------------
    # plan: one kernel (requests counted per owner in LDS, one cursor atomic per workgroup and owner); owners gather the
    # requested rows straight into the send buffer; the requester unpacks straight into the engine's staging tables; the
    # gradient rows are packed into the send buffer (and zeroed) by one kernel and added at the owners by one kernel.
    # Device tensors only (the torch forms above remain for CPU tensors: the gloo routing tests).
    def native_setup(self, device, cap, n_max, width):
        from . import _ffi
        W = self.world
        i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=device)
        ps = (int(width) + 3) & ~3                                # routed rows are padded to whole 16-byte pieces
        self._nat = dict(lib=_ffi.lib(), cap=int(cap), width=int(width), send_idx=i32(W * cap), cursor=i32(W), overflow=i32(1),
                         slot=i32(n_max), rows_out=torch.zeros((W * cap, ps), dtype=torch.float32, device=device),
                         grad_out=torch.zeros((W * cap, ps), dtype=torch.float32, device=device),
                         # row multiplicities (own requests / the other ranks' requests): a row that occurs once is added
                         # without float atomics; the kernels leave both arrays all-zero again (bprx_route.hip)
                         own_cnt=i32(self.ush), cnt=i32(self.ush))
        nat = self._nat
        self._rc(nat["lib"].bprx_route_reset(self._p(nat["send_idx"]), W * nat["cap"], self._p(nat["cursor"]), W, self._s()), "route_reset")

    @staticmethod
    def _p(t):
        import ctypes as C
        return None if t is None else C.c_void_p(t.data_ptr())

    @staticmethod
    def _s():
        import ctypes as C
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _rc(self, rc, what):
        if rc < 0:
            from . import _ffi
            raise _ffi.BprxError(rc, what + " failed")

    def plan_native(self, ids, ids_b=None):
        """ids (, ids_b): int32 device tensors of global row ids, taken back to back (a batch's positives and negatives need no
        concatenation).  Returns recv_idx [world*cap]: the rows the other ranks ask this rank for (-1 = unused slot); the slots of
        this rank's own requests stay in the exchange (fetch_native / give_back_native).  Rows this rank owns itself are served
        from / added to its own tables (no trip through the send buffers); with one rank nothing is routed and no collective runs.
        (The send list and the cursors were returned to "empty" by the previous step's give_back_native.)"""
        nat, W = self._nat, self.world
        n, nb = ids.numel(), (ids_b.numel() if ids_b is not None else 0)
        self._rc(nat["lib"].bprx_route_plan(self._p(ids) if n else None, n, self._p(ids_b) if nb else None, nb, self.ush, W, nat["cap"],
                                            self.rank, self._p(nat["slot"]), self._p(nat["send_idx"]), self._p(nat["cursor"]),
                                            self._p(nat["overflow"]), self._p(nat["own_cnt"]), self._s()), "route_plan")
        nat["n"] = n + nb
        return self._a2a_equal(nat["send_idx"]) if W > 1 else nat["send_idx"]

    def fetch_native(self, t0, t1, recv_idx, dst0, dst1):
        """Owners gather [t0 | t1] rows (t1 may be None) into the send buffer; after the all-to-all the requester's rows land
        in dst0 / dst1 (row r = request r; zero rows for requests that found their bucket full)."""
        nat = self._nat
        w0, w1 = t0.shape[1], (t1.shape[1] if t1 is not None else 0)
        got = nat["rows_out"]
        if self.world > 1:
            self._rc(nat["lib"].bprx_route_gather(self._p(t0), w0, self._p(t1), w1, t0.shape[0], self._p(recv_idx), recv_idx.numel(),
                                                  self._p(nat["rows_out"]), self._p(nat["cnt"]), self._s()), "route_gather")
            got = self._a2a_equal(nat["rows_out"])
        self._rc(nat["lib"].bprx_route_unpack(self._p(got), self._p(nat["slot"]), nat["n"], self._p(dst0), w0, self._p(dst1), w1,
                                              self._p(t0), self._p(t1), t0.shape[0], self._s()), "route_unpack")

    def give_back_native(self, g0, g1, recv_idx, t0, t1, scale):
        """The staging gradient rows (g0 | g1, row r = request r; returned to zero) travel to the owners, who add scale * row into
        their tables."""
        nat = self._nat
        w0, w1 = g0.shape[1], (g1.shape[1] if g1 is not None else 0)
        W = self.world
        self._rc(nat["lib"].bprx_route_pack(self._p(g0), w0, self._p(g1), w1, self._p(nat["slot"]), nat["n"], self._p(nat["grad_out"]),
                                            self._p(t0), self._p(t1), t0.shape[0], float(scale), self._p(nat["own_cnt"]),
                                            self._p(nat["send_idx"]), W * nat["cap"], self._p(nat["cursor"]), W, self._s()), "route_pack")
        if W > 1:
            back = self._a2a_equal(nat["grad_out"])
            self._rc(nat["lib"].bprx_route_scatter_add(self._p(t0), w0, self._p(t1), w1, t0.shape[0], self._p(recv_idx), self._p(back),
                                                       recv_idx.numel(), float(scale), self._p(nat["cnt"]), self._s()), "route_scatter_add")


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
                 max_batch, feat_dtype="bf16", group=None, device=None, fixed_cap=True, slack=2.0, optimizer="sgd"):
        """fixed_cap (default): the row exchange uses equal, fixed-capacity splits (UserRowExchange.plan_fixed): no host
        synchronisation inside the step; False: exact data-dependent splits (one `.cpu()` of the split sizes per step).
        optimizer: 'sgd' | 'adam_tf23' (native fixed-capacity path): the engine takes the Adam steps of its item rows and of
        E|Bp; the returned user-row gradients are summed into a gradient table of the owner's shard, and the owner takes the
        Adam step of its WHOLE shard (bprx_adam_rows: TF-2.3's Adam moves every row every step)."""
        from .engine import Engine, scatter_add, adam_rows
        self._scatter_add, self._adam_rows = scatter_add, adam_rows
        self.rank, self.world, self.group = rank, world, group
        self.lr = lr
        self.fixed_cap = fixed_cap
        self.adam = optimizer == "adam_tf23"
        if optimizer not in ("sgd", "adam_tf23"):
            raise ValueError("optimizer: 'sgd' | 'adam_tf23'")
        if self.adam and not fixed_cap:
            raise NotImplementedError("adam_tf23 in the all-to-all mode: the native fixed-capacity exchange only")
        self.cap = int(min(max_batch, -(-max_batch // world) * slack + 8))
        self.x = UserRowExchange(rank, world, users_total, group)
        k, d = Gu_shard.shape[1], Tu_shard.shape[1]
        self.eng = Engine(model="vbpr", num_users=max_batch, num_items=Gi_shard.shape[0], embed_k=k, embed_d=d,
                          feat_dim=F_shard.shape[1], feat_dtype=feat_dtype, optimizer=optimizer, lr=lr, reg=reg,
                          max_batch=max_batch, device=device, export_user_grad=True)
        dev = self.eng.device
        self.Gu_shard = Gu_shard.to(dev).contiguous()
        self.Tu_shard = Tu_shard.to(dev).contiguous()
        if self.adam:                                            # the owner's Adam state: moments + the summed gradients of a step
            self.adam_state = {n: tuple(torch.zeros_like(t) for _ in range(3)) for n, t in (("Gu", self.Gu_shard), ("Tu", self.Tu_shard))}
        self.stage_Gu = torch.zeros((max_batch, k), dtype=torch.float32, device=dev)
        self.stage_Tu = torch.zeros((max_batch, d), dtype=torch.float32, device=dev)
        self.eng.bind(Gu=self.stage_Gu, Gi=Gi_shard, Bi=Bi_shard, Tu=self.stage_Tu, F=F_shard, E=E, Bp=Bp)
        self.iota = torch.arange(max_batch, dtype=torch.int32, device=dev)
        self.dense = self.eng.dense_grad()
        self.native = bool(fixed_cap)                            # routing in HIP kernels (bprx_route_*)
        if self.native:
            self.x.native_setup(dev, self.cap, max_batch, k + d)

    def step(self, u_global, i_local, j_local, want_loss=False):
        """One global batch-synchronous step; every rank calls it with its own local batch (int32 device tensors)."""
        B = u_global.numel()
        if self.fixed_cap and self.native:
            return self._step_native(u_global, i_local, j_local, want_loss)
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


    def _dense_allreduce(self):
        if self.world > 1:
            if self.x.host_staged:
                h = self.dense.cpu()
                dist.all_reduce(h, group=self.group)
                self.dense.copy_(h)
            else:
                dist.all_reduce(self.dense, group=self.group)     # RCCL, 4*(D*d + D) bytes

    def _step_native(self, u_global, i_local, j_local, want_loss):
        """Fixed-capacity exchange with the routing in HIP kernels: plan, gather, unpack, pack, scatter-add are one launch each
        (round 2: ~30 torch passes, 0.71 ms per step with one rank against 0.24 ms of local step)."""
        B = u_global.numel()
        ridx = self.x.plan_native(u_global)
        self.eng.step_project()                                   # P = F.[E|Bp]: no user rows needed
        self.x.fetch_native(self.Gu_shard, self.Tu_shard, ridx, self.stage_Gu, self.stage_Tu)
        self.eng.step_begin(self.iota[:B], i_local, j_local)     # (rows arrive in batch order)
        self._dense_allreduce()
        loss = self.eng.step_end(want_loss=want_loss)
        dG, dT = self.eng.user_grad()
        if self.adam:
            (mG, vG, gG), (mT, vT, gT) = self.adam_state["Gu"], self.adam_state["Tu"]
            self.x.give_back_native(dG, dT, ridx, gG, gT, 1.0)                               # summed gradients of my users' rows
            lr_t = self.eng.step_lr()
            self._adam_rows(self.Gu_shard, mG, vG, gG, lr_t)
            self._adam_rows(self.Tu_shard, mT, vT, gT, lr_t)
        else:
            self.x.give_back_native(dG, dT, ridx, self.Gu_shard, self.Tu_shard, -self.lr)  # (also re-zeroes the gradient rows)
        self.eng.clear_user_marks(B)
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
    concatenation of all ranks' batches.  sgd, or adam_tf23 (see __init__)."""

    def __init__(self, rank, world, items_total, Gu_shard, Gi_shard, Bi_shard, lr, reg, max_batch, group=None, device=None,
                 fixed_cap=True, slack=2.0, optimizer="sgd"):
        """fixed_cap (default): the row exchange uses equal, fixed-capacity splits (RowExchange.plan_fixed: 2B requested rows
        spread over `world` owners, `slack` x the even share per owner): no host synchronisation inside the step, an
        overflowing bucket raises a device flag (x.overflowed()); False: exact data-dependent splits (one `.cpu()` of the
        split sizes per step)."""
        from .engine import Engine, scatter_add, adam_rows
        self._scatter_add, self._adam_rows = scatter_add, adam_rows
        self.rank, self.world, self.group, self.lr = rank, world, group, lr
        self.fixed_cap = fixed_cap
        # adam_tf23 (native fixed-capacity path): the engine (lazy form) steps the user rows; the item owners sum the returned
        # gradients into a gradient table of their shard and step the WHOLE shard (bprx_adam_rows), every global step
        self.adam = optimizer == "adam_tf23"
        if optimizer not in ("sgd", "adam_tf23"):
            raise ValueError("optimizer: 'sgd' | 'adam_tf23'")
        if self.adam and not fixed_cap:
            raise NotImplementedError("adam_tf23 in the all-to-all mode: the native fixed-capacity exchange only")
        self.cap = int(min(2 * max_batch, -(-2 * max_batch // world) * slack + 8))
        self.x = RowExchange(rank, world, items_total, group)
        k = Gu_shard.shape[1]
        self.eng = Engine(model="bprmf", num_users=Gu_shard.shape[0], num_items=2 * max_batch, embed_k=k, optimizer=optimizer,
                          lr=lr, reg=reg, max_batch=max_batch, device=device, export_item_grad=True)
        dev = self.eng.device
        self.Gi_shard = Gi_shard.to(dev).float().contiguous()                  # [Ish, k]
        self.Bi_col = Bi_shard.to(dev).float().reshape(-1, 1).contiguous()      # [Ish, 1]
        if self.adam:
            self.adam_state = {n: tuple(torch.zeros_like(t) for _ in range(3)) for n, t in (("Gi", self.Gi_shard), ("Bi", self.Bi_col))}
        self.stage_Gi = torch.zeros((2 * max_batch, k), dtype=torch.float32, device=dev)
        self.stage_Bi = torch.zeros(2 * max_batch, dtype=torch.float32, device=dev)
        self.eng.bind(Gu=Gu_shard, Gi=self.stage_Gi, Bi=self.stage_Bi)
        self.iota = torch.arange(2 * max_batch, dtype=torch.int32, device=dev)
        self.k = k
        self.native = bool(fixed_cap)                            # routing in HIP kernels (bprx_route_*)
        if self.native:
            self.x.native_setup(dev, self.cap, 2 * max_batch, k + 1)

    @property
    def Bi_shard(self):
        return self.Bi_col[:, 0]

    def step(self, u_local, i_global, j_global, want_loss=False, loss_out=None, loss_index=0):
        """loss_out / loss_index: see Engine.step.  An EMPTY batch (a rank whose users have no positives) still takes part in
        the three all-to-alls (native path)."""
        B, k = u_local.numel(), self.k
        if self.fixed_cap and self.native:
            # a routed row is [Gi row | Bi] (k + 1 floats, padded to k + 4): the gather / unpack / pack / scatter-add kernels
            # take the two tables as they are (w0 = k, w1 = 1)
            ridx = self.x.plan_native(i_global, j_global)        # (requests 0..B-1: the positives, B..2B-1: the negatives)
            self.x.fetch_native(self.Gi_shard, self.Bi_col, ridx, self.stage_Gi, self.stage_Bi)
            loss = None
            if B or self.adam:                                   # (adam: an empty batch is still a step -- every row moves)
                loss = self.eng.step(u_local, self.iota[:B], self.iota[B:2 * B], want_loss=want_loss, loss_out=loss_out,
                                     loss_index=loss_index)
            dG, dB = self.eng.item_grad()
            if self.adam:
                (mG, vG, gG), (mB, vB, gB) = self.adam_state["Gi"], self.adam_state["Bi"]
                self.x.give_back_native(dG, dB.view(-1, 1), ridx, gG, gB, 1.0)                       # summed gradients of my items' rows
                lr_t = self.eng.step_lr()
                self._adam_rows(self.Gi_shard, mG, vG, gG, lr_t)
                self._adam_rows(self.Bi_col, mB, vB, gB, lr_t)
            else:
                self.x.give_back_native(dG, dB.view(-1, 1), ridx, self.Gi_shard, self.Bi_col, -self.lr)   # (re-zeroes dG / dB rows)
            self.eng.clear_item_marks(2 * B)
            return loss
        items = torch.cat([i_global, j_global])                                   # 2B requested rows
        tabs = [self.Gi_shard, self.Bi_col]
        if self.fixed_cap:
            order, slot, valid, ridx = self.x.plan_fixed(items, self.cap)
            gi, bi = self.x.fetch_fixed(tabs, ridx, slot, valid)                  # [2B, k], [2B, 1] in batch order
            self.stage_Gi[:2 * B].copy_(gi)
            self.stage_Bi[:2 * B].copy_(bi[:, 0])
            loss = self.eng.step(u_local, self.iota[:B], self.iota[B:2 * B], want_loss=want_loss)
            dG, dB = self.eng.item_grad()
            g_back, b_back = self.x.give_back_fixed([dG[:2 * B], dB[:2 * B].reshape(-1, 1)], slot, valid, self.cap)
            self.eng.clear_item_grad(2 * B)
            self._scatter_add(self.Gi_shard, ridx, g_back.contiguous(), -self.lr)  # (index -1 = empty slot: skipped)
            self._scatter_add(self.Bi_col, ridx, b_back.contiguous(), -self.lr)
            return loss
        order, sc, rc, ridx = self.x.plan(items)
        gi, bi = self.x.fetch(tabs, ridx, sc, rc)                                 # owner-sorted order
        inv = torch.empty_like(order)
        inv[order] = torch.arange(order.numel(), device=order.device)             # back to batch order
        self.stage_Gi[:2 * B].copy_(gi.index_select(0, inv))
        self.stage_Bi[:2 * B].copy_(bi.index_select(0, inv)[:, 0])
        loss = self.eng.step(u_local, self.iota[:B], self.iota[B:2 * B], want_loss=want_loss)
        dG, dB = self.eng.item_grad()
        g_back, b_back = self.x.give_back([dG[:2 * B].index_select(0, order), dB[:2 * B].reshape(-1, 1).index_select(0, order)], sc, rc)
        self.eng.clear_item_grad(2 * B)
        self._scatter_add(self.Gi_shard, ridx, g_back.contiguous(), -self.lr)
        self._scatter_add(self.Bi_col, ridx, b_back.contiguous(), -self.lr)
        return loss
