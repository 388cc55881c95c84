"""Item-sharded VBPR with two ranks on ONE GPU (gloo for the collectives, HIP kernels for everything else): after a few
global steps the union of the shards must equal the CPU oracle stepped on the concatenation of both ranks' batches."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, dtype, opt="sgd"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fashionvisualexpl_recommend_amd import synth
        from fashionvisualexpl_recommend_amd.dist import ItemShardedVBPR, shard_size
        from oracle import oracle as orc
        torch.cuda.set_device(0)
        U, I, k, d, D, B, lr, reg = 45, 64, 8, 20, 256, 96, (0.05 if opt == "sgd" else 0.01), 1e-3
        ush, ish = shard_size(U, world), I // world
        rs = np.random.RandomState(1)
        F = synth.make_features(I, D, seed=1)
        F = (F / np.abs(F).max()).astype(np.float32)
        if dtype == "bf16":
            F = orc.bf16_round(F)
        t = dict(Gu=synth.glorot_uniform(rs, U, k), Gi=synth.glorot_uniform(rs, I, k),
                 Bi=(rs.standard_normal(I) * 0.01).astype(np.float32), Tu=synth.glorot_uniform(rs, U, d), F=F,
                 E=synth.glorot_uniform(rs, D, d), Bp=synth.glorot_uniform(rs, D, 1).reshape(-1))
        us, it = slice(rank * ush, min(U, (rank + 1) * ush)), slice(rank * ish, (rank + 1) * ish)
        c = lambda a: torch.as_tensor(a.copy())
        m = ItemShardedVBPR(rank, world, U, c(t["Gu"][us]), c(t["Tu"][us]), c(t["Gi"][it]), c(t["Bi"][it]), c(t["F"][it]),
                            c(t["E"]), c(t["Bp"]), lr, reg, max_batch=B, feat_dtype=dtype, device=0, optimizer=opt)
        o = orc.OracleModel(**t, quant=1 if dtype == "bf16" else 0)
        for step in range(3 if opt == "sgd" else 5):
            batches = []
            for r in range(world):                        # every rank knows every batch (test only) to feed the oracle
                br = np.random.RandomState(100 + step * world + r)
                nb = B - 10 * r                           # ragged
                if opt != "sgd" and step == 1 and r == 1:
                    nb = 0                                 # adam: an empty batch is still a step (every row moves)
                batches.append((br.randint(U, size=nb).astype(np.int32), br.randint(ish, size=nb).astype(np.int32),
                                br.randint(ish, size=nb).astype(np.int32)))
            if opt != "sgd" and step >= 2:                # adam: users that appear in nobody's batch keep moving all the same
                batches = [(b[0] % (10 + 7 * r) + 20 * r, b[1], b[2]) for r, b in enumerate(batches)]
            u, i, j = batches[rank]
            dev = lambda a: torch.as_tensor(a, device="cuda")
            m.step(dev(u), dev(i), dev(j))
            gu = np.concatenate([b[0] for b in batches])
            gi = np.concatenate([b[1] + r * ish for r, b in enumerate(batches)])
            gj = np.concatenate([b[2] + r * ish for r, b in enumerate(batches)])
            o.step(gu, gi, gj, opt, lr, reg)
        m.eng.sync_check()
        rt, at = (2e-5, 2e-6) if dtype == "fp32" else (2e-3, 1e-4)
        if opt != "sgd":
            at = max(at, 2e-3 * lr)                        # see test_gpu_parity.test_bprmf_steps_match_oracle
        chk = lambda got, want, n: np.testing.assert_allclose(got.cpu().numpy(), want, rtol=rt, atol=at, err_msg=n)
        chk(m.Gu_shard, o.Gu[us], "Gu shard")
        chk(m.Tu_shard, o.Tu[us], "Tu shard")
        chk(m.eng.t["Gi"], o.Gi[it], "Gi shard")
        chk(m.eng.t["Bi"], o.Bi[it], "Bi shard")
        chk(m.eng.t["E"], o.E, "E (replicated)")
        chk(m.eng.t["Bp"], o.Bp, "Bp (replicated)")
        # replicas of the shared parameters must agree BIT-exactly across ranks (same all-reduced gradient, same update)
        e = [torch.zeros_like(m.eng.t["E"].cpu()) for _ in range(world)]
        dist.all_gather(e, m.eng.t["E"].cpu())
        assert all(torch.equal(e[0], x) for x in e)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_item_sharded_vbpr_two_ranks_match_oracle(dtype):
    mp.spawn(_worker, args=(2, _free_port(), dtype), nprocs=2, join=True)


def test_item_sharded_vbpr_two_ranks_adam_tf23_match_oracle():
    """The reference's optimizer through the all-to-all mode: the engine (lazy form) steps its item rows and E|Bp, the owners
    of the user rows take the Adam step of their whole shard from the summed returned gradients (bprx_adam_rows) -- including
    a step in which one rank's batch is empty and users that nobody touches for several steps."""
    mp.spawn(_worker, args=(2, _free_port(), "fp32", "adam_tf23"), nprocs=2, join=True)


def _worker_bprmf(rank, world, port, fixed_cap=True, opt="sgd"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fashionvisualexpl_recommend_amd import synth
        from fashionvisualexpl_recommend_amd.dist import UserShardedBPRMF, shard_size
        from oracle import oracle as orc
        torch.cuda.set_device(0)
        U, I, k, B, lr, reg = 60, 75, 16, 128, (0.05 if opt == "sgd" else 0.01), 1e-3
        ush, ish = U // world, shard_size(I, world)
        rs = np.random.RandomState(2)
        t = dict(Gu=synth.glorot_uniform(rs, U, k), Gi=synth.glorot_uniform(rs, I, k),
                 Bi=(rs.standard_normal(I) * 0.01).astype(np.float32))
        us, it = slice(rank * ush, (rank + 1) * ush), slice(rank * ish, min(I, (rank + 1) * ish))
        c = lambda a: torch.as_tensor(a.copy())
        m = UserShardedBPRMF(rank, world, I, c(t["Gu"][us]), c(t["Gi"][it]), c(t["Bi"][it]), lr, reg, max_batch=B, device=0,
                             fixed_cap=fixed_cap, optimizer=opt)
        o = orc.OracleModel(**t)
        for step in range(3 if opt == "sgd" else 5):
            batches = []
            for r in range(world):
                br = np.random.RandomState(200 + step * world + r)
                nb = B - 17 * r
                if opt != "sgd" and step == 1 and r == 1:
                    nb = 0                                 # adam: an empty batch is still a step (every row moves)
                batches.append((br.randint(ush, size=nb).astype(np.int32), br.randint(I, size=nb).astype(np.int32),
                                br.randint(I, size=nb).astype(np.int32)))
            if opt != "sgd" and step >= 2:                # adam: items and users nobody touches for several steps keep moving
                batches = [(b[0] % 11, b[1] % 40, b[2] % 40) for b in batches]
            for b in batches:
                if b[0].size:
                    b[1][:4] = 7                           # the same remote/local item several times, also as negative
                    b[2][4:6] = 7
            u, i, j = batches[rank]
            dev = lambda a: torch.as_tensor(a, device="cuda")
            m.step(dev(u), dev(i), dev(j))
            o.step(np.concatenate([b[0] + r * ush for r, b in enumerate(batches)]), np.concatenate([b[1] for b in batches]),
                   np.concatenate([b[2] for b in batches]), opt, lr, reg)
        m.eng.sync_check()
        if fixed_cap:
            assert not m.x.overflowed()
        at = 2e-6 if opt == "sgd" else 2e-3 * lr
        chk = lambda got, want, n: np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-5, atol=at, err_msg=n)
        chk(m.eng.t["Gu"], o.Gu[us], "Gu shard")
        chk(m.Gi_shard, o.Gi[it], "Gi shard")
        chk(m.Bi_shard, o.Bi[it], "Bi shard")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fixed_cap", [True, False])
def test_user_sharded_bprmf_two_ranks_match_oracle(fixed_cap):
    """fixed_cap: equal, fixed-capacity all-to-all splits (no host synchronisation inside the step); False: exact splits."""
    mp.spawn(_worker_bprmf, args=(2, _free_port(), fixed_cap), nprocs=2, join=True)


def test_user_sharded_bprmf_two_ranks_adam_tf23_match_oracle():
    """adam_tf23 through the user-sharded all-to-all mode (the CLI's default optimizer with --shard user): the engine steps
    the user rows, the item owners step their whole Gi / Bi shard from the summed returned gradients."""
    mp.spawn(_worker_bprmf, args=(2, _free_port(), True, "adam_tf23"), nprocs=2, join=True)


def _worker_replicated(rank, world, port, dtype, opt="sgd", dense_reduce="gather", overlap=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fashionvisualexpl_recommend_amd import synth
        from fashionvisualexpl_recommend_amd.dist import ReplicatedUserVBPR
        from oracle import oracle as orc
        torch.cuda.set_device(0)
        U, I, k, d, D, B, lr, reg = 45, 64, 8, 20, 256, 96, (0.05 if opt == "sgd" else 0.01), 1e-3
        ish = I // world
        rs = np.random.RandomState(1)
        F = synth.make_features(I, D, seed=1)
        F = (F / np.abs(F).max()).astype(np.float32)
        if dtype == "bf16":
            F = orc.bf16_round(F)
        t = dict(Gu=synth.glorot_uniform(rs, U, k), Gi=synth.glorot_uniform(rs, I, k),
                 Bi=(rs.standard_normal(I) * 0.01).astype(np.float32), Tu=synth.glorot_uniform(rs, U, d), F=F,
                 E=synth.glorot_uniform(rs, D, d), Bp=synth.glorot_uniform(rs, D, 1).reshape(-1))
        it = slice(rank * ish, (rank + 1) * ish)
        c = lambda a: torch.as_tensor(a.copy())
        m = ReplicatedUserVBPR(rank, world, c(t["Gu"]), c(t["Tu"]), c(t["Gi"][it]), c(t["Bi"][it]), c(t["F"][it]), c(t["E"]),
                               c(t["Bp"]), lr, reg, max_batch=B, user_cap=U, feat_dtype=dtype, device=0, optimizer=opt,
                               dense_reduce=dense_reduce, overlap=overlap)
        o = orc.OracleModel(**t, quant=1 if dtype == "bf16" else 0)
        nsteps = 4 if opt != "sgd" else 3
        all_batches = []
        for step in range(nsteps):
            batches = []
            for r in range(world):
                br = np.random.RandomState(300 + step * world + r)
                nb = B - 10 * r                           # ragged
                if step == 1 and r == 1:
                    nb = 0                                 # a rank without local positives: EMPTY batch, every collective joined
                batches.append((br.randint(U, size=nb).astype(np.int32), br.randint(ish, size=nb).astype(np.int32),
                                br.randint(ish, size=nb).astype(np.int32)))
            if opt != "sgd" and step >= 2:                # adam: some users only in the OTHER rank's batch, some in nobody's
                batches = [(b[0] % (10 + 7 * r) + 20 * r, b[1], b[2]) for r, b in enumerate(batches)]
            all_batches.append(batches)
        dev = lambda a: torch.as_tensor(a, device="cuda")
        mine = [tuple(dev(x) for x in bs[rank]) for bs in all_batches]
        for step in range(nsteps):
            batches = all_batches[step]
            m.step(*mine[step])
            o.step(np.concatenate([b[0] for b in batches]), np.concatenate([b[1] + r * ish for r, b in enumerate(batches)]),
                   np.concatenate([b[2] + r * ish for r, b in enumerate(batches)]), opt, lr, reg)
        m.eng.sync_check()
        rt, at = (2e-5, 2e-6) if dtype == "fp32" else (2e-3, 1e-4)
        if opt != "sgd":
            at = max(at, 2e-3 * lr)                        # see test_gpu_parity.test_bprmf_steps_match_oracle
        chk = lambda got, want, n: np.testing.assert_allclose(got.cpu().numpy(), want, rtol=rt, atol=at, err_msg=n)
        chk(m.Gu, o.Gu, "Gu (replicated)")
        chk(m.Tu, o.Tu, "Tu (replicated)")
        chk(m.eng.t["Gi"], o.Gi[it], "Gi shard")
        chk(m.eng.t["Bi"], o.Bi[it], "Bi shard")
        chk(m.eng.t["E"], o.E, "E (replicated)")
        chk(m.eng.t["Bp"], o.Bp, "Bp (replicated)")
        # every replicated table must agree BIT-exactly across ranks (same additions in the same order everywhere)
        for n in ("Gu", "Tu", "E", "Bp"):
            mine = m.eng.t[n].cpu()
            parts = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            assert all(torch.equal(parts[0], x) for x in parts), n
        # the staging tables are left all-zero for the next step
        g, tt = m.eng.user_grad()
        assert float(g.abs().max()) == 0.0 and float(tt.abs().max()) == 0.0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_replicated_user_vbpr_two_ranks_match_oracle(dtype):
    mp.spawn(_worker_replicated, args=(2, _free_port(), dtype), nprocs=2, join=True)


def test_replicated_user_vbpr_two_ranks_adam_tf23_match_oracle():
    """The reference's optimizer (Adam, VBPR.py:56,142) through the replicated-user multi-GPU step: lazy-exact rows, users
    summed over the ranks' messages in rank order, replicas bit-identical."""
    mp.spawn(_worker_replicated, args=(2, _free_port(), "fp32", "adam_tf23"), nprocs=2, join=True)


def test_replicated_user_vbpr_dense_allreduce_form():
    """dense_reduce='allreduce': dE|dBp leaves the message and is summed by a collective all-reduce (north_star's form)."""
    mp.spawn(_worker_replicated, args=(2, _free_port(), "fp32", "sgd", "allreduce"), nprocs=2, join=True)


@pytest.mark.parametrize("opt,dense_reduce", [("sgd", "gather"), ("adam_tf23", "gather"), ("sgd", "allreduce")])
def test_replicated_user_vbpr_single_message_order(opt, dense_reduce):
    """overlap=False: the round-1 order (the whole of bprx_step_begin, then ONE message [user rows | dE|dBp])."""
    mp.spawn(_worker_replicated, args=(2, _free_port(), "fp32", opt, dense_reduce, False), nprocs=2, join=True)


@pytest.mark.parametrize("opt", ["sgd", "adam_tf23"])
def test_replicated_single_rank_equals_plain_step(opt):
    """One rank, no process group: pack (many workgroups, duplicate users, one cursor atomic per workgroup) -> chain ->
    apply must give the plain single-GPU step."""
    from fashionvisualexpl_recommend_amd import synth
    from fashionvisualexpl_recommend_amd.dist import ReplicatedUserVBPR
    from fashionvisualexpl_recommend_amd.engine import Engine
    rs = np.random.RandomState(11)
    U, I, k, d, D, B = 3000, 700, 16, 12, 128, 5000
    F = synth.make_features(I, D, seed=3)
    F = (F / np.abs(F).max()).astype(np.float32)
    t = dict(Gu=synth.glorot_uniform(rs, U, k), Gi=synth.glorot_uniform(rs, I, k), Bi=(rs.standard_normal(I) * 0.01).astype(np.float32),
             Tu=synth.glorot_uniform(rs, U, d), F=F, E=synth.glorot_uniform(rs, D, d), Bp=synth.glorot_uniform(rs, D, 1).reshape(-1))
    c = lambda a: torch.as_tensor(a.copy())
    lr, reg = (0.05, 1e-3) if opt == "sgd" else (0.01, 1e-3)
    m = ReplicatedUserVBPR(0, 1, c(t["Gu"]), c(t["Tu"]), c(t["Gi"]), c(t["Bi"]), c(t["F"]), c(t["E"]), c(t["Bp"]), lr, reg,
                           max_batch=B, user_cap=U, feat_dtype="fp32", device=0, optimizer=opt)
    e = Engine(model="vbpr", num_users=U, num_items=I, embed_k=k, embed_d=d, feat_dim=D, feat_dtype="fp32", optimizer=opt, lr=lr,
               reg=reg, max_batch=B, device=0).bind(**{n: c(v) for n, v in t.items()})
    for step in range(3):
        br = np.random.RandomState(40 + step)
        nb = B - 37 * step
        u = torch.as_tensor(br.randint(U if step != 1 else 50, size=nb).astype(np.int32), device="cuda")   # step 1: hot users
        i = torch.as_tensor(br.randint(I, size=nb).astype(np.int32), device="cuda")
        j = torch.as_tensor(br.randint(I, size=nb).astype(np.int32), device="cuda")
        m.step(u, i, j)
        e.step(u, i, j)
    m.eng.sync_check()
    e.sync_check()
    # adam: the first step is lr * g / (|g| + eps / sqrt(1 - beta2)) -- an element whose summed gradient is ~3e-6 moves by a
    # visible fraction of lr when the two paths' summation orders (staging atomics) differ by 1e-7: a handful of the 48 000
    # user elements may differ by a few per cent of lr, none by more than a fraction of a step
    for n in ("Gu", "Tu", "Gi", "Bi", "E", "Bp"):
        got, want = m.eng.t[n].cpu().numpy(), e.t[n].cpu().numpy()
        if opt == "sgd":
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6, err_msg=n)
        else:
            diff = np.abs(got - want)
            assert float((diff > 2e-3 * lr + 2e-5 * np.abs(want)).mean()) < 1e-3, n
            assert float(diff.max()) < 0.25 * lr, n
    g, tt = m.eng.user_grad()
    assert float(g.abs().max()) == 0.0 and float(tt.abs().max()) == 0.0


def test_replicated_user_message_overflow_is_reported():
    """More distinct users in a batch than the message holds: reported by sync_check, never a fault."""
    from fashionvisualexpl_recommend_amd import _ffi, synth
    from fashionvisualexpl_recommend_amd.dist import ReplicatedUserVBPR
    rs = np.random.RandomState(5)
    U, I, k, d, D, B = 40, 32, 8, 4, 128, 64
    F = synth.make_features(I, D, seed=2)
    F = (F / np.abs(F).max()).astype(np.float32)
    c = torch.as_tensor
    m = ReplicatedUserVBPR(0, 1, c(synth.glorot_uniform(rs, U, k)), c(synth.glorot_uniform(rs, U, d)),
                           c(synth.glorot_uniform(rs, I, k)), c(np.zeros(I, np.float32)), c(F), c(synth.glorot_uniform(rs, D, d)),
                           c(synth.glorot_uniform(rs, D, 1).reshape(-1)), 0.05, 0.0, max_batch=B, user_cap=8, feat_dtype="fp32",
                           device=0)
    u = torch.arange(B, dtype=torch.int32, device="cuda") % U
    i = torch.arange(B, dtype=torch.int32, device="cuda") % I
    m.step(u, i, (i + 1) % I)
    with pytest.raises(_ffi.BprxError) as ei:
        m.eng.sync_check()
    assert ei.value.code == _ffi.E_RANGE


def _worker_cli(rank, world, port, root, dataset, epochs, check_quality):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), BPRX_ONE_GPU="1")
    from fashionvisualexpl_recommend_amd import train_rec
    out = train_rec.train(["--dataset", dataset, "--rec", "vbpr", "--world_size", str(world), "--shard", "item",
                           "--dist_backend", "gloo", "--batch_size", "128", "--epochs", str(epochs), "--embed_k", "16", "--embed_d", "8",
                           "--lr", "0.02", "--top_k", "10", "--optimizer", "adam_tf23", "--dtype", "bf16", "--verbose", "2",
                           "--data_root", root, "--results_root", os.path.join(root, "res")])
    try:
        from fashionvisualexpl_recommend_amd import train_rec as tr
        from fashionvisualexpl_recommend_amd.evaluator import _eval_block
        m = tr._last_model
        # every replicated table BIT-identical across the ranks
        for n in ("Gu", "Tu", "E", "Bp"):
            mine = m.engine.t[n].cpu()
            parts = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            assert all(torch.equal(parts[0], x) for x in parts), n
        # the shard-additive device metrics == the reference's definitions on the gathered score rows (host evaluator)
        got = m.metrics(10)
        sc = m.predict_block(0, m.num_users)
        full = m.full_state()
        if rank == 0:
            for key, lst in (("_t", m.data.test_list), ("_v", m.data.validation_list)):
                rows = np.array(_eval_block(sc, 0, m.data.training_list, lst, 10))
                for q, name in enumerate(("hr", "p", "r", "auc", "ndcg")):
                    assert got[name + key] == pytest.approx(rows[:, q].mean(), abs=1e-12), name + key
            res = out[0]
            assert sorted(res.keys()) == list(range(1, epochs + 1))
            assert res[epochs]["hr_t"] == pytest.approx(got["hr_t"], abs=1e-12)
            assert res[epochs]["auc_t"] == res[epochs]["auc_v"]        # the reference's key aliasing (Evaluator.py:220)
            if check_quality:
                assert res[epochs]["hr_t"] > 3 * 10 / 240              # well above a random ranking of the ~240 candidates
                assert res[epochs]["ndcg_t"] > res[1]["ndcg_t"] * 0.9
            # the reference's output files, written by rank 0 (BPRMF.py:158-183)
            dp = m.directory_parameters
            rdir, wdir = os.path.join(root, "res", "rec_results", dataset, "vbpr"), os.path.join(root, "res", "rec_model_weights", dataset, "vbpr")
            files = os.listdir(rdir) + os.listdir(wdir)
            assert f"recs-{epochs}-{dp}.tsv" in files and f"results-metrics-{dp}.pkl" in files, files
            assert any(f.startswith("best-recs-") for f in files) and any(f.startswith("best-weights-") for f in files), files
            assert f"weights-1-{dp}.pt" in files and f"weights-2-{dp}.pt" in files, files
            import pickle
            assert pickle.load(open(os.path.join(rdir, f"results-metrics-{dp}.pkl"), "rb")) == res
            lines = open(os.path.join(rdir, f"recs-{epochs}-{dp}.tsv")).read().splitlines()
            assert len(lines) == m.num_users * 10 and lines[0].split("\t")[0] == "0"
            w = torch.load(os.path.join(wdir, f"weights-2-{dp}.pt"), weights_only=True)
            assert w["Gi"].shape == (m.num_items, 16) and w["Gu"].shape == (m.num_users, 16) and full["Gi"].shape == w["Gi"].shape
        return_shard = m.engine.t["Gi"].cpu().numpy()
        if dataset == "half" and rank == 1:
            # this rank's shard holds no training positive: empty batches, yet every collective was joined; its item rows
            # received no gradient (Adam with m = v = 0 leaves a row where it is)
            rs = np.random.RandomState(0)
            from fashionvisualexpl_recommend_amd.synth import glorot_uniform
            glorot_uniform(rs, m.num_users, 16)
            Gi0 = glorot_uniform(rs, m.num_items, 16)[m.lo:m.hi]
            assert m.local_pos == 0 and m.sampler is None
            np.testing.assert_array_equal(return_shard, Gi0)
    finally:
        dist.destroy_process_group()


def test_train_rec_cli_item_sharded_two_ranks(tmp_path):
    """train_rec.py --world_size 2 --shard item: sharded feature ingestion, GPU-local negatives, the reference's optimizer
    through the replicated-user step, device-side shard-additive evaluation, the reference's output files; two ranks on one
    GPU (gloo)."""
    from fashionvisualexpl_recommend_amd import synth
    tr, va, te = synth.make_interactions_clustered(300, 240, per_user=22, clusters=12, seed=5)
    F = synth.make_features(240, 128, seed=5)
    cl = np.random.RandomState(5).randint(12, size=240)           # (same seed as the generator's item clusters)
    F += 0.5 * np.eye(12, 128, dtype=np.float32)[cl] * 3          # features that carry the cluster: VBPR can use them
    synth.write_dataset(str(tmp_path), "shd", tr, va, te, 240, features=F.astype(np.float64))
    mp.spawn(_worker_cli, args=(2, _free_port(), str(tmp_path), "shd", 6, True), nprocs=2, join=True)


def test_train_rec_cli_rank_without_local_positives(tmp_path):
    """One item shard holds no training positive: that rank steps with EMPTY batches (count-0 message, zero dense
    gradient) and still joins every collective -- no deadlock, replicas bit-identical, its item rows untouched."""
    from fashionvisualexpl_recommend_amd import synth
    rs = np.random.RandomState(9)
    U, I = 120, 240
    tr = [sorted(rs.choice(I // 2, size=12, replace=False).tolist()) for _ in range(U)]     # positives in the lower half only
    va = [[int(I // 2 + rs.randint(I // 2))] for _ in range(U)]                              # held-out items in the upper half
    te = [[int(rs.choice([x for x in range(I // 2) if x not in tr[u]]))] for u in range(U)]
    F = synth.make_features(I, 128, seed=9)
    synth.write_dataset(str(tmp_path), "half", tr, va, te, I, features=F.astype(np.float64))
    mp.spawn(_worker_cli, args=(2, _free_port(), str(tmp_path), "half", 2, False), nprocs=2, join=True)


def _worker_cli_user(rank, world, port, root, opt="sgd", lr="0.15"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), BPRX_ONE_GPU="1")
    from fashionvisualexpl_recommend_amd import train_rec
    epochs = 8
    out = train_rec.train(["--dataset", "ush", "--rec", "bprmf", "--world_size", str(world), "--shard", "user",
                           "--dist_backend", "gloo", "--batch_size", "128", "--epochs", str(epochs), "--embed_k", "16",
                           "--lr", lr, "--top_k", "10", "--optimizer", opt, "--verbose", "2",
                           "--data_root", root, "--results_root", os.path.join(root, "res")])
    try:
        from argparse import Namespace
        from fashionvisualexpl_recommend_amd import train_rec as tr
        m = tr._last_model
        got = m.metrics(10)
        full = m.full_state()
        if rank == 0:
            # the same tables in a single-GPU model: its device evaluator must give the same means (same score kernel)
            from fashionvisualexpl_recommend_amd.models import BPRMF
            params = Namespace(dataset="ush", validation=True, batch_size=128, epochs=1, batch_eval=128, embed_k=16, lr=0.05, reg=0.0,
                               top_k=10, verbose=-1, restore_epochs=1, rec="bprmf", best_metric="ndcg", optimizer="sgd", init_seed=0)
            single = BPRMF(m.data, params, init={n: full[n].numpy() for n in ("Gu", "Gi", "Bi")})
            want = single.evaluator.metrics()
            for key in want:
                assert got[key] == pytest.approx(want[key], abs=1e-12), key
            res = out[0]
            assert sorted(res.keys()) == list(range(1, epochs + 1))
            assert res[epochs]["hr_t"] == pytest.approx(got["hr_t"], abs=1e-12)
            print("AUC/HR per epoch", [(round(res[e]["auc_v"], 4), round(res[e]["hr_t"], 4)) for e in sorted(res)])
            assert res[epochs]["auc_v"] > res[1]["auc_v"] + 0.01 and res[epochs]["hr_t"] > 2.5 * 10 / 240      # it learns
            dp = m.directory_parameters
            rdir, wdir = os.path.join(root, "res", "rec_results", "ush", "bprmf"), os.path.join(root, "res", "rec_model_weights", "ush", "bprmf")
            files = os.listdir(rdir) + os.listdir(wdir)
            assert f"recs-{epochs}-{dp}.tsv" in files and f"results-metrics-{dp}.pkl" in files, files
            assert any(f.startswith("best-recs-") for f in files) and any(f.startswith("best-weights-") for f in files), files
            lines = open(os.path.join(rdir, f"recs-{epochs}-{dp}.tsv")).read().splitlines()
            assert len(lines) == m.num_users * 10 and [l.split("\t")[0] for l in lines[::10]] == [str(u) for u in range(m.num_users)]
            # ... and the single-GPU evaluator writes the same file from the same tables
            p1 = os.path.join(root, "single.tsv")
            single.evaluator.store_recommendation(p1)
            assert open(p1).read() == open(os.path.join(rdir, f"recs-{epochs}-{dp}.tsv")).read()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("opt,lr", [("sgd", "0.15"), ("adam_tf23", "0.02")])
def test_train_rec_cli_user_sharded_two_ranks(tmp_path, opt, lr):
    """train_rec.py --world_size 2 --shard user --rec bprmf: user rows stay on their rank, item rows travel by the fixed-capacity
    all-to-alls (routing in HIP kernels), every rank evaluates its own users on the device; the reference's outputs from rank 0.
    adam_tf23 is the CLI's default optimizer (the reference's): the item owners step their whole shard (bprx_adam_rows)."""
    from fashionvisualexpl_recommend_amd import synth
    tr, va, te = synth.make_interactions_clustered(301, 240, per_user=22, clusters=12, seed=5)      # 301: unequal user shards
    synth.write_dataset(str(tmp_path), "ush", tr, va, te, 240)
    mp.spawn(_worker_cli_user, args=(2, _free_port(), str(tmp_path), opt, lr), nprocs=2, join=True)


def _worker_nosync(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from fashionvisualexpl_recommend_amd import synth
        from fashionvisualexpl_recommend_amd.dist import ItemShardedVBPR, UserShardedBPRMF
        rs = np.random.RandomState(3)
        U, I, k, d, D, B = 300, 256, 16, 8, 128, 200
        c = lambda a: torch.as_tensor(a.copy())
        F = synth.make_features(I, D, seed=3)
        F = (F / np.abs(F).max()).astype(np.float32)
        mf = UserShardedBPRMF(0, 1, I, c(synth.glorot_uniform(rs, U, k)), c(synth.glorot_uniform(rs, I, k)),
                              c(np.zeros(I, np.float32)), 0.05, 1e-3, max_batch=B, device=0)
        vb = ItemShardedVBPR(0, 1, U, c(synth.glorot_uniform(rs, U, k)), c(synth.glorot_uniform(rs, U, d)),
                             c(synth.glorot_uniform(rs, I, k)), c(np.zeros(I, np.float32)), c(F), c(synth.glorot_uniform(rs, D, d)),
                             c(synth.glorot_uniform(rs, D, 1).reshape(-1)), 0.05, 1e-3, max_batch=B, feat_dtype="fp32", device=0)
        from fashionvisualexpl_recommend_amd.dist import ReplicatedUserVBPR
        reps = [ReplicatedUserVBPR(0, 1, c(synth.glorot_uniform(rs, U, k)), c(synth.glorot_uniform(rs, U, d)),
                                   c(synth.glorot_uniform(rs, I, k)), c(np.zeros(I, np.float32)), c(F), c(synth.glorot_uniform(rs, D, d)),
                                   c(synth.glorot_uniform(rs, D, 1).reshape(-1)), 0.05, 1e-3, max_batch=B, user_cap=U, feat_dtype="fp32",
                                   device=0, optimizer=opt, dense_reduce=dr, overlap=ov)
                for opt, dr, ov in (("sgd", "gather", True), ("adam_tf23", "allreduce", True), ("sgd", "gather", False))]
        dev = lambda a: torch.as_tensor(a.astype(np.int32), device="cuda")
        bt = [(dev(rs.randint(U, size=B)), dev(rs.randint(I, size=B)), dev(rs.randint(I, size=B))) for _ in range(4)]
        for m in reps:                                       # the default multi-GPU step (bench.py --gpus N)
            m.step(*bt[0]); m.step(*bt[1])
            torch.cuda.synchronize()
            torch.cuda.set_sync_debug_mode("error")
            try:
                m.step(*bt[2]); m.step(*bt[3])
            finally:
                torch.cuda.set_sync_debug_mode("default")
            torch.cuda.synchronize()
            m.eng.sync_check()
        for m in (mf, vb):
            m.step(*bt[0]); m.step(*bt[1])                     # lazy allocations, first-use paths
            torch.cuda.synchronize()
            torch.cuda.set_sync_debug_mode("error")           # any synchronising call inside the step now raises
            try:
                m.step(*bt[2]); m.step(*bt[3])
            finally:
                torch.cuda.set_sync_debug_mode("default")
            torch.cuda.synchronize()
            assert not m.x.overflowed()
            m.eng.sync_check()
    finally:
        dist.destroy_process_group()


def test_fixed_capacity_steps_enqueue_without_host_synchronisation():
    """The replicated-user step (the default of bench.py --gpus N; overlapped and single-message order, both dense forms, both
    optimizers) and the all-to-all modes with fixed-capacity splits: torch's sync-debug mode ("error") around whole steps --
    no split size, mask count or index is read back to the host (one rank over RCCL: the collectives are real, the routing is
    the same)."""
    mp.spawn(_worker_nosync, args=(1, _free_port()), nprocs=1, join=True)
