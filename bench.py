#!/usr/bin/env python3
"""bench.py -- BPR triplets/s of the VBPR train step on MI355X (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
(for N > 1 the driver launches it under torch.distributed.run, one rank per GPU over RCCL.)

A "step" is one batch-synchronous VBPR train step (bprx_step: projection of every item row, per-triplet
forward/backward, dense E|Bp gradient, sparse + dense SGD) over one batch of B synthetic triplets whose index
tensors, factor tables and bf16 feature table are resident in HBM before the timed region starts.
Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel, HIP events on the launch stream),
`kernels` (per-kernel average ms), `step_roofline` (SURVEY 8(d) per-triplet accounting), `cpu_baseline`
(the CPU oracle timed on a bounded, ratio-preserving sample; rank 0, N == 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)

WORKLOADS = {
    # BASELINE.json configs[1]: VBPR k=64, 4096-d CNN feats, 100K users x 50K items, bf16, 1 x MI355X
    "c2": dict(model="vbpr", U=100_000, I=50_000, k=64, d=64, D=4096, dtype="bf16", B=65_536),
    # BASELINE.json configs[2] per-GPU shard shape (BPRMF k=128, 5M x 1M over 8 GPUs -> 625K users/GPU)
    "c3shard": dict(model="bprmf", U=625_000, I=1_000_000, k=128, d=0, D=0, dtype="fp32", B=65_536),
    # BASELINE.json configs[3] per-GPU shard shape (VBPR k=128, 2M x 500K over 8 GPUs -> 250K users, 62.5K items/GPU)
    "c4shard": dict(model="vbpr", U=250_000, I=62_500, k=128, d=128, D=4096, dtype="bf16", B=65_536),
    # BASELINE.json configs[4]: VBPR k = d = 256, fp8 (e4m3fn) feature table resident in HBM; and its bf16 twin
    # (HBM scale, SURVEY 8 sizing "I = 500 K -> 2 GB": the table is 8x the 256-MiB Infinity Cache; B = 2^18 so that 2B >= I and
    #  the step streams the whole table like C2 does -- at B = 65 536 the library's per-step policy takes the touched-item list)
    "c5": dict(model="vbpr", U=1_000_000, I=500_000, k=256, d=256, D=4096, dtype="fp8", B=262_144),
    "c5list": dict(model="vbpr", U=1_000_000, I=500_000, k=256, d=256, D=4096, dtype="fp8", B=65_536),
    "c5small": dict(model="vbpr", U=100_000, I=50_000, k=256, d=256, D=4096, dtype="fp8", B=65_536),   # fits the Infinity Cache
    "c5bf16": dict(model="vbpr", U=100_000, I=50_000, k=256, d=256, D=4096, dtype="bf16", B=65_536),
    # configs[1] shape with fp8 features (what the fp8 table buys on the headline shape)
    "c2fp8": dict(model="vbpr", U=100_000, I=50_000, k=64, d=64, D=4096, dtype="fp8", B=65_536),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (exactly --steps steps between barrier + synchronize) is run this many times; "
                         "value / ms_per_step are the MEDIAN repeat, every repeat is listed in repeats_ms")
    ap.add_argument("--min-timed-seconds", type=float, default=2.0,
                    help="repeat the timed region until the timed GPU work adds up to at least this (see --repeats)")
    ap.add_argument("--max-repeats", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the workload's batch size")
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "adam_tf23"])
    ap.add_argument("--sampler", default="epoch", choices=["philox", "epoch", "pregen"],
                    help="philox: device sampler, i.i.d. uniform interactions, inside the timed step; epoch: device "
                         "epoch-walk sampler (the reference's visiting order: user-grouped batches); pregen: resident "
                         "pre-generated index batches")
    ap.add_argument("--pos-per-user", type=int, default=20)
    ap.add_argument("--zipf", type=float, default=0.0,
                    help="> 0: positives follow a Zipf(s) item popularity instead of the uniform one (hot items)")
    ap.add_argument("--zipf-ids", default="shuffled", choices=["shuffled", "ranked"],
                    help="--zipf: item id of popularity rank r = a fixed random permutation of r (default) or r itself")
    ap.add_argument("--dist-mode", default="replicated", choices=["replicated", "a2a"],
                    help="N > 1, VBPR: replicated = user tables on every rank, one all-gather per step (default); "
                         "a2a = user tables range-partitioned, rows fetched / gradients returned by all-to-all")
    ap.add_argument("--dense-reduce", default="gather", choices=["gather", "allreduce"],
                    help="N > 1, replicated mode: the ranks' dE|dBp all-gathered and summed in rank order (default: "
                         "bit-identical replicas whatever the collective's reduction order) or summed by an RCCL all-reduce "
                         "(north_star's form: 2(N-1)/N instead of N-1 dense-gradient transfers per rank)")
    ap.add_argument("--no-dist-overlap", action="store_true",
                    help="N > 1, replicated mode: the round-1 order (ONE message [user rows | dE|dBp] after the whole local "
                         "step) instead of the user rows' all-gather travelling beside the backward projection")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sampler-overlap", action="store_true",
                    help="draw each batch one step ahead on a side stream (measured on C2: 0.287 vs 0.280 ms/step without "
                         "-- the sampler then competes with the step's kernels; off by default)")
    ap.add_argument("--cpu-sample-steps", type=int, default=60)
    ap.add_argument("--cpu-eager-seconds", type=float, default=10.0,
                    help="wall-clock budget of the like-for-like eager CPU baseline (BASELINE.md B1); 0 = skip")
    return ap.parse_args()


def algorithmic_bytes_per_triplet(w, B):
    """fused single pass, every touched row read once and (if trainable) written once, int32 indices."""
    k, d, D = w["k"], w["d"], w["D"]
    s = {"fp32": 4, "bf16": 2, "fp8": 1}[w["dtype"]]
    b = 24 * k + 28
    if w["model"] == "vbpr":
        b += 8 * d + 2 * D * s + 2.0 * (D * (d + 1)) * 4 / B
    return b


def make_state(w, device, seed, torch):
    """Synthetic tables generated directly in HBM (no dataset/checkpoint exists offline): Glorot-uniform factors,
    |N(0,1)| half-sparse features divided by their global max (visual_loader_mixin.py:30) and stored as bf16."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def glorot(r, c):
        lim = (6.0 / (r + c)) ** 0.5
        return (torch.rand((r, c), generator=g, device=device, dtype=torch.float32) * 2 - 1) * lim
    t = dict(Gu=glorot(w["U"], w["k"]), Gi=glorot(w["I"], w["k"]), Bi=torch.zeros(w["I"], device=device))
    if w["model"] == "vbpr":
        I, D, d = w["I"], w["D"], w["d"]
        fdt = {"bf16": torch.bfloat16, "fp8": torch.float8_e4m3fn}.get(w["dtype"], torch.float32)
        F = torch.empty((I, D), device=device, dtype=torch.float32 if fdt == torch.float8_e4m3fn else fdt)
        mx = 0.0
        chunk = 8192
        for s in range(0, I, chunk):                      # chunked: never more than ~400 MB of fp32 temporaries
            n = min(chunk, I - s)
            f = torch.randn((n, D), generator=g, device=device).abs_()
            f *= (torch.rand((n, D), generator=g, device=device) < 0.5)
            mx = max(mx, float(f.max()))
            F[s:s + n] = f.to(F.dtype)
        F.div_(mx)
        if fdt == torch.float8_e4m3fn:                    # e4m3fn codes of f * 448 (Engine's feat_scale default)
            F8 = torch.empty((I, D), device=device, dtype=fdt)
            for s in range(0, I, chunk):
                F8[s:s + chunk] = (F[s:s + chunk] * 448.0).to(fdt)
            F = F8
        t.update(Tu=glorot(w["U"], d), F=F, E=glorot(D, d), Bp=glorot(D, 1).reshape(-1))
    return t


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from fashionvisualexpl_recommend_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d (WORLD_SIZE=%d)"
                         % (args.gpus, args.gpus, world))
    # Rehearsal hook for a one-GPU box: BPRX_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses gloo (host-staged
    # collectives) so that the N > 1 control flow can be exercised without N GPUs.  Never set by the driver.
    rehearse = os.environ.get("BPRX_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # BPRX_BENCH_FORCE_SHARDED=1 (never set by the driver): run the N > 1 code path with a single rank over RCCL, to
    # measure what the sharded step costs besides the wire time (routing, host sync, staging copies)
    force_sharded = world == 1 and os.environ.get("BPRX_BENCH_FORCE_SHARDED") == "1"
    if force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=device)
    if world > 1:
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)  # "nccl" IS RCCL on ROCm

    w = dict(WORKLOADS[args.workload])
    if args.batch:
        w["B"] = args.batch
    B, K, W = w["B"], args.steps, args.warmup

    # Weak scaling: every rank holds one item shard + one user shard of this shape.  N > 1 (VBPR): item-sharded step
    # of fashionvisualexpl_recommend_amd/dist.py -- users of a rank's batch are GLOBAL (any shard): their rows are
    # fetched / their gradients returned by all-to-all, E|Bp gradients are all-reduced (RCCL), negatives stay local.
    tables = make_state(w, device, 1234 + rank, torch)
    sharded = None
    users_total = w["U"] * world
    # index ranges the sampler draws from: item-sharded VBPR -> global users, local items;
    # user-sharded BPRMF -> local users, global items
    samp_users = users_total if w["model"] == "vbpr" else w["U"]
    samp_items = w["I"] if w["model"] == "vbpr" else w["I"] * world
    if (world > 1 or force_sharded) and w["model"] == "vbpr":
        for n in ("E", "Bp"):
            if rehearse:
                hcopy = tables[n].cpu()
                dist.broadcast(hcopy, src=0)
                tables[n].copy_(hcopy)
            else:
                dist.broadcast(tables[n], src=0)
        if args.dist_mode == "replicated":
            # users replicated on every rank (identical initial values: one generator seed for all ranks), items / F
            # sharded; ONE fixed-size all-gather per step (dist.ReplicatedUserVBPR)
            from fashionvisualexpl_recommend_amd.dist import ReplicatedUserVBPR
            gg = torch.Generator(device=device)
            gg.manual_seed(4242)
            glo = lambda r, c: (torch.rand((r, c), generator=gg, device=device) * 2 - 1) * (6.0 / (r + c)) ** 0.5
            Gu_all, Tu_all = glo(users_total, w["k"]), glo(users_total, w["d"])
            cap = B if args.sampler != "epoch" else B // args.pos_per_user + 256
            sharded = ReplicatedUserVBPR(rank, world, Gu_all, Tu_all, tables["Gi"], tables["Bi"], tables["F"], tables["E"],
                                         tables["Bp"], lr=1e-4, reg=1e-4, max_batch=B, user_cap=cap, feat_dtype=w["dtype"],
                                         device=local_rank, optimizer=args.optimizer, dense_reduce=args.dense_reduce,
                                         overlap=not args.no_dist_overlap)
        else:
            from fashionvisualexpl_recommend_amd.dist import ItemShardedVBPR
            sharded = ItemShardedVBPR(rank, world, users_total, tables["Gu"], tables["Tu"], tables["Gi"], tables["Bi"],
                                      tables["F"], tables["E"], tables["Bp"], lr=1e-4, reg=1e-4, max_batch=B,
                                      feat_dtype=w["dtype"], device=local_rank, optimizer=args.optimizer)
        eng = sharded.eng
    elif world > 1 or force_sharded:
        # user-sharded BPRMF (configs[2]): user rows stay local, item rows are exchanged by all-to-all, no all-reduce
        from fashionvisualexpl_recommend_amd.dist import UserShardedBPRMF
        sharded = UserShardedBPRMF(rank, world, w["I"] * world, tables["Gu"], tables["Gi"], tables["Bi"], lr=1e-4, reg=1e-4,
                                   max_batch=B, device=local_rank, optimizer=args.optimizer)
        eng = sharded.eng
    else:
        eng = Engine(model=w["model"], num_users=w["U"], num_items=w["I"], embed_k=w["k"], embed_d=w["d"],
                     feat_dim=w["D"], feat_dtype=w["dtype"], optimizer=args.optimizer, lr=1e-4, reg=1e-4, max_batch=B,
                     device=local_rank).bind(**tables)

    gi = torch.Generator(device=device)
    gi.manual_seed(99 + rank)
    if args.sampler in ("philox", "epoch"):
        # synthetic training interactions resident in HBM: pos-per-user uniform items per user, CSR sorted per user
        from fashionvisualexpl_recommend_amd.engine import EpochWalkSampler, PhiloxSampler
        npu = args.pos_per_user
        # (N > 1: every GLOBAL user has pos-per-user positives inside this rank's item shard)
        if args.zipf > 0:
            # Zipf(s) item popularity (SURVEY 8(d) "throughput realism" variant): item ranks drawn by inverse-CDF from
            # weights 1/rank^s; duplicates inside a user's list are allowed (they only make that positive likelier)
            wts = 1.0 / torch.arange(1, samp_items + 1, device=device, dtype=torch.float64) ** args.zipf
            cdf = torch.cumsum(wts / wts.sum(), 0)
            r = torch.rand((samp_users, npu), generator=gi, device=device, dtype=torch.float64)
            ranks = torch.searchsorted(cdf, r).clamp_(max=samp_items - 1)
            if args.zipf_ids == "shuffled":
                # popularity rank -> item id through a fixed random permutation: in a real catalogue the hot items are not the
                # ids 0, 1, 2, ... (with --zipf-ids ranked the 16 hottest items share ONE 64-byte line of every per-item
                # counter array, and the memory-side atomics on it serialise: the adversarial layout)
                gp = torch.Generator(device=device)
                gp.manual_seed(777)
                ranks = torch.randperm(samp_items, generator=gp, device=device)[ranks]
            items = ranks.to(torch.int32).sort(dim=1).values
        else:
            items = torch.randint(samp_items, (samp_users, npu), generator=gi, device=device, dtype=torch.int32).sort(dim=1).values
        indptr = torch.arange(samp_users + 1, device=device, dtype=torch.int64) * npu
        pos_user = torch.arange(samp_users, device=device, dtype=torch.int32).repeat_interleave(npu)
        cls = EpochWalkSampler if args.sampler == "epoch" else PhiloxSampler
        sampler = cls.from_csr(indptr, items.reshape(-1), pos_user, samp_items, seed=2024 + rank)
        bufs = tuple(torch.empty(B, dtype=torch.int32, device=device) for _ in range(3))
        batches, nb = None, 0
    else:
        nb = min(K + W, 16)                               # distinct resident index batches, cycled
        batches = [(torch.randint(samp_users, (B,), generator=gi, device=device, dtype=torch.int32),
                    torch.randint(samp_items, (B,), generator=gi, device=device, dtype=torch.int32),
                    torch.randint(samp_items, (B,), generator=gi, device=device, dtype=torch.int32)) for _ in range(nb)]
    # The triplet stream does not depend on the parameters, so the batch of step s+1 CAN be drawn on a side stream while
    # step s runs (--sampler-overlap: double-buffered index arrays, still one batch per step inside the timed region).
    pipe = batches is None and args.sampler_overlap
    if batches is None and not pipe and (sharded is None or type(sharded).__name__ == "ReplicatedUserVBPR"):
        sampler.feeds(eng)                                # the sampler also leaves the byte planes the step's index pass scans
                                                          # (the replicated-user step hands the same index arrays to its engine)
    if pipe:
        side = torch.cuda.Stream(device=device)
        bufs2 = (bufs, tuple(torch.empty(B, dtype=torch.int32, device=device) for _ in range(3)))
        ready = [torch.cuda.Event(), torch.cuda.Event()]     # buffer b holds a fresh batch
        done = [torch.cuda.Event(), torch.cuda.Event()]      # the step that read buffer b has finished
        state = {"n": 0}

        def prefetch(slot):
            with torch.cuda.stream(side):
                side.wait_event(done[slot])                   # (recorded below; a never-recorded event does not block)
                sampler.sample(B, out=bufs2[slot])
                ready[slot].record(side)

        prefetch(0)

    def one_step(s):
        if pipe:
            slot = state["n"] & 1
            state["n"] += 1
            prefetch(slot ^ 1)                                # next batch, concurrently with this step
            torch.cuda.current_stream().wait_event(ready[slot])
            u, i, j = bufs2[slot]
        else:
            u, i, j = sampler.sample(B, out=bufs) if batches is None else batches[s % nb]
        if sharded is None:
            eng.step(u, i, j, want_loss=False)
        else:
            sharded.step(u, i, j)                         # all-to-all user rows, all-reduce E|Bp grads (RCCL over xGMI)
        if pipe:
            done[slot].record(torch.cuda.current_stream())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the loop runs on a non-default stream (BPRX_GRAPH=1 lets libbprx replay the sgd step as one hipGraph there; the
    # legacy default stream cannot be captured -- measured slower than plain launches, so it is off by default)
    run_stream = torch.cuda.Stream(device=device)
    run_stream.wait_stream(torch.cuda.current_stream())
    repeats = []
    with torch.cuda.stream(run_stream):
        for s in range(W):
            one_step(s)
        # Every timed region is EXACTLY K steps between barrier + synchronize.  A region of the driver's K = 20 is ~5 ms of
        # GPU work, invisible to an outside utilisation sampler, so the region is repeated until the timed GPU work adds up to
        # --min-timed-seconds (default 2 s): an untimed pilot region sizes the repeat count (the same on every rank: MAX over
        # ranks), the median region is the one reported, every region is listed in repeats_ms.
        barrier()
        t0 = time.perf_counter()
        for s in range(K):
            one_step(W + s)
        barrier()
        pilot = time.perf_counter() - t0
        if world > 1:
            pt = torch.tensor([pilot], device="cpu" if rehearse else device, dtype=torch.float64)
            dist.all_reduce(pt, op=dist.ReduceOp.MAX)
            pilot = float(pt.item())
        n_rep = max(1, args.repeats, int(np.ceil(args.min_timed_seconds / max(pilot, 1e-6))))
        n_rep = min(n_rep, args.max_repeats)
        for rep in range(n_rep):
            barrier()
            t0 = time.perf_counter()
            for s in range(K):                                # EXACTLY K steps between barrier + synchronize
                one_step(W + (rep + 1) * K + s)
            barrier()
            repeats.append(time.perf_counter() - t0)
    torch.cuda.current_stream().wait_stream(run_stream)
    if world > 1:
        tmax = torch.tensor(repeats, device="cpu" if rehearse else device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)           # MAX over ranks, per repeat
        repeats = [float(x) for x in tmax.tolist()]
    elapsed = float(np.median(repeats))                       # the median repeat is the reported one
    eng.sync_check()

    # what this device's HBM delivers to a plain streaming-read kernel (bprx_probe_stream_read), on the launch stream
    measured_peak = measured_peak_nt = None
    if rank == 0:
        try:
            measured_peak = hbm_stream_probe(torch, device, run_stream)
            measured_peak_nt = hbm_stream_probe(torch, device, run_stream, nt=True)
        except Exception as e:                                # never let the probe take the bench line down
            print("hbm probe failed: %r" % (e,), file=sys.stderr)

    # per-kernel durations: a second pass over the same K steps with HIP events around every kernel launch
    eng.profile(True)
    for s in range(K):
        one_step(W + s)
    torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile(False)
    torch.cuda.synchronize()
    ub, ib, jb = sampler.sample(B, out=bufs) if batches is None else batches[0]
    loss = float((eng.step(ub, ib, jb) if sharded is None else sharded.step(ub, ib, jb, want_loss=True)).item())
    assert np.isfinite(loss), loss

    if rank == 0:
        value = world * B * K / elapsed
        per_trip = algorithmic_bytes_per_triplet(w, B)
        kernels = {p: {"avg_ms": ms / n, "launches": n} for p, (ms, n) in prof.items()}
        dom = max(kernels, key=lambda p: kernels[p]["avg_ms"])
        s = {"fp32": 4, "bf16": 2, "fp8": 1}[w["dtype"]]
        if w["model"] == "vbpr":
            PS = 16 * ((w["d"] + 1 + 15) // 16)
            kern_bytes = {   # algorithmic bytes per launch, per kernel (DESIGN.md "Kernels")
                "proj_fwd": w["I"] * (w["D"] * s + PS * 4) + PS * w["D"] * min(s, 2),
                "proj_bwd": w["I"] * (w["D"] * s + PS * 4) + w["D"] * PS * 4,
                "triplet_grad": B * (24 * w["k"] + 28 + 8 * w["d"] + 2 * PS * 4 + 2 * PS * 4) / 1.0,
                "apply": 3 * B * 0 + B * (12 * (w["k"] + w["d"]) + 24 * w["k"]),
            }
            if 2 * B < w["I"]:      # touched-item list mode: the projections move the batch's distinct rows (count known on
                kern_bytes.pop("proj_fwd"), kern_bytes.pop("proj_bwd")     # the device only): no algorithmic figure quoted
        else:
            kern_bytes = {"triplet_grad": B * (24 * w["k"] + 28), "apply": B * 36 * w["k"]}
        # HBM bytes per launch from PMC counters, measured offline with scripts/pmc.sh on this same workload and
        # committed under profiles/ (bench.py itself cannot run under rocprofv3 --pmc); null when not available
        traffic, pmc, pmc_ok = None, {}, False
        list_mode_run = w["model"] == "vbpr" and 2 * B < w["I"]      # libbprx's per-step policy (touched-item list)
        try:
            pmc = json.load(open(os.path.join(REPO, "profiles", "r03_pmc_traffic.json")))
            pmc_ok = args.workload == "c2" and B == WORKLOADS["c2"]["B"] and args.optimizer == "sgd"
            if pmc_ok and dom in pmc:
                traffic = pmc[dom]["hbm_bytes"]
        except Exception:
            traffic = None
        rl = None
        if dom in kern_bytes:
            ach = kern_bytes[dom] / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
            rl = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "bytes_per_launch": kern_bytes[dom],
                  "avg_ms": kernels[dom]["avg_ms"],
                  # the same achieved rate against what a plain streaming-read kernel reaches on THIS device, measured
                  # in this run (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
                  "measured_peak": measured_peak, "frac_of_measured": (ach / measured_peak) if measured_peak else None,
                  # the same streaming-read kernel with `nt` loads (the policy of the bf16 feature passes): the higher ceiling
                  "measured_peak_nt": measured_peak_nt,
                  "frac_of_measured_nt": (ach / measured_peak_nt) if measured_peak_nt else None}
            # MFMA utilisation of the two projections (north_star: "MFMA utilisation on the projection against gfx950
            # peak"): flops EXECUTED per launch (2*I*D*PS each, padded columns included) / HIP-event duration / dense peak
            # (2.5 PF bf16; fp8 features: the forward uses the fp8 MFMA -> 5 PF, the backward widens F to bf16 -> 2.5 PF);
            # busy_pmc = SQ_VALU_MFMA_BUSY_CYCLES share measured offline by scripts/pmc.sh (profiles/), when available
            if w["model"] == "vbpr" and w["dtype"] != "fp32" and not list_mode_run:
                fl = 2.0 * w["I"] * w["D"] * PS
                mu = {}
                for ph, peak in (("proj_fwd", 5000.0 if w["dtype"] == "fp8" else 2500.0), ("proj_bwd", 2500.0)):
                    if ph in kernels:
                        tf = fl / (kernels[ph]["avg_ms"] * 1e-3) / 1e12
                        mu[ph] = {"flops_per_launch": fl, "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                                  "busy_pmc": (pmc.get(ph, {}) or {}).get("mfma_busy_frac") if pmc_ok else None}
                rl["mfma_util"] = mu
        out = {
            "metric": "BPR triplets/sec", "value": value, "unit": "triplets/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": elapsed / K * 1e3, "repeats": len(repeats), "timed_region_s": float(sum(repeats)),
            "repeats_ms": [round(r / K * 1e3, 5) for r in repeats],
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": w["dtype"] if w["model"] == "vbpr" else "fp32", "data": "synthetic",
            "config": {"workload": "%s: %s k=%d d=%d D=%d, %d users x %d items per GPU, %s features, B=%d per GPU, %s"
                                   % (args.workload, w["model"].upper(), w["k"], w["d"], w["D"], w["U"], w["I"],
                                      w["dtype"], B, args.optimizer),
                       "global_batch": B * world, "parallelism": "single" if (world == 1 and sharded is None) else
                       ((("item-shard x%d, users replicated: all-gather of the distinct users' gradient rows %s; dE|dBp %s; "
                          "local negatives" % (world, "after the local step" if args.no_dist_overlap else
                                               "beside the backward projection",
                                               "all-gathered and summed in rank order" if args.dense_reduce == "gather" else
                                               "by RCCL all-reduce")) if args.dist_mode == "replicated" else
                         ("item-shard x%d: all-to-all user rows + all-reduce(E|Bp), local negatives" % world))
                        if w["model"] == "vbpr" else ("user-shard x%d: all-to-all item rows, no all-reduce" % world)),
                       "sampler": (("device philox, uniform positive + rejection negative" if args.sampler == "philox" else
                                    "device epoch walk (every positive once per epoch, user-grouped) + philox negative")
                                   + ", one batch per step inside the timed region%s (%d positives/user)"
                                   % (", drawn one step ahead on a side stream" if pipe else
                                      "",
                                      args.pos_per_user)
                                   + (", Zipf(%.2f) item popularity, ids %s" % (args.zipf, args.zipf_ids) if args.zipf > 0 else ""))
                       if batches is None else "pre-generated uniform (u,i,j), resident"},
            "roofline": rl, "kernels": kernels,
            "step_roofline": {"bytes_per_triplet": per_trip, "achieved": value * per_trip / 1e9 / world,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": value * per_trip / 1e9 / world / HBM_PEAK_GBS},
            "loss_after": loss,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, tables, args.cpu_sample_steps, args.optimizer)
            if args.cpu_eager_seconds > 0:
                out["cpu_baseline_eager"] = cpu_baseline_eager(w, tables, args.cpu_eager_seconds)
        print(json.dumps(out))
    if world > 1 or force_sharded:
        dist.destroy_process_group()


def hbm_stream_probe(torch, device, stream, nt=False):
    """GB/s of bprx_probe_stream_read (nt: bprx_probe_stream_read_nt) over a 1-GiB buffer (far larger than the 256-MiB Infinity Cache), best of 5 timed
    launches after 2 warm-ups, timed with events on the launch stream."""
    import ctypes as C
    from fashionvisualexpl_recommend_amd import _ffi
    L = _ffi.lib()
    nbytes = 1 << 30
    buf = torch.ones(nbytes // 4, dtype=torch.float32, device=device)
    sink = torch.zeros(4096, dtype=torch.float32, device=device)
    best = 0.0
    with torch.cuda.stream(stream):
        sp = C.c_void_p(stream.cuda_stream)
        for it in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            got = (L.bprx_probe_stream_read_nt if nt else L.bprx_probe_stream_read)(C.c_void_p(buf.data_ptr()), nbytes,
                                                                                    C.c_void_p(sink.data_ptr()), sp)
            b.record(stream)
            b.synchronize()
            if got < 0:
                raise RuntimeError("bprx_probe_stream_read failed: %d" % got)
            if it >= 2:
                best = max(best, got / (a.elapsed_time(b) * 1e-3) / 1e9)
    del buf, sink
    return best


def cpu_baseline_eager(w, tables, seconds):
    """the reference's train_step restated op-for-op in torch CPU EAGER mode (gather -> mul/reduce or matmul
    -> softplus -> autograd -> optimizer), batch 256 like the reference's default (train_rec.py:23), full-size fp32 tables on
    the host, Adam over dense gradients (every row of every table moves every step, like TF-2.3's non-lazy sparse apply:
    BPRMF.py:52,123 / VBPR.py:56,142), default intra-op threads.  Mirrors the STRUCTURE of the reference's eager TF loop
    (one kernel per op, a host sync per step for the loss, BPRMF.py:125); it is a timing baseline, not a parity check."""
    import torch
    B = 256
    vb = w["model"] == "vbpr"
    host = lambda n: tables[n].detach().float().cpu()
    params = {n: torch.nn.Parameter(host(n)) for n in (("Gu", "Gi", "Bi", "Tu", "E", "Bp") if vb else ("Gu", "Gi", "Bi"))}
    F = None
    if vb:
        F = host("F")
        if w["dtype"] == "fp8":
            F /= 448.0
    opt = torch.optim.Adam(list(params.values()), lr=1e-3, eps=1e-7)
    g = torch.Generator().manual_seed(11)
    reg = 1e-4

    def call(u, i):                                             # BPRMF.py:55-76 / VBPR.py:59-86
        gu, gi, bi = params["Gu"].index_select(0, u), params["Gi"].index_select(0, i), params["Bi"].index_select(0, i)
        x = bi + (gu * gi).sum(1)
        tu = None
        if vb:
            tu, fi = params["Tu"].index_select(0, u), F.index_select(0, i)
            x = x + (tu * (fi @ params["E"])).sum(1) + fi @ params["Bp"]
        return x, bi, gu, gi, tu

    def step():
        u = torch.randint(w["U"], (B,), generator=g)
        i = torch.randint(w["I"], (B,), generator=g)
        j = torch.randint(w["I"], (B,), generator=g)
        xp, bi, gu, gi, tu = call(u, i)
        xn, bj, _, gj, _ = call(u, j)
        loss = torch.nn.functional.softplus(-torch.clamp(xp - xn, -80.0, 1e8)).sum()
        loss = loss + reg * ((gu ** 2).sum() + (gi ** 2).sum() + (gj ** 2).sum()) + reg * (bi ** 2).sum() + reg / 10 * (bj ** 2).sum()
        if vb:
            loss = loss + reg * (tu ** 2).sum() + reg * ((params["E"] ** 2).sum() + (params["Bp"] ** 2).sum())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return float(loss.detach())                             # loss.numpy(): the reference's per-step host sync

    step()                                                      # warm-up (allocations, Adam slots)
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        if time.perf_counter() - t0 >= seconds and n >= 3:
            break
    dt = time.perf_counter() - t0
    return {"value": n * B / dt, "unit": "triplets/s", "cores": torch.get_num_threads(), "kind": "eager",
            "sample": "%d steps of B=%d (the reference's default batch) in %.1f s on the FULL %d x %d tables, fp32, torch-CPU "
                      "eager + autograd + Adam(dense grads, eps=1e-7), intra-op threads = %d"
                      % (n, B, dt, w["U"], w["I"], torch.get_num_threads())}


def cpu_baseline(w, tables, steps, optimizer):
    """The CPU oracle (oracle/bpr_oracle.c, OpenMP) on a bounded sample that preserves the workload's B:U:I ratios:
    a 1/8 slice of the users and items with B/8 triplets per step (same rows-per-triplet duplication as the
    full batch), same k/d/D and the same bf16 operand rounding."""
    from oracle import oracle as orc
    f = 8
    U, I, B = max(w["U"] // f, 8), max(w["I"] // f, 8), max(w["B"] // f, 8)
    cpu = lambda t, n: t[:n].float().cpu().numpy()
    kw = dict(Gu=cpu(tables["Gu"], U), Gi=cpu(tables["Gi"], I), Bi=cpu(tables["Bi"], I))
    if w["model"] == "vbpr":
        Fh = cpu(tables["F"], I)
        if w["dtype"] == "fp8":
            Fh = Fh / np.float32(448.0)
        kw.update(Tu=cpu(tables["Tu"], U), F=Fh, E=tables["E"].cpu().numpy(),
                  Bp=tables["Bp"].cpu().numpy(), quant={"bf16": 1, "fp8": 2}.get(w["dtype"], 0))
    o = orc.OracleModel(**kw)
    rs = np.random.RandomState(5)
    cores = orc.lib().orc_num_threads()
    o.step(rs.randint(U, size=B), rs.randint(I, size=B), rs.randint(I, size=B), optimizer, 1e-4, 1e-4)   # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        o.step(rs.randint(U, size=B), rs.randint(I, size=B), rs.randint(I, size=B), optimizer, 1e-4, 1e-4)
    dt = time.perf_counter() - t0
    return {"value": steps * B / dt, "unit": "triplets/s", "cores": cores, "kind": "port",
            "sample": "%d steps of B=%d triplets on a 1/%d slice (%d users x %d items) of the same tables, same k/d/D; the "
                      "oracle's projections and F^T W are OpenMP-parallel over %d threads, its per-triplet loop is serial"
                      % (steps, B, f, U, I, cores)}


if __name__ == "__main__":
    main()
