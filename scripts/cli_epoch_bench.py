"""End-to-end epoch time of the CLI surface (train_rec.train) on a synthetic dataset written in the reference's file formats:
what a user of the reference sees after switching (defaults: batch 256, adam_tf23, fp32 features, the reference's index stream).
   python scripts/cli_epoch_bench.py [U I D epochs]"""
import os, sys, tempfile, time, re, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fashionvisualexpl_recommend_amd import synth, train_rec

U, I, D, EP = [int(x) for x in (sys.argv[1:5] + ["20000", "10000", "4096", "2"][len(sys.argv) - 1:])]
root = tempfile.mkdtemp(prefix="bprx_cli_")
tr, va, te = synth.make_interactions(U, I, per_user=22, seed=1)
F = synth.make_features(I, D, seed=1)
synth.write_dataset(root, "cli", tr, va, te, I, features=F)
npos = sum(len(l) for l in tr)
for extra in ([], ["--optimizer", "sgd", "--dtype", "bf16"], ["--batch_size", "8192", "--optimizer", "sgd", "--dtype", "bf16", "--sampler", "philox"]):
    for rec in ("vbpr", "bprmf"):
        if rec == "bprmf" and "--dtype" in extra:
            extra = [x for x in extra if x not in ("--dtype", "bf16")]
        buf = io.StringIO()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(buf):
            train_rec.train(["--dataset", "cli", "--rec", rec, "--epochs", str(EP), "--top_k", "10", "--data_root", root,
                             "--results_root", os.path.join(root, "res")] + extra)
        wall = time.perf_counter() - t0
        tt = re.findall(r"Train Time: (\d+):(\d+):([\d.]+)", buf.getvalue())
        et = re.findall(r"Evaluation Time: (\d+):(\d+):([\d.]+)", buf.getvalue())
        sec = lambda m: [int(h) * 3600 + int(mi) * 60 + float(s) for h, mi, s in m]
        print("%-6s %-60s train/epoch %s s  eval/epoch %s s  wall %.1f s  (%d positives/epoch -> %.2e triplets/s)" % (
            rec, " ".join(extra) or "(reference defaults)", ["%.3f" % x for x in sec(tt)], ["%.3f" % x for x in sec(et)], wall, npos,
            npos / min(sec(tt)) if tt else 0), flush=True)
