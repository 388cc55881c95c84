import os, sys, torch, json
sys.path.insert(0, "/root/repo")
import bench
from fashionvisualexpl_recommend_amd.engine import Engine
dev = torch.device("cuda", 0)
for name in ("c5bf16",):
    w = dict(bench.WORKLOADS[name], U=1024)
    t = bench.make_state(w, dev, 1, torch)
    for var in ("12", "8"):
        os.environ["BPRX_FWD_VARIANT"] = var
        e = Engine(model="vbpr", num_users=w["U"], num_items=w["I"], embed_k=w["k"], embed_d=w["d"], feat_dim=w["D"],
                   feat_dtype=w["dtype"], optimizer="sgd", lr=0.05, reg=1e-4, max_batch=1024).bind(**t)
        for _ in range(3): e.step_project()
        torch.cuda.synchronize(); e.profile(True)
        for _ in range(10): e.step_project()
        torch.cuda.synchronize(); p = e.profile_read()
        print(name, "fwd_variant", var, round(p["proj_fwd"][0] / p["proj_fwd"][1] * 1e3, 1), "us", flush=True)
        e.close()
