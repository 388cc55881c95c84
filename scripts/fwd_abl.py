#!/usr/bin/env python3
"""Timing-only ablations of the forward projection kernel (BPRX_FWD_ABL, one process per setting)."""
import json, os, subprocess, sys
res = {}
for abl in os.environ.get("ABLS", "0,1,2,3,4,6,7,14,15").split(","):
    env = dict(os.environ, BPRX_FWD_ABL=abl, SWEEP_FWD=os.environ.get("SWEEP_FWD", "10"), SWEEP_BWD="26", SWEEP_SK="32", SWEEP_ROUNDS="2")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "proj_sweep.py")], env=env, capture_output=True, text=True, timeout=300)
    for l in out.stdout.splitlines():
        if l.startswith("{"):
            res[abl] = json.loads(l)["proj_fwd"]
    print(abl, res.get(abl), flush=True)
print(json.dumps(res))
