#!/bin/bash
# A/B of environment settings on one box: scripts/ab.sh "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...   (rocprofv3 kernel averages)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/ab
args=$1; shift
n=0
for envs in "$@"; do
  n=$((n+1)); rm -rf gpurun_out/ab/p$n
  env $envs timeout -k 10 200 python bench.py --steps 100 --repeats 3 --min-timed-seconds 0.5 --no-cpu-baseline $args > gpurun_out/ab/r$n.log 2>&1
  python - "$envs" gpurun_out/ab/r$n.log <<'PY'
import json, sys
for l in open(sys.argv[2]):
    if l.startswith('{"metric"'):
        d = json.loads(l); print("%-40s %.4f ms/step" % (sys.argv[1], d["ms_per_step"]), {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernels"].items()})
PY
done
