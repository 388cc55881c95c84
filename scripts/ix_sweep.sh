#!/bin/bash
# index-kernel timing ablations on the C2 bench (rocprofv3 kernel stats): scripts/ix_sweep.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/ix
for d in 0 1 2 4 8 12 13; do
  rm -rf gpurun_out/ix/p$d
  BPRX_IX_DBG=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ix/p$d -o x -- python bench.py --steps 30 --repeats 1 --min-timed-seconds 0 --no-cpu-baseline > gpurun_out/ix/d$d.log 2>&1
  python - $d <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/ix/p%s/**/x_kernel_stats.csv" % sys.argv[1], recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_index_seg" in r["Name"]:
        print("dbg", sys.argv[1], "k_index_seg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
