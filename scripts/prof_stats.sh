#!/bin/bash
# rocprofv3 kernel stats of one bench.py invocation: scripts/prof_stats.sh <name> <bench args...>   (env passes through)
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
name=$1; shift
mkdir -p gpurun_out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -o $name -- python bench.py --no-cpu-baseline "$@" > gpurun_out/prof_$name.log 2>&1 || { echo "rocprof failed"; tail -5 gpurun_out/prof_$name.log; exit 1; }
python - "$name" <<'PY'
import csv, glob, sys, json
name = sys.argv[1]
f = glob.glob("gpurun_out/prof_%s/**/%s_kernel_stats.csv" % (name, name), recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:18]:
    print("%-72s %6s %10.1f us" % (r["Name"].replace("(anonymous namespace)::", "")[:72], r["Calls"], float(r["AverageNs"]) / 1e3))
for l in open("gpurun_out/prof_%s.log" % name):
    if l.startswith('{"metric"'):
        print("ms_per_step", json.loads(l)["ms_per_step"])
PY
