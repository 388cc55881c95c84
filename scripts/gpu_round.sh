#!/bin/bash
# Runs on the GPU box via gpurun: parity tests, smoke, bench, rocprof.  Stops at the first TIMED-OUT GPU step.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
step() {  # name, timeout, cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/round.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/round.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/round.log; exit 1; fi
  return 0
}
step pytest 900 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-}
tail -5 gpurun_out/pytest.log
step smoke 300 python __graft_entry__.py --smoke
tail -2 gpurun_out/smoke.log
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  step bench 600 python bench.py --steps ${BENCH_STEPS:-30} --warmup 5
  tail -1 gpurun_out/bench.log
  if [ "${SKIP_PROF:-0}" != "1" ]; then
    rm -rf gpurun_out/prof
    step rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline
    find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -20
  fi
fi
