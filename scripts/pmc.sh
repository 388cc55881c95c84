#!/bin/bash
# PMC passes (separate runs, counters only + kernel-trace) on the bench workload; summaries under gpurun_out/pmc*.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-}"
run() {  # name counters...
  local name=$1; shift
  rm -rf gpurun_out/pmc_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -- python bench.py $ARGS > gpurun_out/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU || exit 1
run fetch FETCH_SIZE TCC_HIT_sum || exit 1
run write WRITE_SIZE TCC_MISS_sum TCC_EA0_ATOMIC_sum || exit 1
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_sq gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_summary.txt 2>&1
cat gpurun_out/pmc_summary.txt
