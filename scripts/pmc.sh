#!/bin/bash
# PMC passes (separate runs, counters only + kernel-trace) on the bench workload; summaries under gpurun_out/pmc*.
# The program itself follows `--` (no env / shell hop): rocprofv3's preloaded library initialises the GPU first.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --repeats 1 --no-cpu-baseline ${BENCH_ARGS:-}"
run() {  # name counters...
  local name=$1; shift
  rm -rf gpurun_out/pmc_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -- python bench.py $ARGS > gpurun_out/pmc_$name.log 2>&1
  local rc=$?
  echo "pmc $name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in pmc $name: stopping"; exit 1; fi
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run fetch FETCH_SIZE TCC_HIT_sum
run write WRITE_SIZE TCC_MISS_sum TCC_EA0_ATOMIC_sum
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
python scripts/pmc_summary.py --json gpurun_out/pmc_traffic.json gpurun_out/pmc_sq gpurun_out/pmc_mfma gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_summary.txt 2>&1
cat gpurun_out/pmc_summary.txt
