#!/bin/bash
# PMC passes for the sparse kernels (issue / wait breakdown): scripts/pmc_sparse.sh <name> <bench args...>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
name=$1; shift
run() {
  local pass=$1; shift
  rm -rf gpurun_out/pmcs_${name}_$pass
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcs_${name}_$pass -- python bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline $ARGS > gpurun_out/pmcs_${name}_$pass.log 2>&1
  local rc=$?
  echo "pmc $pass rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; exit 1; fi
}
ARGS="$*"
if [ "${PMC_ONLY:-}" = "mem" ]; then
  run c FETCH_SIZE TCC_HIT_sum
  run d WRITE_SIZE TCC_MISS_sum TCC_EA0_ATOMIC_sum
  python scripts/pmc_summary.py gpurun_out/pmcs_${name}_c gpurun_out/pmcs_${name}_d > gpurun_out/pmcs_${name}_mem.txt 2>&1
  cat gpurun_out/pmcs_${name}_mem.txt
  exit 0
fi
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run c FETCH_SIZE TCC_HIT_sum
run d WRITE_SIZE TCC_MISS_sum TCC_EA0_ATOMIC_sum
python scripts/pmc_summary.py gpurun_out/pmcs_${name}_a gpurun_out/pmcs_${name}_b gpurun_out/pmcs_${name}_c gpurun_out/pmcs_${name}_d > gpurun_out/pmcs_${name}_summary.txt 2>&1
cat gpurun_out/pmcs_${name}_summary.txt
