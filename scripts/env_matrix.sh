#!/bin/bash
# GPU parity suite under the library's mode switches (one pytest process per setting, sequentially).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
SETTINGS=("BPRX_LIST_MODE=0" "BPRX_LIST_MODE=2" "BPRX_ITEM_MODE=0" "BPRX_ITEM_MODE=2" "BPRX_ADAM_LAZY=0" "BPRX_GRAPH=1" "BPRX_GRAPH=2" "BPRX_SIDE_STREAM=0" "BPRX_SIDE_STREAM=5")
if [ -n "${MATRIX_ONLY:-}" ]; then SETTINGS=("BPRX_LIST_MODE=0" "BPRX_LIST_MODE=2" "BPRX_ITEM_MODE=0" "BPRX_ITEM_MODE=2" "BPRX_ADAM_LAZY=0" "BPRX_GRAPH=1" "BPRX_GRAPH=2" "BPRX_SIDE_STREAM=0" "BPRX_SIDE_STREAM=5")
for envs in "${SETTINGS[@]}"; do
  tag=$(echo "$envs" | tr ' =' '__')
  ( for kv in $envs; do export "$kv"; done
    timeout -k 10 600 python -m pytest tests -q -m gpu -x -k "(parity or listmode or adam or fullsize or hint or graph or train_e2e) and not replicated and not train_rec_cli" > gpurun_out/matrix_$tag.log 2>&1 )
  rc=$?
  echo "== $envs rc=$rc: $(tail -1 gpurun_out/matrix_$tag.log)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; exit 1; fi
done
