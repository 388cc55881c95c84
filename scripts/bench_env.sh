#!/bin/bash
# One gpurun call: "name|ENV=1 ENV2=2|bench args" -> gpurun_out/bench_<name>.log and a one-line summary per spec.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for spec in "$@"; do
  IFS='|' read -r name envs args <<< "$spec"
  ( for kv in $envs; do export "$kv"; done
    timeout -k 10 300 python bench.py --no-cpu-baseline $args > gpurun_out/bench_$name.log 2>&1 )
  rc=$?
  echo "== $name [$envs] rc=$rc"
  grep "^{\"metric\"" gpurun_out/bench_$name.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
    print('  %.4f ms/step  %.3e trip/s' % (d['ms_per_step'], d['value']))
    print('  ' + ' '.join('%s=%.1f' % (k, v['avg_ms']*1e3) for k, v in sorted(d['kernels'].items(), key=lambda kv: -kv[1]['avg_ms'])))
except Exception as e:
    print('  parse error', e)
"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; exit 1; fi
done
