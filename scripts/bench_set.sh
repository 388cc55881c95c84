#!/bin/bash
# scripts/bench_set.sh <outdir> : the default C2 line + one line per other workload into gpurun_out/<outdir>/ (one box)
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$1; mkdir -p $O
t() { timeout -k 10 "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
summ() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d = json.loads(l)
        print("  %.4f ms/step  %s" % (d["ms_per_step"], {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernels"].items()}))
PY
}
echo "== c2 (driver form)"; t 300 python bench.py --steps 20 --warmup 5 ${C2_ARGS:-} > $O/bench_c2.log 2>&1; summ $O/bench_c2.log
: > $O/bench_others.jsonl
IFS='|' read -ra LIST <<< "${SPECS:---workload c3shard|--workload c4shard|--workload c5|--workload c2fp8|--optimizer adam_tf23|--sampler philox|--zipf 1.0|--batch 256|--batch 4096}"
for spec in "${LIST[@]}"; do
  t 300 python bench.py --no-cpu-baseline $spec > $O/tmp.log 2>&1
  grep '^{"metric"' $O/tmp.log | tail -1 | python -c "
import json,sys
l=sys.stdin.read().strip()
if l:
    d=json.loads(l); d['bench_args']='$spec'; print(json.dumps(d))" >> $O/bench_others.jsonl
  echo "-- $spec"; summ $O/tmp.log
done
if [ "${FORCED:-0}" = "1" ]; then
  echo "== forced sharded N=1 (RCCL world 1)"
  BPRX_BENCH_FORCE_SHARDED=1 t 300 python bench.py --no-cpu-baseline > $O/tmp.log 2>&1; grep '^{"metric"' $O/tmp.log | tail -1 > $O/bench_c2_forced_sharded_n1.json; summ $O/tmp.log
  BPRX_BENCH_FORCE_SHARDED=1 t 300 python bench.py --no-cpu-baseline --optimizer adam_tf23 > $O/tmp.log 2>&1; grep '^{"metric"' $O/tmp.log | tail -1 > $O/bench_c2_forced_sharded_n1_adam.json; summ $O/tmp.log
  BPRX_BENCH_FORCE_SHARDED=1 t 300 python bench.py --no-cpu-baseline --workload c3shard > $O/tmp.log 2>&1; grep '^{"metric"' $O/tmp.log | tail -1 > $O/bench_c3shard_forced_sharded_n1.json; summ $O/tmp.log
fi
