#!/bin/bash
# One gpurun call that regenerates what profiles/r03_* is copied from (gpurun_out/r03/): the default bench line (driver form:
# --steps 20 --warmup 5), its rocprofv3 kernel stats, the PMC passes, the other workloads' bench lines, the forced-sharded
# N=1 lines and the two-rank rehearsals.  The program itself follows `--` under rocprofv3 (no env / shell hop).
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03; rm -rf $O; mkdir -p $O
t() { timeout -k 10 "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
echo "== default bench (driver form)"; t 600 python bench.py --steps 20 --warmup 5 > $O/bench_c2.log 2>&1; grep '^{"metric"' $O/bench_c2.log | tail -1 > $O/bench_c2.json; cut -c1-300 $O/bench_c2.json
echo "== default bench (200 steps per region)"; t 600 python bench.py --no-cpu-baseline > $O/bench_c2_200.log 2>&1; grep '^{"metric"' $O/bench_c2_200.log | tail -1 > $O/bench_c2_200.json
echo "== rocprof kernel stats"; t 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o c2 -- python bench.py --steps 200 --repeats 3 --min-timed-seconds 0 --no-cpu-baseline > $O/prof.log 2>&1
find $O/prof -name "c2_kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/bench_c2_kernel_stats.csv
t 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof5 -o c5 -- python bench.py --workload c5 --steps 30 --repeats 2 --min-timed-seconds 0 --no-cpu-baseline > $O/prof5.log 2>&1
find $O/prof5 -name "c5_kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/bench_c5_kernel_stats.csv
rm -rf $O/prof $O/prof5
echo "== pmc (c2)"; t 600 scripts/pmc.sh > $O/pmc_run.log 2>&1; cp gpurun_out/pmc_summary.txt $O/pmc_summary.txt; cp gpurun_out/pmc_traffic.json $O/pmc_traffic.json; rm -rf gpurun_out/pmc_sq gpurun_out/pmc_mfma gpurun_out/pmc_fetch gpurun_out/pmc_write
echo "== pmc (c5)"; BENCH_ARGS="--workload c5" t 600 scripts/pmc.sh > $O/pmc5_run.log 2>&1; cp gpurun_out/pmc_summary.txt $O/pmc_c5_summary.txt; cp gpurun_out/pmc_traffic.json $O/pmc_c5_traffic.json; rm -rf gpurun_out/pmc_sq gpurun_out/pmc_mfma gpurun_out/pmc_fetch gpurun_out/pmc_write   # (raw counter CSVs: > 64 MiB, gpurun would not copy anything back)
echo "== other workloads"
: > $O/bench_others.jsonl
IFS='|' read -ra LIST <<< "--workload c3shard|--workload c4shard|--workload c5 --steps 50|--workload c5list --steps 100|--workload c5small|--workload c2fp8|--workload c5bf16|--optimizer adam_tf23|--workload c4shard --optimizer adam_tf23|--workload c3shard --optimizer adam_tf23|--batch 256|--batch 4096|--batch 16384|--sampler philox|--zipf 1.0"
for spec in "${LIST[@]}"; do
  t 300 python bench.py --no-cpu-baseline $spec > $O/tmp.log 2>&1
  grep '^{"metric"' $O/tmp.log | tail -1 | python -c "
import json,sys
l=sys.stdin.read().strip()
if l:
    d=json.loads(l); d['bench_args']='$spec'; d.pop('repeats_ms', None); print(json.dumps(d))" >> $O/bench_others.jsonl
  echo "   $spec: $(tail -1 $O/bench_others.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms/step' % d['ms_per_step'], {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})" 2>/dev/null)"
done
echo "== forced sharded N=1 (RCCL world 1)"
for spec in "c2:" "c2_adam:--optimizer adam_tf23" "c2_a2a:--dist-mode a2a" "c3shard:--workload c3shard"; do
  name=${spec%%:*}; a=${spec#*:}
  BPRX_BENCH_FORCE_SHARDED=1 t 300 python bench.py --no-cpu-baseline $a > $O/tmp.log 2>&1; grep '^{"metric"' $O/tmp.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); d.pop('repeats_ms', None); print(json.dumps(d))" > $O/bench_forced_sharded_n1_$name.json
  echo "   $name: $(python -c "import json; print(json.load(open('$O/bench_forced_sharded_n1_$name.json'))['ms_per_step'])")"
done
echo "== 2-rank rehearsal on one GPU (gloo, host-staged collectives: control flow only, not a timing)"
for spec in "c2_overlap_gather:--workload c2" "c4shard_overlap_allreduce:--workload c4shard --dense-reduce allreduce" "c2_adam:--workload c2 --optimizer adam_tf23" "c3shard:--workload c3shard"; do
  name=${spec%%:*}; a=${spec#*:}
  BPRX_BENCH_REHEARSE=1 t 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 20 --warmup 3 --repeats 1 --min-timed-seconds 0 --no-cpu-baseline $a > $O/tmp.log 2>&1
  grep '^{"metric"' $O/tmp.log | tail -1 > $O/rehearse_2ranks_1gpu_gloo_$name.json
  echo "   $name: $(python -c "import json; print(json.load(open('$O/rehearse_2ranks_1gpu_gloo_$name.json'))['ms_per_step'])" 2>&1 | tail -1)"
done
echo "== reference CLI defaults"; t 300 python scripts/default_cli_profile.py > $O/default_cli_step_profile.txt 2>&1; cat $O/default_cli_step_profile.txt
t 300 python scripts/cli_epoch_bench.py > $O/cli_epoch_bench.txt 2>&1; tail -6 $O/cli_epoch_bench.txt
rm -f $O/tmp.log
