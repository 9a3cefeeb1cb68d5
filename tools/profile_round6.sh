#!/bin/bash
# Round-6 profile set (summaries land in gpurun_out/prof_r06/; the ones to be judged are copied into profiles/ by hand):
#   1. the whole GPU test suite
#   2. the driver's command (default bench.py) and the same with the driver's arguments
#   3. rocprofv3 kernel statistics, one step at a time: config 2, config 3, config 4 (8 images per launch chain), config 5
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r06
rm -rf $OUT && mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err; echo "bench (driver's arguments) rc=$?"
python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/prof_r06/bench_driver_args.json').read().strip().splitlines()[-1])
print('c2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_kernel_us'], d.get('gpu_equals_reference_run'), d.get('device_resident', {}).get('value'), d.get('host_link'), d.get('ms_per_step_including_fit'))
for k,v in d['other_configs'].items(): print('   ', k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('latency_ms'), v.get('ms_per_step_including_fit'), v.get('gpu_equals_reference_run'), v.get('gpu_slic_equals_scikit_image'), (v.get('cpu_baseline') or {}).get('kind'), v.get('wall_s'), (v.get('host_link') or {}).get('fraction_of_ceiling'))
P
cd /tmp && export TMPDIR=/tmp
ks() {   # name, bench arguments
  rm -rf $OUT/kt
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o bench -- python $REPO/bench.py $2 --no-cpu-baseline --no-other-configs > $OUT/bench_$1.json 2> $OUT/kt_$1.err
  DB=$(find $OUT/kt -name "*.db" | head -1)
  python $REPO/tools/prof_summary.py $DB > $OUT/kernel_stats_$1.txt; rm -rf $OUT/kt
  head -8 $OUT/kernel_stats_$1.txt | cut -c1-140
}
ks cfg2_inflight1 "--steps 10 --warmup 2 --inflight 1"
ks cfg3_inflight1 "--config 3 --steps 3 --warmup 1 --inflight 1"
ks cfg4_batch_inflight1 "--config 4 --steps 12 --warmup 2 --inflight 1"
ks cfg5_inflight1 "--config 5 --steps 1 --warmup 1 --inflight 1"
