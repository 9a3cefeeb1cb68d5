#!/bin/bash
# round 5, final check: the whole GPU suite, the driver's command, kernel statistics of config 5
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r5final
rm -rf $OUT && mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err; echo "bench (driver's arguments) rc=$?"
python - <<'P'
import json
for name in ('bench_default', 'bench_driver_args'):
    d=json.loads(open('/root/repo/gpurun_out/r5final/%s.json' % name).read().strip().splitlines()[-1])
    print(name, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_kernel_us'], d.get('gpu_equals_reference_run'), d.get('device_resident', {}).get('value'), d.get('host_link', {}).get('ceiling_mpixels_per_s'))
    for k,v in d['other_configs'].items(): print('   ', k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('host_model_fit_ms_per_step'), v.get('ms_per_step_excluding_fit'), v.get('gpu_equals_reference_run'), v.get('gpu_slic_equals_scikit_image'))
P
bash tools/r4_c5_kstats.sh > /dev/null 2>&1; head -16 gpurun_out/c5ks/kernel_stats.txt | cut -c1-150
