#!/bin/bash
# round 6, GPU session 3: malloc thresholds vs the host fit; the driver's default command with the new line items
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/s3
cd $REPO
timeout 300 python tools/fit_malloc_probe.py > gpurun_out/s3/fit_malloc.txt 2>&1; cat gpurun_out/s3/fit_malloc.txt | tail -5
timeout 900 python bench.py > gpurun_out/s3/bench_default.json 2> gpurun_out/s3/bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/s3/bench_default.err
python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/s3/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('fit_inclusive'), d.get('host_model_fit_ms_per_step'), d.get('ms_per_step_including_fit'))
print('cpu', {k: d['cpu_baseline'].get(k) for k in ('value','model_fit_s','value_including_fit','kind')})
for k,v in d['other_configs'].items():
    print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('host_model_fit_ms_per_step'), v.get('ms_per_step_including_fit'), v.get('ms_per_step_excluding_fit'), v.get('gpu_equals_reference_run'), v.get('gpu_slic_equals_scikit_image'), v.get('wall_s'))
    print('    cpu:', (v.get('cpu_baseline') or {}).get('value'), (v.get('cpu_baseline') or {}).get('sample'))
P
