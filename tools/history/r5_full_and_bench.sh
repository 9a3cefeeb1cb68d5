#!/bin/bash
# round 5: full GPU suite + the default line
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r5s7
rm -rf $OUT && mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/r5s7/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_kernel_us'], d.get('gpu_equals_reference_run'), d.get('device_resident'))
print(d.get('stage_ms_per_step'))
for k,v in d['other_configs'].items(): print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('gpu_equals_reference_run'), v.get('stage_ms_per_step'))
P
