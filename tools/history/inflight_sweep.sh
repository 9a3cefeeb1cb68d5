#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for n in ${@:-3 4 6}; do
  timeout 120 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --inflight $n 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight', $n, d['value'], 'Mpx/s', d['ms_per_step'], 'ms; resident', d['device_resident']['ms_per_step'], 'eq', d.get('gpu_equals_reference_run'))"
done
