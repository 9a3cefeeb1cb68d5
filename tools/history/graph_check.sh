#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/t.sh 150 tests
for g in 1 0 1 0; do
  [ $g = 1 ] && export IMSEGM_SLIC_GRAPH=1 || unset IMSEGM_SLIC_GRAPH
  timeout 100 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph', $g, d['value'], 'Mpx/s', d['ms_per_step'], 'ms; resident', d['device_resident']['ms_per_step'], 'latency', d['latency_ms'], 'slic stage', d['stage_ms_per_step']['slic'], 'eq', d.get('gpu_equals_reference_run'))"
done
