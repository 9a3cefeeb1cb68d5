#!/bin/bash
# round 6, session 44: the one-word grid barrier -- parity tests of the cut, then config 5's cut against the number of workgroups
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -m gpu -x -q 2>&1 | tail -3
for b in 256 128 64 32; do
  IMSEGM_GC_GRID_BLOCKS=$b python bench.py --config 5 --steps 1 --warmup 1 --inflight 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('blocks $b', 'graphcut stage ms', d.get('stage_ms_per_step',{}).get('graphcut'), 'ms/step', d.get('ms_per_step'))"
done
