#!/bin/bash
# round 6, session 43: the grid-wide cut of config 5 against the number of workgroups of its grid (IMSEGM_GC_GRID_BLOCKS)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for b in 256 128 64 32 16; do
  IMSEGM_GC_GRID_BLOCKS=$b python bench.py --config 5 --steps 1 --warmup 1 --inflight 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('blocks $b', 'graphcut stage ms', d.get('stage_ms_per_step',{}).get('graphcut'), 'ms/step', d.get('ms_per_step'))"
done
