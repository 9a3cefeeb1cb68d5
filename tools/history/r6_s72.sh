#!/bin/bash
# round 6, session 72: the default line against the images in flight per GPU (4 is the default), alternating on one box
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for rep in 1 2; do for n in 4 6 8; do
  python bench.py --no-other-configs --no-cpu-baseline --steps 60 --inflight $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('inflight $n', d['value'], d['ms_per_step'], d['host_link']['fraction_of_ceiling'], d['config'].get('input_memory'))"
done; done
