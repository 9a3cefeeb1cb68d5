#!/bin/bash
# round 6, session 60: config 5 with three volumes in flight against the gate of the host fits (slots, BLAS callers per fit)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for g in "2,28" "2,19" "3,19" "2,28" "3,19"; do
  IMSEGM_FIT_GATE=$g python bench.py --config 5 --steps 6 --warmup 2 --inflight 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('gate $g', 'ms/step', d.get('ms_per_step'), 'latency', d.get('latency_ms'), 'fit alone', d.get('latency_host_model_fit_ms'), 'fit in flight', d.get('host_model_fit_ms_in_flight'))"
done
