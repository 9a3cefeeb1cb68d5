#!/bin/bash
# config 5, one volume at a time: allocator thresholds raised (default) against left alone, alternating on one box
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/s13
cd $REPO
for v in tuned keep tuned keep; do
  if [ $v = keep ]; then export IMSEGM_FIT_KEEP_MALLOC=1; else unset IMSEGM_FIT_KEEP_MALLOC; fi
  timeout 300 python bench.py --config 5 --no-cpu-baseline --inflight 1 --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['host_model_fit_ms_per_step'], d['latency_ms'], d['latency_host_model_fit_ms'], d['ms_per_step_excluding_fit'])"
done
