#!/bin/bash
# A/B on one box: 3-D assignment kernel at its natural register count (80, 6 waves per SIMD) against 72 (7) and 64 (8, spills)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for v in base waves7 waves8 base waves7 waves8; do
  lib=$PWD/pyimsegm_amd/libimsegm_hip.so; [ $v != base ] && lib=$PWD/pyimsegm_amd/build/variants/$v.so
  IMSEGM_HIP_LIBRARY=$lib timeout 300 python bench.py --config 5 --volume 64,2048,2048 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['roofline']['avg_kernel_us'], d['stage_ms_per_step']['slic'], d['latency_ms'])"
done
