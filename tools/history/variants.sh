#!/bin/bash
# A/B of library variants (pyimsegm_amd/build/variants/*.so) on the same box: assignment kernel time, un-overlapped
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
cp pyimsegm_amd/libimsegm_hip.so /tmp/lib_orig.so
for round in 1 2; do
for v in "$@"; do
  cp pyimsegm_amd/build/variants/$v.so pyimsegm_amd/libimsegm_hip.so
  timeout 100 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --inflight 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', 'assign us', r['avg_kernel_us'], 'frac', r['frac'], 'slic ms', d['stage_ms_per_step']['slic'], 'equal_ref', d.get('gpu_equals_reference_run'))"
done
done
cp /tmp/lib_orig.so pyimsegm_amd/libimsegm_hip.so
