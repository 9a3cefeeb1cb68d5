#!/bin/bash
# assignment kernel time for the unit variants (IMSEGM_ASSIGN_UNITS) -- un-overlapped pass of bench.py
for u in 1 2 4; do
  IMSEGM_ASSIGN_UNITS=$u python bench.py --steps 6 --warmup 2 --no-cpu-baseline --inflight 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('units', $u, 'assign us', r['avg_kernel_us'], 'frac', r['frac'], 'slic ms', d['stage_ms_per_step']['slic'], 'equal_ref', d.get('gpu_equals_reference_run'))"
done
