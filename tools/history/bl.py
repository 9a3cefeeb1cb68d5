"""one-line summary of a bench.py JSON line read from stdin"""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('value', d['value'], 'ms', d['ms_per_step'], 'lat', d.get('latency_ms'), 'resident', (d.get('device_resident') or {}).get('ms_per_step'),
      'frac', r['frac'], 'us', r.get('avg_kernel_us'), 'n', r.get('launches'), 'sweeps', r.get('sweeps_per_launch'),
      'eq', d.get('gpu_equals_reference_run'), 'slic', d['stage_ms_per_step'].get('slic'), 'assign', d['stage_ms_per_step'].get('slic_assign'))
