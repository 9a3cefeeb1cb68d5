#!/bin/bash
# round 3: persistent sweep kernel -- correctness first, then A/B against the per-sweep launches and the occupancy-4 build
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${1:-r3b}; mkdir -p $OUT
cd $REPO
timeout 300 python -m pytest tests/test_gpu_sweeps.py -x -q 2>&1 | tail -15 > $OUT/pytest_sweeps.log
cat $OUT/pytest_sweeps.log
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest_all.log
cat $OUT/pytest_all.log
for mode in persistent per_sweep; do
  if [ $mode = persistent ]; then export IMSEGM_SLIC_PERSISTENT=1; else unset IMSEGM_SLIC_PERSISTENT; fi
  timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --inflight 1 > $OUT/bench_${mode}_if1.json 2>> $OUT/bench_$mode.err
  timeout 200 python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_${mode}_c4.json 2>> $OUT/bench_$mode.err
  tail -c 300 $OUT/bench_$mode.err
done
export IMSEGM_SLIC_PERSISTENT=1
cp pyimsegm_amd/libimsegm_hip.so /tmp/lib_orig.so
for v in "${@:2}"; do
  cp pyimsegm_amd/build/variants/$v.so pyimsegm_amd/libimsegm_hip.so
  timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --inflight 1 > $OUT/bench_${v}_if1.json 2>> $OUT/bench_$v.err
done
cp /tmp/lib_orig.so pyimsegm_amd/libimsegm_hip.so
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as ex:
        print(os.path.basename(f), 'NO LINE', ex); continue
    r = d['roofline']
    print(os.path.basename(f), 'value', d['value'], 'ms', d['ms_per_step'], 'lat', d.get('latency_ms'), 'res', (d.get('device_resident') or {}).get('ms_per_step'),
          'frac', r['frac'], 'us', r.get('avg_kernel_us'), 'n', r.get('launches'), 'sw', r.get('sweeps_per_launch'), 'eq', d.get('gpu_equals_reference_run'),
          'slic', d['stage_ms_per_step'].get('slic'), 'assign', d['stage_ms_per_step'].get('slic_assign'))
PY
