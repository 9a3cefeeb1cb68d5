#!/bin/bash
# assignment kernel: units per wave 1 / 2 -- parity (SLIC tests) and the un-overlapped kernel time
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/units_${1:-a}
rm -rf $OUT && mkdir -p $OUT
cd $REPO
for u in 2 1; do
  IMSEGM_ASSIGN_UNITS=$u timeout 600 python -m pytest tests -m gpu -x -q -k "slic or skimage or parity or fused or reference" > $OUT/pytest_u$u.log 2>&1; echo "units $u pytest rc=$?"; tail -2 $OUT/pytest_u$u.log
  IMSEGM_ASSIGN_UNITS=$u python bench.py --steps 10 --warmup 2 --no-cpu-baseline --inflight 1 2>$OUT/b1_u$u.err > $OUT/b1_u$u.json
  IMSEGM_ASSIGN_UNITS=$u python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>$OUT/b3_u$u.err > $OUT/b3_u$u.json
  IMSEGM_ASSIGN_UNITS=$u IMSEGM_PHASE_PROF=1 IMSEGM_PHASE_DUMP=$OUT/phase.bin python bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> $OUT/phase_u$u.err
  grep "phase prof" $OUT/phase_u$u.err | tail -2
  python tools/phase_timeline.py $OUT/phase.bin; rm -f $OUT/phase.bin
done
python - $OUT <<'PY'
import json, sys, os, glob
for f in sorted(glob.glob(os.path.join(sys.argv[1], 'b*.json'))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['value'], 'Mpx/s', d['ms_per_step'], 'ms', 'frac', d['roofline']['frac'], 'assign us', d['roofline']['avg_kernel_us'], 'eq_ref', d.get('gpu_equals_reference_run'), 'slic', d['stage_ms_per_step']['slic'])
    except Exception as ex:
        print(f, 'ERR', ex)
PY
