#!/bin/bash
# A/B of library variants on the same box: rocprofv3 average of one kernel (grep pattern $1), variants $2...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
cp $REPO/pyimsegm_amd/libimsegm_hip.so /tmp/lib_orig.so
for v in "$@"; do
  cp $REPO/pyimsegm_amd/build/variants/$v.so $REPO/pyimsegm_amd/libimsegm_hip.so
  rm -rf /tmp/kt_$v
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o bench -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --inflight 1 > /tmp/b_$v.json 2> /tmp/kt_$v.err
  DB=$(find /tmp/kt_$v -name "*.db" | head -1)
  echo "== $v: $(python -c "import json; d=json.loads(open('/tmp/b_$v.json').read().strip().splitlines()[-1]); print('eq_ref', d.get('gpu_equals_reference_run'), 'lat', d.get('latency_ms'))")"
  python $REPO/tools/prof_summary.py $DB | grep -E "$PAT" | cut -c1-150
done
cp /tmp/lib_orig.so $REPO/pyimsegm_amd/libimsegm_hip.so
