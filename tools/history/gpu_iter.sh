#!/bin/bash
# one development iteration on the GPU box: full -m gpu suite, rocprofv3 kernel stats of one image at a time,
# the default bench (three in flight).  Output under gpurun_out/it_<tag>.
TAG=${1:-a}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/it_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt1 -o bench -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --inflight 1 > $OUT/b1.json 2> $OUT/kt1.err
DB=$(find $OUT/kt1 -name "*.db" | head -1)
python $REPO/tools/prof_summary.py $DB > $OUT/kernel_stats_inflight1.txt; rm -rf $OUT/kt1
cd $REPO
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/b3.json 2> $OUT/b3.err
python - $OUT <<'PY'
import json, sys, os
for n in ('b1', 'b3'):
    try:
        d = json.loads(open(os.path.join(sys.argv[1], n + '.json')).read().strip().splitlines()[-1])
        print(n, d['value'], 'Mpx/s', d['ms_per_step'], 'ms', 'frac', d['roofline']['frac'], 'assign us', d['roofline']['avg_kernel_us'], 'eq_ref', d.get('gpu_equals_reference_run'), 'lat', d.get('latency_ms'))
        print('   ', d['stage_ms_per_step'])
    except Exception as ex:
        print(n, 'ERR', ex)
PY
head -60 $OUT/kernel_stats_inflight1.txt
