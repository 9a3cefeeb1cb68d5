#!/bin/bash
# quick look: bench variants + one un-overlapped rocprofv3 kernel trace (gpurun_out/qp_<tag>)
TAG=${1:-q}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/qp_$TAG
rm -rf $OUT && mkdir -p $OUT
python $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/b40.json 2> $OUT/b40.err
python $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline --pinned-input 1 > $OUT/b40_pinned.json 2>> $OUT/b40.err
python $REPO/bench.py --steps 100 --warmup 5 --no-cpu-baseline > $OUT/b100.json 2>> $OUT/b40.err
rocprofv3 --kernel-trace --stats -d $OUT/kt1 -o bench -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --inflight 1 > $OUT/b_rocprof_inflight1.json 2> $OUT/kt1.err
DB=$(find $OUT/kt1 -name "*.db" | head -1)
python $REPO/tools/prof_summary.py $DB > $OUT/kernel_stats_inflight1.txt
rm -rf $OUT/kt1
python - $OUT <<'PY'
import json, sys, os
for n in ('b40', 'b40_pinned', 'b100', 'b_rocprof_inflight1'):
    try:
        d = json.loads(open(os.path.join(sys.argv[1], n + '.json')).read().strip().splitlines()[-1])
        print(n, d['value'], 'Mpx/s', d['ms_per_step'], 'ms; cold', d['ms_per_step_incl_fill_drain'], 'lat', d.get('latency_ms'), 'resident', d.get('device_resident', {}).get('ms_per_step'), 'frac', d['roofline']['frac'])
        if n == 'b40': print(d['stage_ms_per_step'])
    except Exception as ex:
        print(n, 'ERR', ex)
PY
head -45 $OUT/kernel_stats_inflight1.txt
