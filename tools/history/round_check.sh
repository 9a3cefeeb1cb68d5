#!/bin/bash
# GPU box: full -m gpu suite, the driver's bench command, the in-kernel phase profile of the assignment kernel and
# the same kernel on 8192^2 (Lab planes 1.6 GB >> the 256 MiB Infinity Cache).  Output under gpurun_out/chk_<tag>.
TAG=${1:-a}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/chk_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20.json 2>> $OUT/bench.err
IMSEGM_PHASE_PROF=1 IMSEGM_PHASE_DUMP=$OUT/phase.bin timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --inflight 1 > $OUT/bench_phase.json 2> $OUT/phase.err
grep "phase prof" $OUT/phase.err | tail -4
python tools/phase_timeline.py $OUT/phase.bin > $OUT/phase_timeline.txt 2>&1; cat $OUT/phase_timeline.txt
rm -f $OUT/phase.bin
timeout 300 python bench.py --size 8192 --steps 4 --warmup 1 --no-cpu-baseline --inflight 1 > $OUT/bench8192.json 2> $OUT/bench8192.err
python - $OUT <<'PY'
import json, sys, os
for n in ('bench', 'bench20', 'bench_phase', 'bench8192'):
    try:
        d = json.loads(open(os.path.join(sys.argv[1], n + '.json')).read().strip().splitlines()[-1])
        print(n, d['value'], 'Mpx/s', d['ms_per_step'], 'ms', 'frac', d['roofline']['frac'], 'assign us', d['roofline']['avg_kernel_us'], 'eq_ref', d.get('gpu_equals_reference_run'), d.get('gpu_equals_cpu_oracle'))
        print('   ', d['stage_ms_per_step'])
    except Exception as ex:
        print(n, 'ERR', ex)
PY
