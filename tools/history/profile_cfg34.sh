#!/bin/bash
# bench lines and rocprofv3 kernel stats of bench.py --config 3 and 4 -> gpurun_out/prof_cfg34
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_cfg34
rm -rf $OUT && mkdir -p $OUT
for c in 3 4; do
  timeout 200 python $REPO/bench.py --config $c --no-cpu-baseline > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt$c -o bench -- python $REPO/bench.py --config $c --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > $OUT/bench_cfg${c}_rocprof_run.json 2> $OUT/kt$c.err
  DB=$(find $OUT/kt$c -name "*.db" | head -1)
  python $REPO/tools/prof_summary.py $DB > $OUT/rocprof_cfg${c}_kernel_stats.txt
  rm -rf $OUT/kt$c
  python -c "import json; d=json.loads(open('$OUT/bench_cfg$c.json').read().strip().splitlines()[-1]); print('cfg', $c, d['value'], d['unit'], d['ms_per_step'], 'ms/step', d.get('roofline', {}).get('frac'))"
  head -8 $OUT/rocprof_cfg${c}_kernel_stats.txt | cut -c1-140
done
