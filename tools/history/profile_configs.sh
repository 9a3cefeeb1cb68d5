#!/bin/bash
# rocprofv3 kernel stats of bench.py --config 3 | 4 | 5 (one image in flight), summaries into gpurun_out/prof_cfg
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_cfg
rm -rf $OUT && mkdir -p $OUT
for c in 3 4 5; do
  extra="--steps 2 --warmup 1 --inflight 1"
  [ $c = 5 ] && extra="--steps 1 --warmup 0"
  rocprofv3 --kernel-trace --stats -d $OUT/kt$c -o bench -- python $REPO/bench.py --config $c $extra --no-cpu-baseline > $OUT/bench_cfg${c}_rocprof_run.json 2> $OUT/kt$c.err
  DB=$(find $OUT/kt$c -name "*.db" | head -1)
  python $REPO/tools/prof_summary.py $DB > $OUT/rocprof_cfg${c}_kernel_stats.txt
  rm -rf $OUT/kt$c
  head -14 $OUT/rocprof_cfg${c}_kernel_stats.txt
done
