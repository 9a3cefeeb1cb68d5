#!/bin/bash
# round 4: rocprofv3 kernel statistics of config 4 with 8 images per launch chain -- one step at a time (inflight 1: un-overlapped
# kernel durations) and three steps in flight (as timed) -> gpurun_out/<tag>/kernel_stats_cfg4_batch[_inflight1].txt
TAG=${1:-r4ks}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in inflight1 load; do
  extra=""; [ $mode = inflight1 ] && extra="--inflight 1"
  rm -rf $OUT/kt
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o bench -- python $REPO/bench.py --config 4 --steps 12 --warmup 2 --no-cpu-baseline $extra > $OUT/bench_cfg4_batch_$mode.json 2> $OUT/kt_$mode.err
  DB=$(find $OUT/kt -name "*.db" | head -1)
  python $REPO/tools/prof_summary.py $DB > $OUT/kernel_stats_cfg4_batch_$mode.txt
  rm -rf $OUT/kt
done
head -60 $OUT/kernel_stats_cfg4_batch_inflight1.txt
