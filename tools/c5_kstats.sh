#!/bin/bash
# rocprofv3 kernel statistics of config 5 at BASELINE size (one warm-up + one timed volume) -> gpurun_out/c5ks/kernel_stats.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/c5ks
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt -o bench -- python $REPO/bench.py --config 5 --steps 1 --warmup 1 --inflight 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
DB=$(find $OUT/kt -name "*.db" | head -1)
python $REPO/tools/prof_summary.py $DB > $OUT/kernel_stats.txt
rm -rf $OUT/kt
head -34 $OUT/kernel_stats.txt | cut -c1-150
