#!/bin/bash
# rocprofv3 kernel statistics of config 3 (one image in flight) -> gpurun_out/c3ks/kernel_stats.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/c3ks
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o bench -- python $REPO/bench.py --config 3 --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
DB=$(find $OUT/kt -name "*.db" | head -1)
python $REPO/tools/prof_summary.py $DB > $OUT/kernel_stats.txt
rm -rf $OUT/kt
head -30 $OUT/kernel_stats.txt
