#!/bin/bash
# per-kernel statistics of one config-5 volume
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/c5k
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/kt -o bench -- python $REPO/bench.py --config 5 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt
DB=$(find $OUT/kt -name "*.db" | head -1)
python $REPO/tools/prof_summary.py $DB > $OUT/kernel_stats_cfg5.txt
rm -rf $OUT/kt
head -30 $OUT/kernel_stats_cfg5.txt | cut -c1-150
