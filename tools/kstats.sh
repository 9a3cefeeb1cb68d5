#!/bin/bash
# rocprofv3 kernel stats of one image at a time -> gpurun_out/ks_<tag>/kernel_stats_inflight1.txt
TAG=${1:-a}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/ks_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt1 -o bench -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --inflight 1 > $OUT/b1.json 2> $OUT/kt1.err
DB=$(find $OUT/kt1 -name "*.db" | head -1)
python $REPO/tools/prof_summary.py $DB > $OUT/kernel_stats_inflight1.txt; rm -rf $OUT/kt1
head -${2:-70} $OUT/kernel_stats_inflight1.txt
