#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/phase_${1:-a}
rm -rf $OUT && mkdir -p $OUT
cd $REPO
IMSEGM_PHASE_PROF=1 IMSEGM_PHASE_DUMP=$OUT/phase.bin python bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> $OUT/phase.err
grep "phase prof" $OUT/phase.err | tail -2
python tools/phase_timeline.py $OUT/phase.bin | tee $OUT/timeline.txt; rm -f $OUT/phase.bin
