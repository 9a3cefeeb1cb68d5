#!/bin/bash
# config 2 (default line): images in flight x hardware queues of the process
mkdir -p gpurun_out/c2m
for rep in 1 2; do
for qf in ${POINTS:-4:3 6:4 8:4 8:6}; do
    q=${qf%%:*}; f=${qf##*:}
    GPU_MAX_HW_QUEUES=$q timeout 120 python bench.py --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-other-configs --inflight $f $EXTRA 2>/dev/null | tail -1 > gpurun_out/c2m/q${q}_f${f}_$rep.json
    echo "rep $rep queues $q inflight $f: $(python tools/bl.py < gpurun_out/c2m/q${q}_f${f}_$rep.json | cut -c1-60)"
done
done
