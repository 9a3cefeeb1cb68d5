#!/bin/bash
mkdir -p gpurun_out/c4q
for rep in 1 2; do
for qf in ${POINTS:-4:12 8:6 8:8 8:12 6:8 6:12}; do
    q=${qf%%:*}; f=${qf##*:}
    GPU_MAX_HW_QUEUES=$q timeout 120 python bench.py --config 4 --steps 60 --warmup 5 --no-cpu-baseline --inflight $f 2>/dev/null | tail -1 > gpurun_out/c4q/q${q}_f${f}_$rep.json
    echo "rep $rep queues $q inflight $f: $(python tools/bl.py < gpurun_out/c4q/q${q}_f${f}_$rep.json | cut -c1-40)"
done
done
