#!/bin/bash
# config 4 against the number of hardware queues of the process and the images in flight (run on the GPU box)
mkdir -p gpurun_out/queues
for q in 4 8 16; do
  for f in 8 12 16 24; do
    GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --config 4 --steps 60 --warmup 5 --no-cpu-baseline --inflight $f 2>/dev/null | tail -1 > gpurun_out/queues/c4_q${q}_f${f}.json
    python - <<PY
import json
d=json.load(open('gpurun_out/queues/c4_q${q}_f${f}.json'))
print('queues', $q, 'inflight', $f, 'Mpx/s', d['value'], 'ms/step', d['ms_per_step'])
PY
  done
done
