#!/bin/bash
# config 4: plain launches against the captured sweep graph, images in flight
mkdir -p gpurun_out/c4m
run() { # name, env..., -- args
  name=$1; shift
  env "$@" timeout 200 python bench.py --config 4 --steps 60 --warmup 5 --no-cpu-baseline --inflight $F 2>/dev/null | tail -1 > gpurun_out/c4m/$name.json
  python - <<PY
import json
d=json.load(open('gpurun_out/c4m/$name.json'))
print('$name', 'Mpx/s', d['value'], 'ms/step', d['ms_per_step'], 'equal', d.get('gpu_equals_reference_run'))
PY
}
for F in 4 6 12; do
  run plain_f$F A=1
  run graph_f$F IMSEGM_SLIC_GRAPH=1
  run sepfin_f$F IMSEGM_SEPARATE_FINALIZE=1
done
