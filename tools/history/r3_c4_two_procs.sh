#!/bin/bash
# is config 4 bound by the launch rate of ONE process?  two benchmark processes on the same GPU at the same time
mkdir -p gpurun_out/c4p
for n in 1 2 3; do
  pids=""
  for i in $(seq 1 $n); do
    timeout 200 python bench.py --config 4 --steps 120 --warmup 10 --no-cpu-baseline --inflight ${F:-8} 2>/dev/null | tail -1 > gpurun_out/c4p/n${n}_$i.json &
    pids="$pids $!"
  done
  wait $pids
  python - <<PY
import json
v=[json.load(open('gpurun_out/c4p/n${n}_%d.json' % i))['value'] for i in range(1, $n + 1)]
print('processes', $n, 'each', v, 'sum', round(sum(v), 1))
PY
done
