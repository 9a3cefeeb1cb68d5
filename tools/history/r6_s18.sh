#!/bin/bash
# config 5: two side-by-side fits at once (gate of two slots) -- 2 / 3 volumes in flight, twice each; any crash shows as rc != 0
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/s18
cd $REPO
for m in 2 3 2 3; do
  timeout 600 python bench.py --config 5 --no-cpu-baseline --inflight $m --steps $((4*m)) --warmup $m > gpurun_out/s18/bench_c5_inflight$m.json 2> gpurun_out/s18/err$m.log; echo "inflight $m rc=$?"; tail -2 gpurun_out/s18/err$m.log
  python - $m <<'P'
import json, sys
d=json.loads(open('/root/repo/gpurun_out/s18/bench_c5_inflight%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','ms_per_step','steps','warmup','host_model_fit_ms_per_step','host_model_fit_ms_in_flight','ms_per_step_excluding_fit','volumes_in_flight','latency_ms','ms_per_step_incl_fill_drain','gpu_slic_equals_scikit_image')})
P
done
