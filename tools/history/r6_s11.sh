#!/bin/bash
# config 5 with one, two and three volumes in flight on one box
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/s11
cd $REPO
for m in 2 3 1; do
  timeout 600 python bench.py --config 5 --no-cpu-baseline --inflight $m --steps $((3*m)) --warmup $m > gpurun_out/s11/bench_c5_inflight$m.json 2> gpurun_out/s11/err$m.log; echo "inflight $m rc=$?"; tail -2 gpurun_out/s11/err$m.log
  python - $m <<'P'
import json, sys
d=json.loads(open('/root/repo/gpurun_out/s11/bench_c5_inflight%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','ms_per_step','steps','warmup','host_model_fit_ms_per_step','ms_per_step_excluding_fit','volumes_in_flight','latency_ms','latency_host_model_fit_ms','ms_per_step_incl_fill_drain','gpu_slic_equals_scikit_image')}, d['roofline']['avg_kernel_us'])
P
done
