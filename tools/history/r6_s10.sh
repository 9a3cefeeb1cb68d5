#!/bin/bash
# A/B on one box: assignment kernel with / without the speculative first batch (one volume at a time), then the pipelined bench
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/s10
cd $REPO
for v in spec1 spec0 spec1 spec0; do
  lib=$PWD/pyimsegm_amd/libimsegm_hip.so; [ $v = spec0 ] && lib=$PWD/pyimsegm_amd/build/variants/spec0.so
  IMSEGM_HIP_LIBRARY=$lib timeout 300 python bench.py --config 5 --volume 64,2048,2048 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['roofline']['avg_kernel_us'], d['stage_ms_per_step']['slic'], d['ms_per_step'])"
done
timeout 600 python bench.py --config 5 --no-cpu-baseline > gpurun_out/s10/bench_c5.json 2> gpurun_out/s10/bench_c5.err; echo "bench rc=$?"; tail -3 gpurun_out/s10/bench_c5.err
python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/s10/bench_c5.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','ms_per_step','steps','warmup','host_model_fit_ms_per_step','ms_per_step_excluding_fit','volumes_in_flight','latency_ms','latency_host_model_fit_ms','ms_per_step_incl_fill_drain','gpu_slic_equals_scikit_image','roofline')})
P
