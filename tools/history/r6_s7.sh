#!/bin/bash
# round 6, GPU session 7: phase profile of the 3-D assignment kernel; config 5 with two volumes in flight
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/s7
cd $REPO
IMSEGM_HIP_LIBRARY=$PWD/pyimsegm_amd/build/variants/volprof.so timeout 300 python tools/vol_phase_probe.py 64,1024,1024 > gpurun_out/s7/vol_phases.txt 2>&1; cat gpurun_out/s7/vol_phases.txt | tail -14
timeout 600 python bench.py --config 5 --no-cpu-baseline > gpurun_out/s7/bench_c5.json 2> gpurun_out/s7/bench_c5.err; echo "bench rc=$?"; tail -3 gpurun_out/s7/bench_c5.err
python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/s7/bench_c5.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','ms_per_step','steps','warmup','host_model_fit_ms_per_step','ms_per_step_excluding_fit','volumes_in_flight','latency_ms','latency_host_model_fit_ms','ms_per_step_incl_fill_drain','gpu_slic_equals_scikit_image')})
P
