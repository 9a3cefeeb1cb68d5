#!/bin/bash
# round 6, GPU session 8: assignment kernel with entries in the brick lists + LDS-hashed boxes -- tests, phases, config 5
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/s8
cd $REPO
timeout 1500 python -m pytest tests/test_gpu_volume.py tests/test_gpu_zz_skimage.py tests/test_gpu_zz_configs.py -m gpu -x -q > gpurun_out/s8/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/s8/pytest.log
IMSEGM_HIP_LIBRARY=$PWD/pyimsegm_amd/build/variants/volprof.so timeout 300 python tools/vol_phase_probe.py 64,1024,1024 > gpurun_out/s8/vol_phases.txt 2>&1; cat gpurun_out/s8/vol_phases.txt | tail -12
bash tools/r4_c5_kstats.sh 2>&1 | tail -34 | head -14
python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/c5ks/bench.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','ms_per_step','steps','warmup','host_model_fit_ms_per_step','ms_per_step_excluding_fit','volumes_in_flight','latency_ms','latency_host_model_fit_ms','ms_per_step_incl_fill_drain','gpu_slic_equals_scikit_image')})
P
