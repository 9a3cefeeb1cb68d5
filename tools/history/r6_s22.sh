#!/bin/bash
# round 6, GPU session 22: two slices per workgroup in the 3-D assignment, wave-per-centroid scatter, 8 voxels per round in the update
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/s22
cd $REPO
timeout 1500 python -m pytest tests/test_gpu_volume.py tests/test_gpu_zz_skimage.py tests/test_gpu_zz_configs.py tests/test_gpu_zz_reference.py -m gpu -x -q > gpurun_out/s22/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/s22/pytest.log | tail -2
bash tools/c5_kstats.sh 2>&1 | tail -34 | head -16
python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/c5ks/bench.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','ms_per_step','latency_ms','ms_per_step_excluding_fit','gpu_slic_equals_scikit_image')}, d['roofline']['avg_kernel_us'])
P
