#!/bin/bash
# round 6, GPU session 2: the rewritten 3-D adjacency / table CSR / blur / prepared graph -- tests, then config 5 kernel statistics
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $REPO/gpurun_out/s2
cd $REPO
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_volume.py tests/test_gpu_zz_consumers.py tests/test_gpu_zz_configs.py -m gpu -x -q > gpurun_out/s2/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/s2/pytest.log
bash tools/r4_c5_kstats.sh 2>&1 | tail -40
python - <<'P'
import json
d=json.loads(open('/root/repo/gpurun_out/c5ks/bench.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','ms_per_step','host_model_fit_ms_per_step','ms_per_step_excluding_fit','gpu_slic_equals_scikit_image','stage_ms_per_step')})
P
