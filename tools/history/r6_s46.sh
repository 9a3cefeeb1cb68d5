#!/bin/bash
# round 6, session 46: graph-cut terms of a volume by the whole device, y / x blur with compile-time radii, lazy values in the update
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_fused.py tests/test_gpu_volume.py tests/test_gpu_zz_configs.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_s46.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/pytest_s46.log | tail -5
bash tools/c5_kstats.sh | head -40 | cut -c1-150
