#!/bin/bash
# round 6, session 62: one-channel statistics with the limbs formed once per voxel -- volume tests, config 5 kernel statistics
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_volume.py tests/test_gpu_fused.py tests/test_gpu_zz_configs.py -m gpu -x -q > gpurun_out/pytest_s62.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/pytest_s62.log | tail -3
bash tools/c5_kstats.sh | grep "total kernel\|k_color_stats"
