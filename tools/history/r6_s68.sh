#!/bin/bash
# round 6, session 68: the y / x blur's tile rows requested together
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_volume.py tests/test_gpu_zz_configs.py -m gpu -x -q > gpurun_out/pytest_s68.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/pytest_s68.log | tail -3
bash tools/c5_kstats.sh | grep "total kernel\|k_vol_blur"
