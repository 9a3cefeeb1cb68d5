#!/bin/bash
# round 6, session 66: the neighbour table's six cells of a voxel looked at together -- fused / volume / config tests, kernel statistics
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_fused.py tests/test_gpu_volume.py tests/test_gpu_zz_configs.py -m gpu -x -q > gpurun_out/pytest_s66.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/pytest_s66.log | tail -3
bash tools/c5_kstats.sh | grep "total kernel\|adjacency\|k_first_voxel"
