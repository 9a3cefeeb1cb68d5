#!/bin/bash
# round 6, session 70: the connectivity merge's unions started from the runs' first voxels (found in registers)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_connectivity.py tests/test_gpu_volume.py tests/test_gpu_zz_configs.py tests/test_gpu_texture_size.py tests/test_gpu_golden.py -m gpu -x -q > gpurun_out/pytest_s70.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/pytest_s70.log | tail -3
bash tools/c5_kstats.sh | grep "total kernel\|k_ccl_merge"
