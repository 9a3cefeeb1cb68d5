#!/bin/bash
# round 6, session 67: four rows per wave in the connectivity forest's initialisation
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_connectivity.py tests/test_gpu_volume.py tests/test_gpu_zz_configs.py -m gpu -x -q > gpurun_out/pytest_s67.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/pytest_s67.log | tail -3
bash tools/c5_kstats.sh | grep "total kernel\|k_ccl_init"
