#!/bin/bash
# round 6, session 47: flatten + sizes by tiles with an LDS table; the connectivity merge with eight rows per wave (variant rows8)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_connectivity.py tests/test_gpu_volume.py tests/test_gpu_zz_configs.py -m gpu -x -q > gpurun_out/pytest_s47.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/pytest_s47.log | tail -5
bash tools/c5_kstats.sh | grep "total kernel\|k_ccl_"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/rows8.so python -m pytest tests/test_gpu_connectivity.py tests/test_gpu_volume.py -m gpu -x -q 2>&1 | grep "passed\|failed"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/rows8.so bash tools/c5_kstats.sh | grep "total kernel\|k_ccl_"
