#!/bin/bash
# round 6, session 71: the raster-order update as one sequence of rounds with the next round's labels requested ahead (variant
# rowloops: the nested z / y / x loops)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_volume.py tests/test_gpu_fused_update.py tests/test_gpu_zz_configs.py -m gpu -x -q > gpurun_out/pytest_s71.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/pytest_s71.log | tail -3
bash tools/c5_kstats.sh | grep "total kernel\|update_f32"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/rowloops.so bash tools/c5_kstats.sh | grep "total kernel\|update_f32"
