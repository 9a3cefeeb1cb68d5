#!/bin/bash
# round 6, session 49: 32 x 8 tiles (whole lines of the label map per row) against 16 x 16 tiles in the float32 3-D assignment
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_volume.py -m gpu -x -q 2>&1 | grep "passed\|failed"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/tile32.so python -m pytest tests/test_gpu_volume.py -m gpu -x -q 2>&1 | grep "passed\|failed"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/strip64.so python -m pytest tests/test_gpu_volume.py -m gpu -x -q -k slic 2>&1 | grep "passed\|failed"
bash tools/c5_kstats.sh | grep "total kernel\|k_vol_assign"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/tile32.so bash tools/c5_kstats.sh | grep "total kernel\|k_vol_assign"
