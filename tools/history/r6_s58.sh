#!/bin/bash
# round 6, session 58: the raster-order update with an eighth of the centroids per XCD (variant updxcd)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_volume.py -m gpu -x -q 2>&1 | grep "passed\|failed"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/updxcd.so python -m pytest tests/test_gpu_volume.py -m gpu -x -q 2>&1 | grep "passed\|failed"
bash tools/c5_kstats.sh | grep "update_f32"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/updxcd.so bash tools/c5_kstats.sh | grep "update_f32"
