#!/bin/bash
# round 6, session 63: the raster-order update with 1 / 2 / 4 quads of a lane requested per round
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_volume.py -m gpu -x -q 2>&1 | grep "passed\|failed"
bash tools/c5_kstats.sh | grep "update_f32"
for v in updq2 updq4; do
  IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/$v.so python -m pytest tests/test_gpu_volume.py -m gpu -x -q 2>&1 | grep "passed\|failed"
  IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/$v.so bash tools/c5_kstats.sh | grep "update_f32"
done
