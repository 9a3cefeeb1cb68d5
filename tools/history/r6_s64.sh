#!/bin/bash
# round 6, session 64: the raster-order update with 4 / 6 / 8 quads of a lane requested per round
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for v in updq4 updq6 updq8; do
  IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/$v.so python -m pytest tests/test_gpu_volume.py -m gpu -x -q 2>&1 | grep "passed\|failed"
  IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/$v.so bash tools/c5_kstats.sh | grep "update_f32"
done
