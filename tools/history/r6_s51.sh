#!/bin/bash
# round 6, session 51: the tiles' labels through LDS and out as whole rows (variant direct: half lines straight from the tiles)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_volume.py tests/test_gpu_fused_update.py -m gpu -x -q 2>&1 | grep "passed\|failed"
bash tools/c5_kstats.sh | grep "total kernel\|k_vol_assign\|update_f32"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/direct.so bash tools/c5_kstats.sh | grep "total kernel\|k_vol_assign\|update_f32"
