#!/bin/bash
# round 6, session 52: the comparison of a candidate with & and | (masks + selects) against && and || (variant branchy: the compiler's branches)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_volume.py tests/test_gpu_fused_update.py -m gpu -x -q 2>&1 | grep "passed\|failed"
bash tools/c5_kstats.sh | grep "total kernel\|k_vol_assign"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/branchy.so bash tools/c5_kstats.sh | grep "total kernel\|k_vol_assign"
