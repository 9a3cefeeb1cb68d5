#!/bin/bash
# round 6, session 48: the connectivity merge with an eighth of every slice's row groups per XCD (variant noxcd: blocks in launch order)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
python -m pytest tests/test_gpu_connectivity.py tests/test_gpu_volume.py -m gpu -x -q 2>&1 | grep "passed\|failed"
bash tools/c5_kstats.sh | grep "total kernel\|k_ccl_merge"
IMSEGM_HIP_LIBRARY=$REPO/pyimsegm_amd/build/variants/noxcd.so bash tools/c5_kstats.sh | grep "total kernel\|k_ccl_merge"
